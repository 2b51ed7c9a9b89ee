// dcmrta_env.hip -- MI355X (gfx950) batched coalition-formation + routing environment.
//
// One wavefront (64 lanes) per env instance, one env per 64-thread workgroup.  The env's
// canonical record (S = 64 + 48*A + 96*T bytes, SURVEY.md §8d / DESIGN.md §3) is copied
// HBM -> LDS with 16-byte-per-lane coalesced loads, every phase of the reference state
// machine then runs wave-parallel on the LDS copy (lanes stride over tasks for
// task_update / mask / task observation and over agents for agent_update / agent
// observation / next_decision; wave-wide reductions use ballots and DPP row shifts),
// and the mutable part of the record is copied back.  The persistent rollout kernel keeps
// the record in LDS for whole episodes.  All times and positions are fp64 with the
// reference's operation order (compile with -ffp-contract=off); observations are rounded to
// fp32 exactly where the reference casts (worker.py:62,64).  No MFMA: the path is
// elementwise/reduction work, not a dense contraction.
//
// The simulator is a template over the batch shape (A agents, T tasks): the BASELINE shapes
// (20/50, 50/200, 100/500) are compiled with constant LDS offsets and fully unrolled
// lane-chunk loops, any other shape runs the <0,0> instantiation with runtime sizes.
// Wave-uniform state (header, choice-protocol keys, group bitmasks) lives in SGPRs.
//
// Reference restated: env/task_env.py (TaskEnv) and worker.py:41-112 (the rollout loop).
// Every device function cites the lines it follows.
#include <cstdlib>
#include <mutex>

#include "common.hpp"

using namespace dcm;

// Translation units (Makefile).  The library is built from this file twice: the main unit (-DDCM_SPLIT_G: every kernel but the
// mid-size persistent one, which it reaches through dcm::launch_rollout_fast_g) with -mllvm -phi-elim-split-all-critical-edges=1,
// and the unit of k_rollout_fast_g alone (-DDCM_TU_G) without that option -- it costs that kernel 4 % (8.19 -> 8.53 ms per 4096 x
// 70A/130T launch) while it buys the one-chunk and the 50A/200T kernel 2.1 % / 1.3 %.  With neither macro (the developer tools'
// one-command builds) everything is in one unit.
#ifdef DCM_TU_G
#define DCM_DEVICE_ONLY_TU 1
#endif
namespace dcm {
#ifndef DCM_TU_G
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* a, const char* b) {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}
#endif
// k_rollout_fast_g<NAC, NTC, OBS> (rollout_fast_g.hpp): same arguments as the kernel behind the launch geometry
void launch_rollout_fast_g(int nac, int ntc, bool obs, unsigned grid, unsigned lds_bytes, hipStream_t stream, int A, int T, int PA, int PT,
                           KP kp, unsigned char* state, int episodes, float* agents_out, float* tasks_out, uint8_t* mask_out,
                           int64_t* steps_out, double* summary, uint16_t* ablog, const int32_t* sizes, int64_t budget_all,
                           const int64_t* budget_in, unsigned char* gscr, double* retlog, int retcap);
}  // namespace dcm

namespace {

// Optional per-phase cycle accounting of the rollout kernel (tools/phase_profile.py builds a separate
// libdcmrta_prof.so with -DDCM_PROFILE_PHASES; the product build compiles these macros to nothing).
#ifdef DCM_PROFILE_PHASES
__device__ unsigned long long g_phase_cycles[16];
// k_step (lockstep kernel): per-env rows (no contended atomics: one wave owns its row).  Row e = 32 words: [0..7] phase cycles
// summed over the launches, [8..15] their maxima over the launches, [16..27] the inner marks of apply_and_advance summed,
// [28] launches in which the env was active  (tools/phase_profile.py lockstep)
#define DCM_STEP_PROF_ENVS 65536
__device__ unsigned long long g_step_rows[DCM_STEP_PROF_ENVS * 32];
#define PHK_DECL unsigned long long phk_t = __builtin_readcyclecounter(); const unsigned long long phk_start = phk_t; \
                 unsigned long long* const phk_row = g_step_rows + (size_t)(blockIdx.x % DCM_STEP_PROF_ENVS) * 32
#define PHK_ADD(i, v) do { if (lane == 0) { phk_row[i] += (v); if (phk_row[8 + (i)] < (v)) phk_row[8 + (i)] = (v); } } while (0)
#define PHK_MARK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); PHK_ADD(i, t_ - phk_t); phk_t = t_; } while (0)
#define PHK_TOTAL(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); PHK_ADD(i, t_ - phk_start); \
                          if (lane == 0) phk_row[28] += 1; } while (0)
#define PHK_INNER() do { if (lane == 0) for (int i_ = 0; i_ < 12; i_++) phk_row[16 + i_] += ph_acc[i_]; } while (0)
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PH_MARK(i) do { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t0; ph_t0 = t_; } while (0)
#define PH_FLUSH(lane) do { if ((lane) == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_phase_cycles[i_], ph_acc[i_]); } while (0)
#define PH_ARGS , unsigned long long& ph_t0, unsigned long long (&ph_acc)[12]
#define PH_PASS , ph_t0, ph_acc
// the register-resident persistent kernel (rollout_fast.hpp): its own marks, kept in the Fast<> object (tools/phase_fast.py)
__device__ unsigned long long g_fast_cycles[16];
#define FPH_MEMBERS mutable unsigned long long fph_t = 0, fph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define FPH_START(f) ((f).fph_t = __builtin_readcyclecounter())
#define FPH(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); this->fph_acc[i] += t_ - this->fph_t; this->fph_t = t_; } while (0)
#define FPHK(f, i) do { const unsigned long long t_ = __builtin_readcyclecounter(); (f).fph_acc[i] += t_ - (f).fph_t; (f).fph_t = t_; } while (0)
#define FPH_FLUSH(f, lane) do { if ((lane) == 0) for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_fast_cycles[i_], (f).fph_acc[i_]); } while (0)
#elif defined(DCM_PHASE_MARKS)
// tools/loop_insts.py --phases: the same marks as assembler comments, to attribute the decision loop's instructions to phases
#define FPH_MEMBERS
#define FPH_START(f)
#define FPH(i) asm volatile("; FPHMARK " #i)
#define FPHK(f, i) asm volatile("; FPHMARK " #i)
#define FPH_FLUSH(f, lane)
#define FPM(i) asm volatile("; FPHMARK " #i)     // sub-phase brackets (20/21 member removal, 22/23 one follower draw): marks mode only
#else
#define FPH_MEMBERS
#define FPH_START(f)
#define FPH(i)
#define FPHK(f, i)
#define FPH_FLUSH(f, lane)
#endif
#ifndef FPM
#define FPM(i)
#endif
#ifndef DCM_PROFILE_PHASES
#define PHK_DECL
#define PHK_MARK(i)
#define PHK_TOTAL(i)
#define PHK_INNER()
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH(lane)
#define PH_ARGS
#define PH_PASS
#endif

// Optional dynamic path counts of the register-resident rollout kernel (tools/path_counts.py builds a separate library with
// -DDCM_COUNT_PATHS; behind profiles/r06_budget.md).  The product build compiles CNT() to nothing.
#ifdef DCM_COUNT_PATHS
__device__ unsigned long long g_path_counts[32];
#define CNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_path_counts[i], 1ull); } while (0)
#else
#define CNT(i)
#endif

// ================================================================================== the simulator
// Three kinds of instantiation:
//   <CA, CT, false>  exact shape: sizes, record layout and lane-chunk trip counts are compile-time constants
//   <CA, CT, true>   runtime sizes (uniform or per-env) inside the CONSTANT record layout Lay{CA,CT}: every shape with
//                    A <= CA and T <= CT -- the reference's whole training range for <20,50> (parameters.py:15-16,
//                    driver.py:114-115 draws a new shape every round) -- keeps constant LDS offsets and unrolled lane-chunk
//                    loops; only the loop guards read the sizes
//   <0, 0, false>    anything else: runtime sizes and a runtime layout Lay{pA,pT}
// MG (exact multi-chunk shapes, persistent kernel only): the member arrival times f64[M][T] -- 43 % of a 50A/200T LDS image -- are
// not copied into LDS at all: the kernel works on the marr section of the env's own HBM record (one wave owns the record; a
// wave's global accesses are issued and served in order, so it sees its own stores, and the section is read only at the head of
// task_update's chain and by the terminal metrics).  The sections behind it move up by MSH bytes in the LDS image.
// MC: member slots per task (5; 16 = DCM_MAX_MEMBERS_WIDE, two id words, on a DCM_PARAM_WIDE_MEMBERS handle, which always runs the
// <0,0> instantiation).
template <int CA, int CT, bool RS, bool MG = false, int MC = M>
struct Sim {
    static_assert(MC == M || (MC == MW && CA == 0 && !MG), "wide member slots: runtime-size instantiation only");
    static constexpr int IW = (MC + 7) / 8;               // 64-bit id words per task
    using Ids = IdW<IW>;
    static constexpr int NAW = CA ? (CA + 63) / 64 : AW_MAX;  // agent chunks == words of an agent bitmask
    static constexpr int NTC = CT ? (CT + 63) / 64 : 0;       // task lane chunks (0 = runtime)
    static constexpr bool EXACT = (CA != 0) && !RS;
    // <CA, CT, true> with more than one task chunk (<128,256,true>): the template only BOUNDS the sizes -- agent-mask words and
    // lane-chunk trip counts are constants, loops unrolled -- while the record keeps the batch's own layout Lay{pA,pT} (a
    // Lay{128,256} image would be 30 KB per env whatever the batch needs); every mid-size shape shares this one instantiation
    static constexpr bool RL = RS && (CT > WAVE);
    int rA, rT;           // this env's own sizes (ignored by the exact instantiation)
    int pA, pT;           // record layout dims (read by the <0,0> instantiation only)
    unsigned char* base;  // record base (LDS in the env kernels)
    unsigned char* scr;   // terminal-metrics scratch (LDS behind the record, or this env's slice of the HBM scratch)
    double* gm = nullptr; // MG: the marr section of this env's HBM record
    static constexpr uint32_t MSH = MG ? 8u * (uint32_t)MC * (uint32_t)CT : 0u;   // = Lay::mids() - Lay::marr()
    // Exact multi-chunk shapes (50A/200T, 100A/500T): the task coordinates -- read-only instance data that only the task's own
    // lane and, for the chosen task, the whole wave ever read -- live in two registers per lane chunk (struct XY, owned by the
    // kernel and handed to observe / apply_and_advance) instead of 16 bytes per task of LDS.  The LDS image of a 50A/200T env
    // shrinks from 21.7 to 18.5 KB = 8 instead of 7 resident workgroups per CU: config 4 launch 9.63 -> 9.08 ms (-5.8 %),
    // k_step<50,200> at 16 384 envs -4 %.  (For the one-chunk layouts the same change was measured SLOWER: their kernels are
    // limited by registers, not LDS.)
    static constexpr bool IRB = (CT > WAVE) && !RS;
    static_assert(!MG || ((CT > WAVE) && !RS), "MG needs an exact multi-chunk shape");
    struct XY { double x[IRB ? NTC : 1], y[IRB ? NTC : 1]; };
    // the persistent kernel of the one-chunk shapes keeps the scratch in LDS (9.5 KB per env at 20A/50T with the per-agent term
    // lists, + 512 B of dummy slots: sixteen workgroups fill the CU's 160 KB exactly and all 4096 envs of the BASELINE batch are
    // resident); every other kernel trades it for more resident workgroups
    static constexpr bool SCR_IN_LDS = (CA != 0 && !RL && Lay{CA, CT}.lds_bytes() <= 10240);   // 16 workgroups per CU still fit

    __device__ __forceinline__ int A() const { return EXACT ? CA : rA; }
    __device__ __forceinline__ int T() const { return EXACT ? CT : rT; }
    __device__ __forceinline__ Lay L() const { return Lay{(CA && !RL) ? CA : pA, (CT && !RL) ? CT : pT, MC}; }
    __device__ __forceinline__ int PT() const { return (CT && !RL) ? CT : pT; }     // pitch of the [M][T] member-arrival slots
    // batch dims (shapes of the output arrays) given the kernel's (A,T) arguments
    __device__ __forceinline__ static int BA(int A) { return EXACT ? CA : A; }
    __device__ __forceinline__ static int BT(int T) { return EXACT ? CT : T; }
    // lane-chunk loops over tasks / agents: constant trip count (fully unrolled, guard per chunk) unless the layout is runtime
    __device__ __forceinline__ int tchunks() const { return CT ? NTC : (rT + WAVE - 1) / WAVE; }
    __device__ __forceinline__ int achunks() const { return CA ? NAW : (rA + WAVE - 1) / WAVE; }
    template <class F>
    __device__ __forceinline__ void for_tasks(int lane, F&& f) const {        // f(t) for t = lane, lane + 64, ... < T
        if constexpr (RS) {
#pragma unroll
            for (int c = 0; c < NTC; c++) { const int t = c * WAVE + lane; if (t < rT) f(t); }
        } else {
            for (int t = lane; t < T(); t += WAVE) f(t);
        }
    }
    template <class F>
    __device__ __forceinline__ void for_agents(int lane, F&& f) const {
        if constexpr (RS) {
#pragma unroll
            for (int c = 0; c < NAW; c++) { const int a = c * WAVE + lane; if (a < rA) f(a); }
        } else {
            for (int a = lane; a < A(); a += WAVE) f(a);
        }
    }
    __device__ __forceinline__ double* ax() const { return (double*)(base + L().ax()); }
    __device__ __forceinline__ double* ay() const { return (double*)(base + L().ay()); }
    __device__ __forceinline__ double* arr() const { return (double*)(base + L().arr()); }
    __device__ __forceinline__ double* nd() const { return (double*)(base + L().nd()); }
    __device__ __forceinline__ double* tdist() const { return (double*)(base + L().tdist()); }
    __device__ __forceinline__ int32_t* cur() const { return (int32_t*)(base + L().cur()); }
    __device__ __forceinline__ uint32_t* ainfo() const { return (uint32_t*)(base + L().ainfo()); }
    __device__ __forceinline__ double* ts() const { return (double*)(base + L().ts()); }
    __device__ __forceinline__ double* tf() const { return (double*)(base + L().tf()); }
    __device__ __forceinline__ double* marr() const { if constexpr (MG) return gm; else return (double*)(base + L().marr()); }
    __device__ __forceinline__ uint64_t* mids() const { return (uint64_t*)(base + L().mids() - MSH); }   // u64[IW][PT], word-major
    __device__ __forceinline__ Ids load_ids(int t) const { Ids x;
#pragma unroll
        for (int i = 0; i < IW; i++) x.w[i] = mids()[i * PT() + t];
        return x; }
    __device__ __forceinline__ void store_ids(int t, const Ids& x) const {
#pragma unroll
        for (int i = 0; i < IW; i++) mids()[i * PT() + t] = x.w[i]; }
    __device__ __forceinline__ uint32_t* tinfo() const { return (uint32_t*)(base + L().tinfo() - MSH); }
    __device__ __forceinline__ uint32_t* tnab() const { return (uint32_t*)(base + L().tnab() - MSH); }
    __device__ __forceinline__ double* tx() const { return (double*)(base + L().tx()); }
    __device__ __forceinline__ double* ty() const { return (double*)(base + L().ty()); }
    __device__ __forceinline__ double* tdur() const { return (double*)(base + (IRB ? L().tx() : L().tdur()) - MSH); }   // IRB image: no x / y sections
    __device__ __forceinline__ uint32_t aux_off() const { return (IRB ? L().tx() + 8u * (uint32_t)PT() : L().aux()) - MSH; }
    // (+ 4 T bytes of wake-up times, see task_update, for the layouts with more than one lane chunk of tasks)
    static __host__ __device__ constexpr uint32_t lds_image_bytes(Lay l) {
        return (IRB ? l.tx() + 8u * (uint32_t)l.T + 48u : l.lds_rec()) - MSH + ((CT == 0 || CT > WAVE) ? align16(4u * (uint32_t)l.T) : 0u);
    }
    // coordinates of task k (wave-uniform k): lane k & 63 of chunk k >> 6 holds them, or the LDS image does
    __device__ __forceinline__ void task_xy(int k, const XY& xy, double& x, double& y) const {
        if constexpr (IRB) {
            const int kc = k >> 6, kl = k & 63;
            x = 0.; y = 0.;
#pragma unroll
            for (int c = 0; c < NTC; c++) if (c == kc) {
                x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xy.x[c]), kl), __builtin_amdgcn_readlane(__double2loint(xy.x[c]), kl));
                y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xy.y[c]), kl), __builtin_amdgcn_readlane(__double2loint(xy.y[c]), kl));
            }
        } else { x = tx()[k]; y = ty()[k]; }
    }
    __device__ __forceinline__ double* tw() const { return (double*)(scr + L().s_tw()); }
    __device__ __forceinline__ double* aw() const { return (double*)(scr + L().s_aw()); }
    __device__ __forceinline__ uint16_t* absort() const { return (uint16_t*)(scr + L().s_absort()); }
    __device__ __forceinline__ double* terms() const { return (double*)(scr + L().s_terms()); }
    __device__ __forceinline__ unsigned long long* amask() const { return (unsigned long long*)(scr + L().s_amask()); }
    // this env's rows of the abandonment side table (pointer stashed in LDS by the kernel prologue: no SGPRs held)
    __device__ __forceinline__ uint16_t* ablog() const { return *(uint16_t* const*)(base + aux_off()); }
    __device__ __forceinline__ uint8_t* abcnt() const { return *(uint8_t* const*)(base + aux_off() + 16); }
    // (pitch_A / pitch_T = sizes the side tables are laid out for = the batch maximum; equal A() / T() unless the batch is
    //  ragged.  One workgroup per env: gridDim.x is the batch size, the count tables follow the logs of all envs.)
    __device__ __forceinline__ void set_ablog(uint16_t* table, int env_index, int pitch_A, int pitch_T, int lane) const {
        if (lane == 0) {
            *(uint16_t**)(base + aux_off()) = table + (size_t)env_index * pitch_A * AB_CAP;
            *(uint8_t**)(base + aux_off() + 16) = (uint8_t*)(table + (size_t)gridDim.x * pitch_A * AB_CAP) +
                                                  (size_t)env_index * abcnt_pitch(pitch_A, pitch_T);
        }
    }

    // record HBM -> LDS (base): every load in flight before the first LDS write when the layout is a compile-time constant
    // ALL = false: the plain chunk-by-chunk loop (4 VGPRs).  The persistent kernel copies its record once per launch of
    // hundreds of decisions, so the copy's latency is irrelevant there, while the 32 transient VGPRs of the all-in-flight copy
    // pushed k_rollout_random<64,64,runtime sizes> from 128 to 132 VGPRs = 3 instead of 4 waves per SIMD = two rounds of
    // workgroups for a 4096-env batch (5.7e8 instead of 8.1e8 steps/s at 21A/51T).
    template <bool NT = true, bool ALL = true>
    __device__ __forceinline__ void load_record(const unsigned char* rec, int lane, XY& xy) const {
        if constexpr (IRB) {
            constexpr Lay l{CA, CT};
#pragma unroll
            for (int c = 0; c < NTC; c++) {
                const int t = c * WAVE + lane < CT ? c * WAVE + lane : 0;
                xy.x[c] = ((const double*)(rec + l.tx()))[t];
                xy.y[c] = ((const double*)(rec + l.ty()))[t];
            }
            if constexpr (MG) {        // ... and the member arrival times stay where they are
                static_assert(l.marr() % 16 == 0 && l.mids() % 16 == 0 && MSH % 16 == 0, "16-byte copies");
                copy16_in(base, rec, l.marr(), lane);
                copy16_in(base + l.mids() - MSH, rec + l.mids(), l.mut_bytes() - l.mids(), lane);
                copy16_in(base + l.tx() - MSH, rec + l.tdur(), align16(8 * CT), lane);
            } else if constexpr (ALL) {       // mutable part, then the durations right behind it (the x / y sections are skipped)
                copy16_in_all<l.mut_bytes(), NT>(base, rec, lane);
                copy16_in_all<align16(8 * CT), NT>(base + l.tx(), rec + l.tdur(), lane);
            } else {
                copy16_in(base, rec, l.mut_bytes(), lane);
                copy16_in(base + l.tx(), rec + l.tdur(), align16(8 * CT), lane);
            }
        } else if constexpr (CA != 0 && !RL && ALL) copy16_in_all<Lay{CA, CT}.rec_bytes(), NT>(base, rec, lane);
        else copy16_in(base, rec, L().rec_bytes(), lane);
    }

    // mutable part of the LDS image -> HBM record
    __device__ __forceinline__ void store_record(unsigned char* rec, int lane) const {
        const Lay l = L();
        if constexpr (MG) {
            copy16(rec, base, l.marr(), lane);
            copy16(rec + l.mids(), base + l.mids() - MSH, l.mut_bytes() - l.mids(), lane);
        } else copy16(rec, base, l.mut_bytes(), lane);
    }

    // optional return log (dcm_set_return_log): this env's ring of `cap` episode returns; pointer kept in the LDS image
    __device__ __forceinline__ void set_retlog(double* log, int cap, int env_index, int lane) const {
        if (lane == 0) {
            *(double**)(base + aux_off() + 32) = log ? log + (size_t)env_index * cap : nullptr;
            *(int32_t*)(base + aux_off() + 40) = cap;
        }
    }

    struct AMask { uint64_t w[NAW]; };
    __device__ __forceinline__ static int am_count(const AMask& m) { int n = 0;
#pragma unroll
        for (int i = 0; i < NAW; i++) n += __popcll(m.w[i]);
        return n; }
    __device__ __forceinline__ static bool am_test(const AMask& m, int a) {
        bool r = false;
#pragma unroll
        for (int i = 0; i < NAW; i++) if (i == (a >> 6)) r = (m.w[i] >> (a & 63)) & 1ull;
        return r; }
    __device__ __forceinline__ static void am_clear(AMask& m, int a) {
#pragma unroll
        for (int i = 0; i < NAW; i++) if (i == (a >> 6)) m.w[i] &= ~(1ull << (a & 63)); }
    __device__ __forceinline__ static void am_set(AMask& m, int a) {
#pragma unroll
        for (int i = 0; i < NAW; i++) if (i == (a >> 6)) m.w[i] |= (1ull << (a & 63)); }
    __device__ __forceinline__ static int am_nth(const AMask& m, int idx, int lane) {
        if constexpr (NAW == 1) return nth_set_bit(m.w[0], idx, lane);
        int b = 0, res = -1;
#pragma unroll
        for (int i = 0; i < NAW; i++) {
            const int c = __popcll(m.w[i]);
            const int p = nth_set_bit(m.w[i], idx - b, lane);
            if (res < 0 && idx - b >= 0 && idx - b < c) res = i * 64 + p;
            b += c;
        }
        return res; }
    // agents whose pending group id equals g
    __device__ __forceinline__ AMask group_mask(int g, int lane) const {
        AMask m;
#pragma unroll
        for (int i = 0; i < NAW; i++) {
            const int a = i * 64 + lane;
            const uint32_t ai = ainfo()[a < A() ? a : 0];
            m.w[i] = __ballot((a < A()) && (int)((ai >> 8) & 0xFFu) == g);
        }
        return m; }

    // ------------------------------------------------------------------------------ task_update
    // env/task_env.py:245-281.  Lanes stride over tasks; each lane holds its task's <=M ordered member
    // arrivals in registers.  status = requirements - len(members) is the coalition capability-vs-
    // requirement reduction.  Straight-line predicated code; the member-removal compaction is the only
    // (rare) divergent branch.
    //
    // Incremental mode (shapes with more than one lane pass over the tasks, persistent kernel only): `only` names the task
    // the deciding agents have just joined (-2: they went to the depot, -1: every task).  At an unchanged `now` a second
    // call can only change a task whose member list was touched or that became feasible in the previous call (a
    // Q1-skipped member, the stale status after the spread branch, `finished` of a freshly feasible task :273), so when
    // the call only revisits the 64-task lane chunks the previous call touched (bitmask in inc_state()[1]) plus the chunk
    // of `only`; every other task is at a fixed point of task_update.  inc_state()[0] carries the number of infeasible
    // tasks for np.all(feasible) :279.
    // only == -3: the call of a NEW event (advance(): `now` has moved, no agent_step in between).  With the time alone a task that
    // was at a fixed point changes in two ways only: a feasible one finishes (now >= time_finish :273), or the earliest member of
    // an infeasible one has waited max_waiting_time (now - arrival >= mwt :269; also a member the rounding of that expression
    // left listed when its agent moved on).  Every visit of a task therefore leaves the time of its next possible change --
    // time_finish, or min(arrival) + mwt, +inf if neither applies, -inf if the visit itself removed members -- rounded DOWN to
    // fp32 (a margin of 1e-7 relative, far above the rounding of the two fp64 expressions) in wake()[t], an LDS-only array behind
    // the record image; the call of a new event compares `now` with it and visits the lane chunks with a due task plus those the
    // previous call touched, instead of all of them (one or two of the four at 50A/200T).  tools/inc_selfcheck.py dry-runs
    // every skipped task.
    static constexpr bool INC = (CT == 0 || CT > WAVE);
    __device__ __forceinline__ int32_t* inc_state() const { return (int32_t*)(base + aux_off() + 8); }
    __device__ __forceinline__ float* wake() const { return (float*)(base + aux_off() + 48); }    // f32[PT], INC kernels only
    // k_step only: which task sections this call has written (bit 0 time_start / time_finish, bits 1..MC member-arrival row j,
    // bit 20 member ids, bit 21 abandonment counts), so that the write-back can skip the rest (DIRTY_ALL after a reset)
    static_assert(MC + 1 < 20, "dirty mask: bits 1..MC are the arrival rows, 20 / 21 the ids / abandonment counts");
    static constexpr uint32_t DIRTY_TIMES = 1u, DIRTY_ROWS = ((1u << MC) - 1u) << 1, DIRTY_IDS = 1u << 20, DIRTY_NAB = 1u << 21,
                              DIRTY_ALL = DIRTY_TIMES | DIRTY_ROWS | DIRTY_IDS | DIRTY_NAB;
    static constexpr uint32_t DIRTY_TASK_SHIFT = 22;   // bits 22..31: the joined task (T <= 1023), valid while IDS is set and NAB is not
    static_assert(DCM_MAX_TASKS < (1 << (32 - 22)), "the joined task must fit the dirty word");
    __device__ __forceinline__ uint32_t* dirty() const { return (uint32_t*)(base + aux_off() + 24); }
    // second tracking word (lockstep kernels): bits 24..31 = how many tasks became feasible in this step, bits 0..23 = the sum of
    // their ids -- i.e. THE task when the count is 1 (the common case), so that the write-back can send its 64-byte pieces of
    // time_start / time_finish instead of the two sections
    __device__ __forceinline__ uint32_t* dirty2() const { return (uint32_t*)(base + aux_off() + 28); }
    __device__ __forceinline__ void task_update(const HdrRegs& h, const KP& P, int lane, int only = -1, bool track = false) const {
        const double now = h.now, mwt = P.mwt;
        const int T_ = T(), PT_ = PT();
        bool allf = true, touched = false;
        auto one = [&](int t) {
            uint32_t info = tinfo()[t];
            const bool feas0 = info & T_FEAS;
            const int req = info & 0xFF;
            const int n = (info >> 16) & 0xFF;                               // :250
            // Unused slots (j >= n) hold NaN in LDS (reset / compaction keep that invariant): v_max/v_min ignore
            // them and every comparison against them is false, so no per-slot validity predicate is needed.
            double av[MC];
#pragma unroll
            for (int j = 0; j < MC; j++) av[j] = marr()[j * PT_ + t];        // :251
            const double tfin = tf()[t], dur = tdur()[t];
            const int status = req - n;                                      // :252
            double mx = av[0], mn = av[0];
#pragma unroll
            for (int j = 1; j < MC; j++) { mx = nanmax2(mx, av[j]); mn = nanmin2(mn, av[j]); }
            const bool le0 = status <= 0;                                    // :254
            const bool ok = le0 && (mx - mn <= mwt);                         // :255
            const double thr = mx - mwt;                                     // :262
            // Does any member leave?  Spread branch (:262): some av[j] <= thr  <=>  mn <= thr.  Waiting branch (:269): some
            // now - av[j] >= mwt  <=>  now - mn >= mwt (fp subtraction and comparison are monotone in av[j]; the first
            // expired member is never skipped by Q1).  NaN (no members) makes both false.  The per-slot masks are only
            // built inside this rare divergent branch.
            const bool any_drop = !feas0 && (le0 ? (!ok && mn <= thr) : (now - mn >= mwt));
            if (!feas0) {
                if (ok) { ts()[t] = mx; tf()[t] = mx + dur; info |= T_FEAS; if (track) { atomicOr(dirty(), DIRTY_TIMES); atomicAdd(dirty2(), (1u << 24) | (uint32_t)t); } }  // :256-258
                int nn = n;
                if (any_drop) {  // rare: compact the surviving members in order
                    uint32_t spread = 0, q1 = 0;
                    bool prev = false;
#pragma unroll
                    for (int j = 0; j < MC; j++) {
                        spread |= (av[j] <= thr) ? (1u << j) : 0u;           // :262-265
                        // :268-271 iterates task['members'] while removing from it: after a removal the element
                        // that slides into the freed slot is skipped by the list iterator (quirk Q1).
                        const bool e = !prev && (now - av[j] >= mwt);        // :269
                        q1 |= e ? (1u << j) : 0u;
                        prev = e;
                    }
                    const uint32_t drop = le0 ? spread : q1;
                    const Ids ids = load_ids(t);
                    Ids nids{};
                    int k = 0;
#pragma unroll
                    for (int j = 0; j < MC; j++) if (j < n) {
                        const uint32_t id = ids.byte(j);
                        if (drop & (1u << j)) {
                            // abandoned_agent.append(member) :265/:271; the agent stops being listed at `t`
                            const uint32_t nth = atomicAdd(&ainfo()[id], 1u << 16) >> 16;
                            if (nth < (uint32_t)AB_CAP) ablog()[id * AB_CAP + nth] = (uint16_t)t;
                            else { const uint32_t ci = (uint32_t)(id * T_ + t); atomicAdd((uint32_t*)abcnt() + (ci >> 1), 1u << (16 * (ci & 1))); }
                            if (cur()[id] == t) atomicAnd(&ainfo()[id], ~A_MEMBER);
                        } else {
                            nids.put(k, id);
                            marr()[k * PT_ + t] = av[j];
                            k++;
                        }
                    }
                    for (int j = k; j < n; j++) marr()[j * PT_ + t] = __builtin_nan("");  // vacated slots
                    store_ids(t, nids);
                    tnab()[t] += (uint32_t)(n - k);
                    nn = k;
                    if (track) atomicOr(dirty(), DIRTY_ALL & ~DIRTY_TIMES);   // slots compacted: every arrival row, ids, counts
                }
                info = (info & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)nn << 16);
            } else {
                info |= (now >= tfin) ? T_FIN : 0u;                          // :273-274
            }
            tinfo()[t] = info;
            if constexpr (INC) {
                // when can the time alone change this task next?  (mn: the earliest listed arrival; NaN without members)
                double w = (info & T_FEAS) ? ((info & T_FIN) ? __builtin_inf() : (feas0 ? tfin : mx + dur)) : mn + mwt;
                w = (w == w) ? w : __builtin_inf();
                wake()[t] = any_drop ? -__builtin_inff() : __double2float_rd(w);
            }
            allf = allf && (info & T_FEAS);
            // a freshly feasible task only changes again at this `now` if it is already over (finished is evaluated one
            // call later, :273): now >= time_finish needs zero duration and every member already there
            touched = touched || any_drop || (!feas0 && ok && now >= mx + dur);
        };
        bool all_feasible;
        if constexpr (!INC) {
            for_tasks(lane, one);
            all_feasible = __all(allf);
        } else {
            // st[0] = number of infeasible tasks, st[1] = lane-chunk bitmask of the tasks the previous call touched
            int32_t* st = inc_state();
            const uint32_t nchunk = (uint32_t)(T_ + WAVE - 1) / WAVE, all = nchunk >= 32 ? ~0u : ((1u << nchunk) - 1u);
            uint32_t todo = all;
            if (only == -3 && T_ > WAVE) {
                uint32_t cand = 0;
                for (uint32_t ch = 0; ch < nchunk; ch++) {
                    const int t = (int)(ch * WAVE) + lane;
                    cand |= __ballot(t < T_ && now >= (double)wake()[t < T_ ? t : 0]) ? (1u << ch) : 0u;
                }
                todo = ((uint32_t)uni(st[1]) | cand) & all;
            } else if (only != -1 && only != -3 && T_ > WAVE) {
                todo = ((uint32_t)uni(st[1]) | (only >= 0 ? (1u << (only >> 6)) : 0u)) & all;
            }
            const bool full = todo == all;
            int n_infeas = full ? 0 : uni(st[0]);
            uint32_t dirty = 0;
#ifdef DCM_INC_DEBUG   // developer self-check of the chunk skipping (tools/inc_selfcheck.py); never in the product build
#include "../../tools/inc_selfcheck.inc"
#endif
            for (uint32_t c = 0; c < nchunk; c++) {                            // uniform trip count and branch: ballots inside
                if (!((todo >> c) & 1u)) continue;
                const int t = (int)(c * WAVE) + lane;
                allf = true; touched = false;
                bool was_infeasible = false;
                if (t < T_) { was_infeasible = !(tinfo()[t] & T_FEAS); one(t); }
                if (full) n_infeas += __popcll(__ballot(!allf));
                else n_infeas -= __popcll(__ballot(was_infeasible && allf));    // became feasible in this call
                if (__any(touched)) dirty |= 1u << c;
            }
            if (lane == 0) { st[0] = n_infeas; st[1] = (int32_t)dirty; }
            all_feasible = n_infeas == 0;
        }
        WSYNC();
        // depot :277-280 (np.all(feasible) is wave-uniform and false until the last task is feasible: nothing to scan before)
        if (all_feasible) {
            for_agents(lane, [&](int a) {
                const uint32_t ai = ainfo()[a];
                if ((ai & A_INDEPOT) && now >= arr()[a]) ainfo()[a] = ai | A_RETURNED;
            });
        }
    }

    // ------------------------------------------------------------------------------ agent_update
    // env/task_env.py:207-243 (non-reactive branch :226).  `agent in current_task['members']` (:230) is the
    // cached A_MEMBER bit (set by agent_step, cleared when task_update drops the agent from that task).
    __device__ __forceinline__ void agent_update(const HdrRegs& h, const KP& P, int lane) const {
        const double now = h.now;
        for_agents(lane, [&](int a) {
            const int c = cur()[a];
            const int K = c < 0 ? 0 : c;
            const uint32_t info = tinfo()[K];                                 // :228
            const double tfK = tf()[K], tsK = ts()[K], av = arr()[a];
            uint32_t ai = ainfo()[a];
            const bool member = (info & T_FEAS) && (ai & A_MEMBER);          // :229-230
            const double ndv = (c == -1) ? __builtin_nan("") : (member ? tfK : av + P.mwt);  // :226,:231,:235,:238
            const uint32_t as = member ? ((ai & A_ASSIGNED) | ((now >= tsK) ? A_ASSIGNED : 0u)) : 0u;  // :232-240
            if (c != -2) {                                                   // :209
                nd()[a] = ndv;
                if (c >= 0) ainfo()[a] = (ai & ~A_ASSIGNED) | as;            // depot leaves `assigned` untouched (Q6)
            }
        });
    }

    // ------------------------------------------------------------------------------ terminal
    // calculate_waiting_time (env/task_env.py:344-364) into LDS scratch tw[T], aw[A].
    // Returns true when a per-(agent, task) abandonment counter saturated (DCM_FLAG_WAIT_ORDER; the only inexact case).
    //
    // The per-task sums are one lane pass.  The per-agent sums (:358-364) must be accumulated in the reference's order -- tasks
    // ascending; for each task first the member term, then +max_waiting_time once per entry of the agent in that task's
    // abandoned_agent list -- a serial fp64 chain per agent.  The walking code does that literally: per agent a bitmask of the
    // tasks that list it, the agent looked up in each of them (status word, id word, arrival slot, the task's latest arrival: a
    // dependent chain of LDS round trips per task), merged with its abandonment entries sorted in place.
    // When the scratch is in LDS (the one-chunk kernels) everything but the additions is taken out of the serial part: a second
    // lane pass over the TASKS writes every member's term to its place in the agent's list (place = the number of lower tasks that
    // list the agent) together with the number of the agent's abandonment entries that belong to lower tasks (its row of the side
    // table is requested from HBM before the first pass), and the agent lane reads its list front to back and adds.  Measured on
    // the wave that ends a 20A/50T episode (s_memrealtime marks): 10.0 -> 6.4 us for the whole terminal call -- and that wave is
    // the slowest wave of a lockstep launch.  An agent listed by more than list_cap tasks or with more than eight abandonments
    // takes the walking code.
    __device__ bool compute_waits(double now, double mwt, int lane) const {
        const int T_ = T(), A_ = A(), PT_ = PT();
        const int TW = (int)L().twords();
        const int LC = (int)L().list_cap();
        // (compiled only into the simulators whose scratch can be in LDS: the code's mere presence in the out-of-line function costs the
        //  general k_step of the mid-size class 2 us per launch -- registers it clobbers are registers the caller cannot keep live)
        const bool gather = SCR_IN_LDS && uni((uint32_t)in_lds(scr)) != 0u && A_ <= WAVE && LC >= 1 && LC <= 16;
#ifdef DCM_PROFILE_PHASES
        const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
        uint4 row{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};       // this lane's agent: its first eight abandonment entries
        if (gather && lane < A_) {
            // (agent-scope loads, past the CU's vector L1: the row was written by this wave's own removal path, possibly after an
            //  earlier terminal call of the same launch had read -- and cached -- the line; the same rule as replay_fast.hpp's gload)
            const unsigned long long* q = (const unsigned long long*)(ablog() + lane * AB_CAP);
            const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            row = uint4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
        }
        for (int i = lane; i < A_ * TW; i += WAVE) amask()[i] = 0ull;         // per agent: bitmask of the tasks listing it
        WSYNC();
        for_tasks(lane, [&](int t) {
            const uint32_t info = tinfo()[t];
            const int n = (info >> 16) & 0xFF;
            const double ab = (double)tnab()[t] * mwt;
            // all M member slots are read back to back (unused ones hold NaN, which v_max_f64 ignores) instead of walking
            // the n valid ones with one LDS round trip each
            double av[MC];
#pragma unroll
            for (int j = 0; j < MC; j++) av[j] = marr()[j * PT_ + t];
            double mx = av[0];
#pragma unroll
            for (int j = 1; j < MC; j++) mx = nanmax2(mx, av[j]);            // np.max(arrival) :350
            mx = n ? mx : 0.;
            const bool feas = info & T_FEAS;
            double s = 0., term[MC];
#pragma unroll
            for (int j = 0; j < MC; j++) { term[j] = feas ? mx - av[j] : now - av[j]; s = (j < n) ? s + term[j] : s; }   // :351 / :354
            if constexpr (MC >= 8) {   // np.sum of eight or more terms is numpy's unrolled pairwise block, not a running sum
                if (n >= 8) {
                    double r8[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) r8[j] = term[j];
                    if constexpr (MC >= 16) { if (n >= 16) {
#pragma unroll
                        for (int j = 0; j < 8; j++) r8[j] += term[8 + j]; } }
                    s = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                    const int nb = n - (n % 8);
#pragma unroll
                    for (int j = 8; j < MC; j++) s = (j >= nb && j < n) ? s + term[j] : s;
                }
            }
            tw()[t] = s + ab;                                                // :351-357
            if (!gather) terms()[t] = mx;                                    // np.max(arrival), reused per agent by the walking code
            const Ids ids = load_ids(t);
            for (int j = 0; j < n; j++)                                      // transpose members -> per-agent task set
                atomicOr(&amask()[(int)ids.byte(j) * TW + (t >> 6)], 1ull << (t & 63));
        });
        uint32_t nab0 = 0;                                                   // this lane's agent: its number of abandonments
        if (gather && lane < A_) {
            // the agent's entries where the task lanes of the second pass find them; entries it does not have become 0xFFFF
            nab0 = ainfo()[lane] >> 16;
            auto pad = [&](uint32_t v, uint32_t k) { return (k < nab0 ? (v & 0xFFFFu) : 0xFFFFu) | (k + 1u < nab0 ? (v & 0xFFFF0000u) : 0xFFFF0000u); };
            *(uint4*)(absort() + lane * AB_CAP) = uint4{pad(row.x, 0u), pad(row.y, 2u), pad(row.z, 4u), pad(row.w, 6u)};
        }
        WSYNC();
        // the member term of agent a in task tm (:360 / :362), looked up
        auto member_term = [&](int a, int tm) {
            const uint32_t info = tinfo()[tm];
            const int n = (info >> 16) & 0xFF;
            const int pos = load_ids(tm).find((uint32_t)a, n);
            const double mine = marr()[pos * PT_ + tm];
            double mx;                                                       // np.max(arrival) :350
            if (gather) {                                                    // (the lists occupy the scratch the first pass would keep it in)
                mx = marr()[tm];
                for (int j = 1; j < MC; j++) mx = nanmax2(mx, marr()[j * PT_ + tm]);
            } else mx = terms()[tm];
            const double wv = now - mine;
            return (info & T_FEAS) ? (mx - mine) : ((wv > 0.) ? wv : 0.);
        };
        bool walk = lane < A_;                                               // this lane's agent still needs the walking code
        if (gather) {
            // (LDS-typed pointers: this function is out of line and sees `base` / `scr` as generic pointers -- flat instructions,
            //  64-bit addresses; here both are known to be LDS)
            typedef __attribute__((address_space(3))) unsigned char* lds_p;
            const lds_p lscr = (lds_p)scr, limg = (lds_p)base;
            auto* const l_amask = (__attribute__((address_space(3))) unsigned long long*)(lscr + L().s_amask());
            auto* const l_terms = (__attribute__((address_space(3))) double*)(lscr + L().s_terms());
            const lds_p l_rows = lscr + L().s_absort();                      // per agent 32 B: eight entries, sixteen count bytes
            auto* const l_marr = (__attribute__((address_space(3))) const double*)(limg + L().marr() - MSH);
            auto* const l_tinfo = (__attribute__((address_space(3))) const uint32_t*)(limg + L().tinfo() - MSH);
            // every member's term to its place in the agent's list, and with it the number of the agent's abandonment entries that
            // belong to lower tasks (the +max_waiting_time additions that precede the term)
            for_tasks(lane, [&](int t) {
                const uint32_t info = l_tinfo[t];
                const int n = (info >> 16) & 0xFF;
                const bool feas = info & T_FEAS;
                double av[MC];
#pragma unroll
                for (int j = 0; j < MC; j++) av[j] = l_marr[j * PT_ + t];
                double mx = av[0];
#pragma unroll
                for (int j = 1; j < MC; j++) mx = nanmax2(mx, av[j]);
                const Ids ids = load_ids(t);
                const int w = t >> 6;
                const uint64_t below = (1ull << (t & 63)) - 1ull;
                const uint32_t tt = (uint32_t)t;
#pragma unroll
                for (int j = 0; j < MC; j++) {
                    if (j < n) {
                        const int a = (int)ids.byte(j);
                        int place = __popcll(l_amask[a * TW + w] & below);
                        for (int w2 = 0; w2 < w; w2++) place += __popcll(l_amask[a * TW + w2]);
                        auto* const ev = (const __attribute__((address_space(3))) uint32_t*)(l_rows + a * (2 * AB_CAP));   // (0xFFFF: no entry)
                        const uint32_t e0 = ev[0], e1 = ev[1], e2 = ev[2], e3 = ev[3];
                        const uint32_t before = (uint32_t)((e0 & 0xFFFFu) < tt) + (uint32_t)((e0 >> 16) < tt) + (uint32_t)((e1 & 0xFFFFu) < tt) +
                                                (uint32_t)((e1 >> 16) < tt) + (uint32_t)((e2 & 0xFFFFu) < tt) + (uint32_t)((e2 >> 16) < tt) +
                                                (uint32_t)((e3 & 0xFFFFu) < tt) + (uint32_t)((e3 >> 16) < tt);
                        const double wv = now - av[j];
                        if (place < LC) {
                            l_terms[a * LC + place] = feas ? (mx - av[j]) : ((wv > 0.) ? wv : 0.);
                            (l_rows + a * (2 * AB_CAP) + 16)[place] = (unsigned char)before;
                        }
                    }
                }
            });
            WSYNC();
            if (lane < A_) {
                const int a = lane;
                int cnt = 0;
                for (int w = 0; w < TW; w++) cnt += __popcll(l_amask[a * TW + w]);
                if (cnt <= LC && nab0 <= 8u) {
                    walk = false;
                    // the list front to back (four terms requested at a time): before each term the entries of lower tasks that
                    // have not been added yet (:363-364), then the term (:360 / :362); the remaining entries at the end
                    auto* const bv = (const __attribute__((address_space(3))) uint32_t*)(l_rows + a * (2 * AB_CAP) + 16);
                    const uint32_t bw[4] = {bv[0], bv[1], bv[2], bv[3]};
                    auto* const mine = l_terms + a * LC;
                    double s = 0.;
                    int done = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (4 * c < cnt) {
                            double v[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) v[k] = mine[(4 * c + k < cnt) ? 4 * c + k : 0];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (4 * c + k < cnt) {
                                    const int before = (int)((bw[c] >> (8 * k)) & 0xFFu);
                                    for (; done < before; done++) s += mwt;
                                    s += v[k];
                                }
                            }
                        }
                    }
                    for (; done < (int)nab0; done++) s += mwt;
                    aw()[a] = s;
                }
            }
        }
#ifdef DCM_PROFILE_PHASES
        const unsigned long long pt1 = __builtin_readcyclecounter();
        if (lane == 0) atomicAdd(&g_phase_cycles[12], pt1 - pt0);
#endif
        bool over = false;
        if (!gather || __any(walk)) {
            // the walking code.  The abandonment entries come from the log (event order), sorted here by task id; entries beyond
            // AB_CAP per episode (never seen) are in the dense count table.
            {   // bring this env's rows of the side table (A x 32 B, contiguous) into LDS with one coalesced pass
                const uint4* src = (const uint4*)ablog();
                uint4* dst = (uint4*)absort();
                for (int i = lane; i < A_ * AB_CAP * 2 / 16; i += WAVE) dst[i] = src[i];
            }
            WSYNC();
            for (int a = lane; a < A_; a += WAVE) {
                if (gather && !walk) continue;
                const uint32_t nab = ainfo()[a] >> 16;
                const int nl = nab < (uint32_t)AB_CAP ? (int)nab : AB_CAP;
                uint16_t* my = absort() + a * AB_CAP;
                for (int i = 1; i < nl; i++) {                               // in-place insertion sort by task id
                    const uint16_t v = my[i];
                    int j = i;
                    while (j > 0 && my[j - 1] > v) { my[j] = my[j - 1]; j--; }
                    my[j] = v;
                }
                double s = 0.;
                if (nab <= (uint32_t)AB_CAP) {
                    // merge, in ascending task id, the tasks that list the agent (member term) with its abandonment entries
                    int p = 0;
                    for (int w = 0; w < TW; w++) {
                        uint64_t m = amask()[a * TW + w];
                        const int wend = (w + 1) * 64;
                        for (;;) {
                            const int tm = m ? (w * 64 + __ffsll((unsigned long long)m) - 1) : wend;
                            if (p < nl && (int)my[p] < tm) { s += mwt; p++; continue; }   // :363-364 of an earlier task
                            if (!m) break;
                            m &= m - 1;
                            s += member_term(a, tm);
                        }
                    }
                } else {
                    // the log overflowed: this agent from the dense count table, tasks ascending, member term first and then
                    // one +max_waiting_time per abandonment by that task (:358-364), exact for any number of abandonments
                    const uint16_t* cnt16 = (const uint16_t*)abcnt() + a * T_;   // abandonments beyond the first AB_CAP (those are in `my`, sorted)
                    int q = 0;
                    for (int t = 0; t < T_; t++) {
                        if ((amask()[a * TW + (t >> 6)] >> (t & 63)) & 1ull) s += member_term(a, t);
                        int c = cnt16[t];
                        over = over || c == 65535;                            // saturated counter: the only inexact case left
                        while (q < nl && my[q] == (uint16_t)t) { c++; q++; }
                        for (int k = 0; k < c; k++) s += mwt;
                    }
                }
                aw()[a] = s;
            }
        }
#ifdef DCM_PROFILE_PHASES
        if (lane == 0) atomicAdd(&g_phase_cycles[13], __builtin_readcyclecounter() - pt1);
#endif
        WSYNC();
        return __any(over);
    }

    // get_episode_reward + perf metrics (env/task_env.py:420-425, worker.py:87,103-108) -> row[8]
    // (header fields are passed by value: a by-reference Hdr would force the caller's header into scratch memory)
    // (called out of line through a by-VALUE copy of the simulator: a noinline member function would take `this`, which
    //  forces the Sim object -- and with it 64 bytes of scratch memory per lane and two scratch stores in every kernel
    //  prologue -- into memory)
    __device__ __noinline__ static bool terminal_metrics(Sim S, double now, double mwt, int lane, double* __restrict__ row) {
        return S.terminal_metrics_body(now, mwt, lane, row);
    }
    __device__ __forceinline__ bool terminal_metrics_body(double now, double mwt, int lane, double* __restrict__ row) const {
        WSYNC();
        const bool over = compute_waits(now, mwt, lane);
        const int T_ = T(), A_ = A();
        int nfin = 0;
        for (int t0 = 0; t0 < T_; t0 += WAVE) {
            const int t = t0 + lane;
            nfin += __popcll(__ballot(t < T_ && (tinfo()[t < T_ ? t : 0] & T_FIN)));
        }
        // :422 check_finished() once more can only re-assign the same `now`
#ifdef DCM_PROFILE_PHASES
        const unsigned long long pt2 = __builtin_readcyclecounter();
#endif
        const double Td = (double)T_, Ad = (double)A_;
        // (numpy's pairwise sum is a single block up to 128 elements: always for agents, A <= DCM_MAX_AGENTS = 128, and for
        //  tasks whenever the layout bounds T by 128)
        double m2, m3, m4, m5;
        if constexpr (CT != 0 && CT <= 128) {
            // the four sums at once, eight lanes each: lane j of a group owns numpy's accumulator r[j] (np.add.reduce's
            // unrolled block loop), the three xor-exchanges form ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) -- fp addition is
            // commutative, so every lane of the group ends with the same bits -- and the tail elements are added last
            const int g = (lane >> 3) & 3, j = lane & 7;
            const double* arr4 = g == 0 ? ts() : g == 1 ? aw() : g == 2 ? tdist() : tw();
            const int n = (g == 0 || g == 3) ? T_ : A_;
            double r;
            if (n < 8) {
                r = 0.;
                for (int i = 0; i < n; i++) r += arr4[i];
            } else {
                const int nb = n - (n % 8);
                r = arr4[j];
                for (int i = 8 + j; i < nb; i += 8) r += arr4[i];
                r += __shfl_xor(r, 1);
                r += __shfl_xor(r, 2);
                r += __shfl_xor(r, 4);
                for (int i = nb; i < n; i++) r += arr4[i];
            }
            auto lane_value = [&](int src) {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r), src), __builtin_amdgcn_readlane(__double2loint(r), src));
            };
            m2 = lane_value(0) / Td;                   // np.nanmean(time_start)      worker.py:105
            m3 = lane_value(8) / Ad;                   // np.mean(agent sum_waiting)  :106
            m4 = lane_value(16);                       // np.sum(travel_dist)         :107
            m5 = lane_value(24) / Td;                  // np.mean(task sum_waiting)   :108
        } else {
            m2 = psum<4>(ts(), T_) / Td;
            m3 = psum_block(aw(), A_) / Ad;
            m4 = psum_block(tdist(), A_);
            m5 = psum<4>(tw(), T_) / Td;
        }
#ifdef DCM_PROFILE_PHASES
        if (lane == 0) atomicAdd(&g_phase_cycles[14], __builtin_readcyclecounter() - pt2);
#endif
        if (lane == 0 && row) {
            row[0] = -now;                             // reward, env/task_env.py:424
            row[1] = (double)nfin;
            row[2] = (double)nfin / Td;                // success_rate :103
            row[3] = now;                              // makespan :104
            row[4] = m2; row[5] = m3; row[6] = m4; row[7] = m5;
        }
        return over;
    }
    __device__ __forceinline__ void terminal(HdrRegs& h, const KP& P, int lane, double* __restrict__ row) const {
#ifdef DCM_PROFILE_PHASES
        const unsigned long long pt = __builtin_readcyclecounter();
#endif
        // (the out-of-line call returns in a VGPR: without the readfirstlane the compiler treats h.flags -- and with it every loop
        //  the flags control -- as divergent, and turns the kernels' scalar branches into exec-mask bookkeeping)
        const bool over = uni((uint32_t)terminal_metrics(*this, h.now, P.mwt, lane, row)) != 0u;
#ifdef DCM_PROFILE_PHASES
        if (lane == 0) atomicAdd(&g_phase_cycles[15], __builtin_readcyclecounter() - pt);
#endif
        h.flags |= DCM_FLAG_DONE | (over ? DCM_FLAG_WAIT_ORDER : 0u);
        if (lane == 0) {                              // cold header fields stay in the LDS record
            const uint32_t n = ((Hdr*)base)->episodes;
            ((Hdr*)base)->episodes = n + 1;
            double* ring = *(double* const*)(base + aux_off() + 32);          // dcm_set_return_log
            if (ring) ring[n % (uint32_t)*(const int32_t*)(base + aux_off() + 40)] = -h.now;   // reward, env/task_env.py:424
        }
        h.cur_group = 0;
    }

    // What terminal() does without the metrics (k_step_fast with a snapshot buffer, see dcm_env::side): the caller parks the record,
    // k_terminal_flush computes reward + metrics from it later.  The informational WAIT_ORDER flag is not kept: the env restarts
    // in this very launch, which clears the flags anyway.
    static constexpr uint32_t FLAG_DEFERRED = 1u << 30;      // internal, never stored in a record
    __device__ __forceinline__ void terminal_deferred(HdrRegs& h, int lane) const {
        h.flags |= DCM_FLAG_DONE | FLAG_DEFERRED;
        if (lane == 0) {
            const uint32_t n = ((Hdr*)base)->episodes;
            ((Hdr*)base)->episodes = n + 1;
            double* ring = *(double* const*)(base + aux_off() + 32);          // dcm_set_return_log
            if (ring) ring[n % (uint32_t)*(const int32_t*)(base + aux_off() + 40)] = -h.now;   // reward, env/task_env.py:424
        }
        h.cur_group = 0;
    }

    // ------------------------------------------------------------------------------ event loop
    // Boxes D + A of SURVEY.md Appendix B: check_finished (worker.py:85, env/task_env.py:366-373), loop test
    // (worker.py:45), next_decision (:283-289), get_unique_group (:291-298), task_update, agent_update
    // (worker.py:50-51).  Returns at the next decision point or after terminal().
    // no_grouping: every deciding agent forms ONE group (individual selection, worker.py:159-198 iterates the deciders without
    // get_unique_group); lockstep API only.
    __device__ __forceinline__ void advance(HdrRegs& h, const KP& P, int lane, double* __restrict__ row PH_ARGS,
                                            bool no_grouping = false, bool track = false, bool defer = false) const {
        const int A_ = A();
        for (;;) {
            WSYNC();
            // ---- D: check_finished.  np.nanmin over next_decision (:287)
            double ndv[NAW];
            double lmin = __builtin_nan("");
#pragma unroll
            for (int i = 0; i < NAW; i++) {
                const int a = i * 64 + lane;
                ndv[i] = (a < A_) ? nd()[a] : __builtin_nan("");
                lmin = nanmin2(lmin, ndv[i]);
            }
            const double tmin = wave_nanmin(lmin);
            const bool any = (tmin == tmin);
            bool finished = false;
            if (!any) {                                                       // :368 nobody can decide any more
                double maxarr = 0.0;
                bool allret = true;
                for_agents(lane, [&](int a) {
                    const double av = (cur()[a] != -2) ? arr()[a] : 0.0;     // max(arrival_time) or 0 :286
                    maxarr = av > maxarr ? av : maxarr;
                    allret = allret && (ainfo()[a] & A_RETURNED);
                });
                // max over the WHOLE arrival lists (:286): the agents' last arrivals, and -- for lists that a masked action made
                // non-monotone -- the running maximum of the episode kept in the header
                const double hmax = uni(((const Hdr*)base)->max_arrival);
                const double lmax = wave_nanmax(maxarr);
                h.now = lmax > hmax ? lmax : hmax;                            // :369
                bool allfin = true;
                for_tasks(lane, [&](int t) { allfin = allfin && (tinfo()[t] & T_FIN); });
                finished = __all(allret) && __all(allfin);                   // :370
            }
            if (finished) h.flags |= DCM_FLAG_FINISHED;
            if (finished || h.now >= P.max_time) {                           // worker.py:45
                if (defer) terminal_deferred(h, lane); else terminal(h, P, lane, row);
                return;
            }
            // ---- A: new event
            h.n_groups = 0;
            if (any) {
                h.now = tmin;                                                 // worker.py:49
                bool dec[NAW];
                uint64_t dm[NAW];
#pragma unroll
                for (int i = 0; i < NAW; i++) { dec[i] = (ndv[i] == tmin); dm[i] = __ballot(dec[i]); }  // :288 exact ==
                int first = -1;
#pragma unroll
                for (int i = NAW - 1; i >= 0; i--) if (dm[i]) first = i * 64 + __ffsll((unsigned long long)dm[i]) - 1;
                // fast paths: a single deciding agent, or every deciding agent on the same point -> one group
                int ndec = 0;
#pragma unroll
                for (int i = 0; i < NAW; i++) ndec += __popcll(dm[i]);
                bool same = true;
                double px[NAW], py[NAW];
                if (ndec > 1 && !no_grouping)
                {
                    const double x0 = ax()[first], y0 = ay()[first];
#pragma unroll
                    for (int i = 0; i < NAW; i++) {
                        const int a = i * 64 + lane;
                        px[i] = ax()[a < A_ ? a : 0]; py[i] = ay()[a < A_ ? a : 0];
                        same = same && (!dec[i] || (px[i] == x0 && py[i] == y0));
                    }
                    same = __all(same);
                }
                if (no_grouping || same) {
#pragma unroll
                    for (int i = 0; i < NAW; i++) {
                        const int a = i * 64 + lane;
                        if (a < A_) ainfo()[a] = (ainfo()[a] & ~A_GRP) | (dec[i] ? (1u << 8) : 0u);
                    }
                    h.n_groups = 1;
                } else {
                    // general: groups in ascending (x, then y) order == rows of np.unique(axis=0) :293
                    bool todo[NAW];
                    uint32_t gid[NAW];
#pragma unroll
                    for (int i = 0; i < NAW; i++) { todo[i] = dec[i]; gid[i] = 0; }
                    int g = 0;
                    for (;;) {
                        double lx = __builtin_nan("");
#pragma unroll
                        for (int i = 0; i < NAW; i++) lx = nanmin2(lx, todo[i] ? px[i] : __builtin_nan(""));
                        const double mxv = wave_nanmin(lx);
                        if (!(mxv == mxv)) break;
                        double ly = __builtin_nan("");
#pragma unroll
                        for (int i = 0; i < NAW; i++) ly = nanmin2(ly, (todo[i] && px[i] == mxv) ? py[i] : __builtin_nan(""));
                        const double myv = wave_nanmin(ly);
                        g++;
#pragma unroll
                        for (int i = 0; i < NAW; i++) if (todo[i] && px[i] == mxv && py[i] == myv) { gid[i] = (uint32_t)g; todo[i] = false; }
                    }
#pragma unroll
                    for (int i = 0; i < NAW; i++) {
                        const int a = i * 64 + lane;
                        if (a < A_) ainfo()[a] = (ainfo()[a] & ~A_GRP) | (gid[i] << 8);
                    }
                    h.n_groups = g;
                }
            }
            WSYNC();
            PH_MARK(6);
            task_update(h, P, lane, -3, track);                               // worker.py:50
            WSYNC();
            PH_MARK(7);
            agent_update(h, P, lane);                                         // worker.py:51
            PH_MARK(8);
            if (!any) {
                if (++h.empty_passes > 4) {
                    h.flags |= DCM_FLAG_TRUNCATED;
                    if (defer) terminal_deferred(h, lane); else terminal(h, P, lane, row);
                    return;
                }
                continue;
            }
            h.empty_passes = 0;
            h.cur_group = 1;
            WSYNC();
            return;
        }
    }

    // reset + clear_decisions (env/task_env.py:116-140); keeps seed, d, episodes
    __device__ __forceinline__ void reset_state(HdrRegs& h, int lane) const {
        const int PT_ = PT();
        if constexpr (INC) { if (lane == 0) inc_state()[1] = -1; }   // incremental task_update: the next call visits every task
        for_tasks(lane, [&](int t) {
            const uint32_t req = tinfo()[t] & 0xFF;
            tinfo()[t] = req | (req << 8);       // status = requirements :131, members [], not feasible/finished
            tnab()[t] = 0;
            store_ids(t, Ids{});
            ts()[t] = 0.0; tf()[t] = 0.0;
#pragma unroll
            for (int j = 0; j < MC; j++) marr()[j * PT_ + t] = __builtin_nan("");  // empty member slots
        });
        {   // abandoned_agent = [] :131.  The count table (HBM, 2*A*T bytes) only holds the abandonments beyond the log's
            // 16 per agent, so it needs clearing only after an episode in which some agent overflowed its log
            bool spilled = false;
            for_agents(lane, [&](int a) { spilled = spilled || (ainfo()[a] >> 16) > (uint32_t)AB_CAP; });
            if (__any(spilled)) {
                uint4* c = (uint4*)abcnt();
                const int n16 = (int)(abcnt_pitch(A(), T()) / 16);
                for (int i = lane; i < n16; i += WAVE) c[i] = uint4{0u, 0u, 0u, 0u};
            }
        }
        for_agents(lane, [&](int a) {
            ax()[a] = ((const Hdr*)base)->depot_x; ay()[a] = ((const Hdr*)base)->depot_y;      // :134
            arr()[a] = 0.0; nd()[a] = 0.0; tdist()[a] = 0.0;  // :135
            cur()[a] = -2; ainfo()[a] = 0;
        });
        h.now = 0.0; h.flags = 0; h.cur_group = 0; h.n_groups = 0; h.empty_passes = 0;  // :139-140
        if (lane == 0) ((Hdr*)base)->max_arrival = 0.0;
    }

    // ------------------------------------------------------------------------------ decisions
    // worker.py:54 -- the deciding agent of the current group (protocol slot 0), or the injected one
    // (Keeping the group bitmask in SGPRs from one decision to the next was measured SLOWER: +12 % launch time from
    // the extra SGPR pressure / spills, so it is recomputed with one LDS read + ballot per decision.)
    // lowest: individual selection (DCM_PARAM_NO_GROUPING) -- the deciders act one by one in ascending id, `for agent_id in
    // decision_agents` of worker.py:170, so the next one is simply the lowest pending id (no draw)
    __device__ __forceinline__ int pick_leader(HdrRegs& h, int lane, int leader_in, uint64_t k1, AMask& gm, bool lowest = false) const {
        gm = group_mask(h.cur_group, lane);
        const int glen = am_count(gm);
        if (glen == 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return -1; }  // unreachable: groups are never empty
        if (leader_in >= 0) {
            if (leader_in >= A() || !am_test(gm, leader_in)) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return -1; }
            return leader_in;
        }
        if (lowest) return am_nth(gm, 0, lane);
        return am_nth(gm, below((uint32_t)(k1 >> 32), glen), lane);
    }

    // worker.py:57-68: mask + both observation tensors relative to `leader`, straight into the policy's input
    // tensors (fp32 casts of worker.py:62,64).
    __device__ __forceinline__ void observe(const HdrRegs& h, int lane, int leader, float* __restrict__ ag,
                                            float* __restrict__ tk, uint8_t* __restrict__ mask, const XY& xy) const {
        const double now = h.now;
        const double lx = ax()[leader], ly = ay()[leader];
        // get_current_agent_status, env/task_env.py:165-180
        if (ag) {
            for_agents(lane, [&](int a) {
                const int c = cur()[a];
                const int K = c < 0 ? 0 : c;
                const double av = arr()[a], tsK = ts()[K], durK = tdur()[K];
                const bool on = c >= 0;                                       // :168
                const double x = av - now, w = now - av, r = tsK + durK - now;
                const double travel = (on && x > 0.) ? x : 0.;                // :169
                const double waiting = (on && now <= tsK && w > 0.) ? w : 0.; // :170
                const double remaining = (on && now >= tsK && r > 0.) ? r : 0.;  // :171
                float* row = ag + 6 * a;                                      // :176-177
                row[0] = (float)travel; row[1] = (float)remaining; row[2] = (float)waiting;
                row[3] = (float)(lx - ax()[a]); row[4] = (float)(ly - ay()[a]);
                row[5] = (ainfo()[a] & A_ASSIGNED) ? 1.f : 0.f;
            });
        }
        // get_current_task_status :182-190 and get_unfinished_task_mask :192-200
        bool allmasked = true;
        for_tasks(lane, [&](int t) {
            const uint32_t info = tinfo()[t];
            const int status = (int)(int8_t)((info >> 8) & 0xFF);
            const bool unfinished = !(info & T_FEAS) && status > 0;           // :199
            allmasked = allmasked && !unfinished;
            if (mask) mask[t + 1] = unfinished ? 0 : 1;                       // :193
            if (tk) {
                float* row = tk + 5 * (t + 1);                                // :185-186
                row[0] = (float)status; row[1] = (float)(info & 0xFF); row[2] = (float)tdur()[t];
                if constexpr (IRB) {
                    double x = 0., y = 0.;
#pragma unroll
                    for (int c = 0; c < NTC; c++) if (c == (t >> 6)) { x = xy.x[c]; y = xy.y[c]; }   // (t = chunk * 64 + lane)
                    row[3] = (float)(x - lx); row[4] = (float)(y - ly);
                } else { row[3] = (float)(tx()[t] - lx); row[4] = (float)(ty()[t] - ly); }
            }
        });
        allmasked = __all(allmasked);
        // (fp64 subtract / convert on all lanes, store by one: single-lane fp64 VALU work is 4x slower on gfx950)
        const float dxf = (float)(((const Hdr*)base)->depot_x - lx), dyf = (float)(((const Hdr*)base)->depot_y - ly);
        if (lane == 0) {
            if (mask) mask[0] = allmasked ? 0 : 1;                            // worker.py:58-61
            if (tk) { tk[0] = 0.f; tk[1] = 0.f; tk[2] = 0.f; tk[3] = dxf; tk[4] = dyf; }  // :188
        }
    }

    // uniform-random valid action (protocol slot 1): valid = ascending unmasked action ids
    __device__ __forceinline__ int pick_random_action(int lane, uint64_t k1) const {
        const int T_ = T();
        if constexpr (CT != 0 && CT <= 64) {
            const uint32_t info = tinfo()[lane < T_ ? lane : 0];
            const uint64_t bm = __ballot((lane < T_) && !(info & T_FEAS) && ((int)(int8_t)((info >> 8) & 0xFF) > 0));
            const int nv = __popcll(bm);
            if (nv == 0) return 0;  // only the depot is unmasked
            return nth_set_bit(bm, below((uint32_t)k1, nv), lane) + 1;
        } else {
            int nv = 0;
            for (int t0 = 0; t0 < T_; t0 += WAVE) {
                const int t = t0 + lane;
                const uint32_t info = tinfo()[t < T_ ? t : 0];
                nv += __popcll(__ballot((t < T_) && !(info & T_FEAS) && ((int)(int8_t)((info >> 8) & 0xFF) > 0)));
            }
            if (nv == 0) return 0;
            int idx = below((uint32_t)k1, nv);
            int action = 0;
            for (int t0 = 0; t0 < T_; t0 += WAVE) {
                const int t = t0 + lane;
                const uint32_t info = tinfo()[t < T_ ? t : 0];
                const uint64_t bm = __ballot((t < T_) && !(info & T_FEAS) && ((int)(int8_t)((info >> 8) & 0xFF) > 0));
                const int c = __popcll(bm);
                const int p = nth_set_bit(bm, idx, lane);
                if (action == 0 && idx >= 0 && idx < c) action = t0 + p + 1;
                idx -= c;
            }
            return action;
        }
    }

    // TaskEnv.step (env/task_env.py:326-342) + agent_step (:300-324) for leader + followers, then
    // task_update / agent_update (worker.py:74-76) and the move to the next decision point.
    // DEV: the action comes from the device's own valid-action policy on a validated instance (persistent kernel): the error
    // exits below cannot be taken and are compiled out -- each `return` in the middle of the step costs the structured control
    // flow of the whole function scalar bookkeeping at every decision.
    template <bool DEV = false>
    __device__ __forceinline__ void apply_and_advance(HdrRegs& h, const KP& P, int lane, int leader, const AMask& gm0,
                                                      int action, uint64_t k1, int nfol_in,
                                                      const int16_t* __restrict__ fol_in, double* __restrict__ row PH_ARGS,
                                                      RouteLog log = RouteLog{nullptr, nullptr, nullptr, 0}, int log_row = 0,
                                                      bool no_grouping = false, int host_actions = 0,
                                                      bool incremental = false, bool track = false, const XY* xyp = nullptr) const {
        // host_actions: 0 = the action comes from the device's own valid-action policy (persistent kernel); 1 = from the host
        // (lockstep API): ANY action in [0, T] is simulated as TaskEnv.step would (env/task_env.py:326-342 has no mask check) --
        // on a masked task (feasible, or status <= 0 incl. the stale status of quirk Q3, :192-200) vacancy <= 0 sends the
        // leader alone (:330-336), the task may then list more members than it requires, and an agent released before it
        // arrives makes its arrival list non-monotone, which is why the header keeps the episode's running maximum; 2 = from
        // the host with DCM_PARAM_STRICT_MASK: such an action freezes the env instead (DCM_FLAG_BAD_ACTION).
        const int A_ = A(), T_ = T();
        if constexpr (!DEV) { if (action < 0 || action > T_) { h.flags |= DCM_FLAG_BAD_ACTION | DCM_FLAG_DONE; return; } }
        if (!DEV && host_actions == 2 && action > 0) {
            const uint32_t ik = uni(tinfo()[action - 1]);
            if ((ik & T_FEAS) || (int)(int8_t)((ik >> 8) & 0xFF) <= 0) { h.flags |= DCM_FLAG_BAD_ACTION | DCM_FLAG_DONE; return; }
        }
        AMask rest = gm0;
        am_clear(rest, leader);                                               // :328 group.remove(leader)
        int rlen = am_count(rest);
        AMask mm;                                                             // members of this step
#pragma unroll
        for (int i = 0; i < NAW; i++) mm.w[i] = 0;
        am_set(mm, leader);
        Ids mlist{};                                                          // ordered member ids of this step, byte j (task actions only)
        mlist.put(0, (uint32_t)leader);
        int nm = 1;
        double tx_, ty_;
        if (action == 0 && nfol_in < 0 && !no_grouping) {   // (individual selection: agent_step(agent, 0) moves that agent only)
            // vacancy = len(group) (:327): every co-located agent returns with the leader (Q9); the draw
            // order of the followers cannot change any state, so no draws are spent.
#pragma unroll
            for (int i = 0; i < NAW; i++) mm.w[i] |= rest.w[i];
            nm += rlen; rlen = 0;
            tx_ = ((const Hdr*)base)->depot_x; ty_ = ((const Hdr*)base)->depot_y;
        } else {
            const int k = action - 1;
            int nf;
            if (nfol_in >= 0) nf = nfol_in;                                   // injected (also for the depot: individual selection)
            else if (rlen == 0 || no_grouping) nf = 0;                        // nobody left to follow / individual selection: agent_step alone (worker.py:186)
            else {
                const int vacancy = (int)(int8_t)((uni(tinfo()[k]) >> 8) & 0xFF);  // :327 task status (may be stale)
                nf = (vacancy > 1) ? ((vacancy - 1 < rlen) ? vacancy - 1 : rlen) : 0;  // :330-331
            }
            if constexpr (!DEV) {
                if (nf > MC - 1 || nf > rlen || (nfol_in >= 0 && nf > DCM_FOLLOWER_COLS)) { h.flags |= DCM_FLAG_OVERFLOW | DCM_FLAG_DONE; return; }
            }
            uint64_t kk = k1;
            for (int j = 0; j < nf; j++) {                                    // :331 choice without replacement
                int f;
                if (nfol_in >= 0) {
                    f = fol_in[j];
                    if (f < 0 || f >= A_ || !am_test(rest, f)) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return; }
                } else {
                    if ((j & 1) == 0) kk = mix64(kk + GAMMA);                 // key_{2+j/2}
                    const uint32_t r = (j & 1) ? (uint32_t)kk : (uint32_t)(kk >> 32);
                    f = am_nth(rest, below(r, rlen), lane);
                }
                am_clear(rest, f); rlen--;                                    // :332-333
                am_set(mm, f);
                mlist.put(nm, (uint32_t)f);
                nm++;
            }
            if (action == 0) { tx_ = ((const Hdr*)base)->depot_x; ty_ = ((const Hdr*)base)->depot_y; }
            else task_xy(k, *xyp, tx_, ty_);
        }
        PH_MARK(11);
        // agent_step for every member (:300-324); independent per agent.  The fp64 arithmetic (distance, sqrt, division) runs
        // on ALL lanes with a clamped agent index and only the stores are predicated on membership: gfx950 executes an fp64
        // VALU instruction with fewer than 16 active lanes 4x slower (17 instead of 4.2 clocks, profiles/r03_calib), and a
        // step moves 1..5 agents.
        double arrv[NAW];
#pragma unroll
        for (int i = 0; i < NAW; i++) {
            const int a = i * 64 + lane;
            const int ac = a < A_ ? a : 0;
            const double d = dist2(ax()[ac], ay()[ac], tx_, ty_);
            const double travel_time = over_velocity(d);                              // :315 velocity 0.2 (:99)
            const double td = tdist()[ac] + d;                                        // :317
            arrv[i] = h.now + travel_time;                                            // :318
            if (a < A_ && ((mm.w[i] >> lane) & 1ull)) {
                tdist()[a] = td;
                arr()[a] = arrv[i];
                ax()[a] = tx_; ay()[a] = ty_;                                 // :320
                cur()[a] = action - 1;                                        // :314 route.append
                uint32_t ai = ainfo()[a] & ~(A_GRP | A_MEMBER);               // leaves the pending group
                ai |= (action == 0) ? A_INDEPOT : A_MEMBER;                   // :321-322 listed in the target's members
                ainfo()[a] = ai;
                if (log.len) {                                                // route.append / arrival_time += (:314,:318)
                    const size_t o = (size_t)log_row + a;                     // log_row = env index x agents per env
                    const int c = log.len[o];
                    if (c < log.cap) { log.task[o * log.cap + c] = (int16_t)(action - 1); log.arrival[o * log.cap + c] = arrv[i]; }
                    log.len[o] = c + 1;
                }
            }
        }
        if (host_actions) {
            // running maximum of every arrival appended in this episode (see Hdr::max_arrival)
            double m = __builtin_nan("");
#pragma unroll
            for (int i = 0; i < NAW; i++) m = nanmax2(m, ((mm.w[i] >> lane) & 1ull) ? arrv[i] : __builtin_nan(""));
            const double wm = wave_nanmax(m);
            if (lane == 0) { Hdr* q = (Hdr*)base; if (wm > q->max_arrival) q->max_arrival = wm; }
        }
        if (action > 0) {
            // :321-322 members.append unless already listed; a re-joining agent keeps its slot but
            // get_arrival_time (:202-205) now returns the new, later arrival (Q4).  Wave-uniform loop over the
            // 1..5 members in order; "already listed" is a SWAR byte search in the packed ordered id word.
            const int k = action - 1;
            const uint32_t info = uni(tinfo()[k]);
            Ids ids = load_ids(k);
#pragma unroll
            for (int i = 0; i < IW; i++) ids.w[i] = uni(ids.w[i]);
            int n = (info >> 16) & 0xFF;
            for (int j = 0; j < nm; j++) {
                const int m = (int)mlist.byte(j);
                int pos = ids.find((uint32_t)m, n);
                if (pos < 0) {
                    if constexpr (!DEV) { if (n >= MC) { h.flags |= DCM_FLAG_OVERFLOW | DCM_FLAG_DONE; return; } }
                    pos = n++;
                    ids.put(pos, (uint32_t)m);                                // bytes above n are always zero
                }
                // the arrival was computed by agent m's own lane above: that lane stores it (no readlane round trip through
                // the scalar unit, and the store does not wait for the sqrt chain of the other members)
#pragma unroll
                for (int i = 0; i < NAW; i++) if (i * WAVE + lane == m) marr()[pos * PT() + k] = arrv[i];
                // (bits 22..31 of the dirty word: the task that was joined -- one per step -- so that the write-back can send its
                //  64-byte pieces of the arrival rows / member ids instead of the sections)
                if (track && lane == 0) *dirty() = (*dirty() & ((1u << DIRTY_TASK_SHIFT) - 1u)) | (2u << pos) | DIRTY_IDS | ((uint32_t)k << DIRTY_TASK_SHIFT);
            }
            if (lane == 0) { store_ids(k, ids); tinfo()[k] = (info & ~0x00FF0000u) | ((uint32_t)n << 16); }
        }
        h.d += 1;
        WSYNC();
        PH_MARK(3);
        task_update(h, P, lane, incremental ? (action > 0 ? action - 1 : -2) : -1, track);   // worker.py:74
        WSYNC();
        PH_MARK(4);
        agent_update(h, P, lane);                                             // worker.py:76
        WSYNC();
        PH_MARK(5);
        if (rlen > 0) return;                                                 // worker.py:53 same group, next leader
        if (h.cur_group < h.n_groups) { h.cur_group++; return; }              // worker.py:52 next group
        advance(h, P, lane, row PH_PASS, no_grouping, track);                 // worker.py:85 -> :45
        PH_MARK(9);
    }

    __device__ __forceinline__ void write_inactive_obs(int lane, float* ag, float* tk, uint8_t* mask) const {
        if (ag) for (int i = lane; i < 6 * A(); i += WAVE) ag[i] = 0.f;
        if (tk) for (int i = lane; i < 5 * (T() + 1); i += WAVE) tk[i] = 0.f;
        if (mask) for (int i = lane; i <= T(); i += WAVE) mask[i] = (i == 0) ? 0 : 1;
    }
    // Ragged batch: rows beyond this env's own A / T+1 are padding in the convention the policy already understands --
    // every feature -1 (attention.py:10-18 get_attn_pad_mask, worker.py:253-261 zero_padding) and mask True (true_padding).
    __device__ __forceinline__ void write_pad_obs(int lane, int pitch_A, int pitch_T, float* ag, float* tk, uint8_t* mask) const {
        if (ag) for (int i = 6 * A() + lane; i < 6 * pitch_A; i += WAVE) ag[i] = -1.f;
        if (tk) for (int i = 5 * (T() + 1) + lane; i < 5 * (pitch_T + 1); i += WAVE) tk[i] = -1.f;
        if (mask) for (int i = T() + 1 + lane; i <= pitch_T; i += WAVE) mask[i] = 1;
    }
};

// Three levels of dims.  (A,T) = batch dims: the shapes of every input / output array and the default env sizes.
// (eA,eT) = this env's own sizes: (A,T) for a uniform batch, sizes[e] for a ragged one (dcm_load_instances_ragged).
// (PA,PT) = layout dims of the record (Lay{PA,PT}, the same for every env of the handle, >= the batch dims): the
// template constants, or the kernel arguments for the <0,0> instantiation.  Only the entries below an env's own sizes
// are ever touched, so the simulator code is the same for all three kinds of instantiation (see Sim).
template <int CA, int CT, bool RS>
__device__ __forceinline__ void env_dims(const int32_t* sizes, int e, int A, int T, int& eA, int& eT) {
    eA = A; eT = T;
    if constexpr (RS || CA == 0) {
        if (sizes) { eA = uni(sizes[2 * e]); eT = uni(sizes[2 * e + 1]); }
    }
}

// ---------------------------------------------------------------------------------- kernels
// XCD-aware env placement of the one-workgroup-per-env kernels.  Workgroups are dealt round-robin over the 8 XCDs (each with its own
// L2 and its own path to memory): workgroup b runs env  xcd(b) * ceil(B/8) + b/8,  i.e. every XCD owns a CONTIGUOUS block of envs.
//  * persistent kernel: the observation rows an XCD rewrites at every decision are one dense region of its L2 instead of every
//    eighth 480 / 1020-byte row, which aliased in the L2 sets and sent ~10 % of the per-decision stores to HBM (243 MB per
//    4096-env launch; 55 MB = the compulsory traffic with this map, profiles/r03_xcd_map);
//  * lockstep kernel: with e = blockIdx.x all XCDs stream through the SAME neighbourhood of the state array at any moment; with
//    contiguous blocks they work in eight separate regions: 147-170 -> 136 us at 65 536 envs (0.80 of the HBM peak).
__device__ __forceinline__ int env_of_workgroup() {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    return x * q + (x < r ? x : r) + i;
}

__global__ __launch_bounds__(WAVE) void k_load_instances(int A, int T, int PA, int PT, int PC, unsigned char* state, const double* depot,
                                                        const double* task_xy, const int32_t* req, const double* dur,
                                                        const int32_t* sizes) {
    const int e = blockIdx.x, lane = threadIdx.x;
    int eA, eT;
    env_dims<0, 0, false>(sizes, e, A, T, eA, eT);
    const Lay L{PA, PT, PC};                                   // input arrays are pitched by the batch dims (A,T)
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    double *tx = (double*)(rec + L.tx()), *ty = (double*)(rec + L.ty()), *td = (double*)(rec + L.tdur());
    uint32_t* ti = (uint32_t*)(rec + L.tinfo());
    bool bad = false;
    for (int t = lane; t < eT; t += WAVE) {
        tx[t] = task_xy[((size_t)e * T + t) * 2];
        ty[t] = task_xy[((size_t)e * T + t) * 2 + 1];
        td[t] = dur[(size_t)e * T + t];
        // requirements outside 1..(member slots of the handle) do not fit the member slots (the reference takes any max_coalition_size,
        // env/task_env.py:71): the env is marked and stays frozen instead of simulating something else
        const int32_t r = req[(size_t)e * T + t];
        bad = bad || r < 1 || r > PC;
        ti[t] = (uint32_t)(r < 1 ? 1 : r > PC ? PC : r);
    }
    bad = __any(bad);
    if (lane == 0) {
        Hdr* h = (Hdr*)rec;
        h->depot_x = depot[2 * (size_t)e]; h->depot_y = depot[2 * (size_t)e + 1];
        h->flags = DCM_FLAG_DONE | (bad ? DCM_FLAG_BAD_INSTANCE : 0u); h->episodes = 0; h->d = 0; h->seed = 0;
        h->groups = 0; h->reserved = 0; h->max_arrival = 0.0;
    }
}

template <int CA, int CT, bool RS, int MC = M>
__global__ __launch_bounds__(WAVE) void k_reset(int A, int T, int PA, int PT, KP P, unsigned char* state, const uint64_t* seeds,
                                               double* summary, uint16_t* ablog, uint32_t mode, const int32_t* sizes,
                                               unsigned char* gscr) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    Sim<CA, CT, RS, false, MC> S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = gscr + (size_t)e * L.scratch_bytes();
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    typename Sim<CA, CT, RS, false, MC>::XY xy;
    S.load_record(rec, lane, xy);
    WSYNC();
    S.set_ablog(ablog, e, S.BA(A), S.BT(T), lane);
    S.set_retlog(nullptr, 0, e, lane);
    HdrRegs h = load_hdr(smem);
    h.seed = seeds[e]; h.d = 0;
    if (lane == 0) ((Hdr*)smem)->episodes = 0;
    const bool bad_instance = h.flags & DCM_FLAG_BAD_INSTANCE;                // set by dcm_load_instances, survives resets
    S.reset_state(h, lane);
    if (lane < 8) summary[(size_t)e * 8 + lane] = __builtin_nan("");
    PH_DECL;
    if (bad_instance) h.flags = DCM_FLAG_DONE | DCM_FLAG_BAD_INSTANCE;
    else S.advance(h, P, lane, summary + (size_t)e * 8 PH_PASS, (mode & DCM_PARAM_NO_GROUPING) != 0);
    WSYNC();
    store_hdr(h, lane);
    WSYNC();
    copy16(rec, smem, L.mut_bytes(), lane);
}

template <int CA, int CT, bool RS, int MC = M>
__global__ __launch_bounds__(WAVE) void k_observe(int A, int T, int PA, int PT, unsigned char* state, float* agents_out, float* tasks_out,
                                                 uint8_t* mask_out, int32_t* leader_out, uint8_t* active_out,
                                                 const int32_t* leader_in, const int32_t* sizes, uint32_t mode) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    using SimT = Sim<CA, CT, RS, false, MC>;
    SimT S{eA, eT, PA, PT, smem, nullptr};   // (observe never reaches the terminal metrics: no scratch)
    const Lay L = S.L();
    const int BA = S.BA(A), BT = S.BT(T);
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    typename SimT::XY xy;
    S.load_record(rec, lane, xy);
    WSYNC();
    HdrRegs h = load_hdr(smem);
    float* ag = agents_out ? agents_out + (size_t)e * 6 * BA : nullptr;
    float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (BT + 1) : nullptr;
    uint8_t* mk = mask_out ? mask_out + (size_t)e * (BT + 1) : nullptr;
    int leader = -1;
    const uint32_t flags0 = h.flags;
    if (!(h.flags & DCM_FLAG_DONE)) {
        typename SimT::AMask gm;
        leader = S.pick_leader(h, lane, leader_in ? leader_in[e] : -1, key1(h.seed, h.d), gm, (mode & DCM_PARAM_NO_GROUPING) != 0);
    }
    if (leader >= 0) S.observe(h, lane, leader, ag, tk, mk, xy);
    else S.write_inactive_obs(lane, ag, tk, mk);
    if constexpr (RS || CA == 0) S.write_pad_obs(lane, BA, BT, ag, tk, mk);
    if (lane == 0) {
        if (leader_out) leader_out[e] = leader;
        if (active_out) active_out[e] = leader >= 0 ? 1 : 0;
        if (h.flags != flags0) ((Hdr*)rec)->flags = h.flags;  // injected-leader error freezes the env
    }
}

template <int CA, int CT, bool RS, int MC = M>
__global__ __launch_bounds__(WAVE) void k_step(int A, int T, int PA, int PT, KP P, unsigned char* state, const int32_t* actions,
                                              const int32_t* leader_in, const int32_t* nfol_in, const int16_t* fol_in,
                                              float* agents_out, float* tasks_out, uint8_t* mask_out,
                                              int32_t* leader_out, uint8_t* active_out, double* summary, RouteLog log,
                                              uint16_t* ablog, uint32_t mode, const int32_t* sizes, unsigned char* gscr,
                                              uint32_t max_episodes, double* retlog, int retcap) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    using SimT = Sim<CA, CT, RS, false, MC>;
    SimT S{eA, eT, PA, PT, smem, nullptr};
    using AMask = typename SimT::AMask;
    const Lay L = S.L();
    S.scr = gscr + (size_t)e * L.scratch_bytes();
    const int BA = S.BA(A), BT = S.BT(T);
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    PHK_DECL;
    // the host's per-env inputs are requested first, so that their memory latency hides behind the record copy instead of
    // being paid at their first use in the middle of the step (phase profile: ~1 us of every wave's critical path)
    const int act_in = actions[e];
    const int lead_in = leader_in ? leader_in[e] : -1;
    const int nf = nfol_in ? nfol_in[e] : -1;
        // (plain loads, not the non-temporal ones of the persistent kernel: with the record read AND rewritten every launch the
    //  default L2 policy measured 5.5 % faster at 65 536 envs, same at 4096)
    typename SimT::XY xy;
    S.template load_record<false>(rec, lane, xy);
    S.set_ablog(ablog, e, BA, BT, lane);
    S.set_retlog(retlog, retcap, e, lane);
    if (lane == 0) { *S.dirty() = 0; *S.dirty2() = 0; }
    WSYNC();
    HdrRegs h = load_hdr(smem);
    PHK_MARK(0);                                   // record HBM -> LDS (issue + wait)
    const bool was_active = !(h.flags & DCM_FLAG_DONE);
    if (was_active) {
        AMask gm;
        const uint64_t k1 = key1(h.seed, h.d);
        const int leader = S.pick_leader(h, lane, lead_in, k1, gm, (mode & DCM_PARAM_NO_GROUPING) != 0);
        PHK_MARK(1);                               // key + leader
        if (leader >= 0) {
            PH_DECL;
            S.apply_and_advance(h, P, lane, leader, gm, act_in, k1, nf,
                                fol_in ? fol_in + (size_t)e * DCM_FOLLOWER_COLS : nullptr, summary + (size_t)e * 8 PH_PASS,
                                log, e * BA, (mode & DCM_PARAM_NO_GROUPING) != 0, (mode & DCM_PARAM_STRICT_MASK) ? 2 : 1, false, true, &xy);
            PHK_MARK(2);                           // apply + updates + advance (+ terminal)
            PHK_INNER();
            // DCM_PARAM_AUTO_RESET: the episode has just ended (its results are in the summary row) -> start the next one from
            // the loaded instance, as k_rollout_random does between its episodes (the decision counter keeps running)
            if ((mode & DCM_PARAM_AUTO_RESET) && (h.flags & DCM_FLAG_DONE) &&
                !(h.flags & (DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER | DCM_FLAG_BAD_INSTANCE)) &&
                (max_episodes == 0 || uni(((const Hdr*)smem)->episodes) < max_episodes)) {
                if (log.len) for (int a = lane; a < eA; a += WAVE) log.len[(size_t)e * BA + a] = 0;
                S.reset_state(h, lane);
                if (lane == 0) *S.dirty() = SimT::DIRTY_ALL;
                S.advance(h, P, lane, summary + (size_t)e * 8 PH_PASS, (mode & DCM_PARAM_NO_GROUPING) != 0);
                PHK_MARK(3);                       // auto-reset: reset_state + first event
            }
        }
    }
    if (was_active) {
        WSYNC();
        store_hdr(h, lane);
        WSYNC();
        // Write back what this step can have changed: the header and the agent arrays always, status words always, and of
        // the other task sections only those marked dirty (one decision typically touches one member-arrival row, the id
        // word of one task and -- when a task became feasible -- the two time arrays: ~2.4 of the 4.6 KB at 20A/50T).
        // Ranges are widened to 16-byte boundaries; the bytes around them are unchanged copies of what HBM already holds.
        const uint32_t dm = uni(*S.dirty());
        const bool big = gridDim.x >= 8192u;           // far more state than the L2s hold: stream the stores (copy16_nt)
        auto put = [&](uint32_t lo, uint32_t hi) {     // [lo, hi) of the record
            lo &= ~15u; hi = (hi + 15u) & ~15u;
            if (big) copy16_nt(rec + lo, smem + lo, hi - lo, lane); else copy16(rec + lo, smem + lo, hi - lo, lane);
        };
        const uint32_t Tn = (uint32_t)S.PT();
        put(0, L.tb());                                                               // header + agent arrays
        const uint32_t d2 = uni(*S.dirty2());
        if ((dm & SimT::DIRTY_TIMES) && !(dm & SimT::DIRTY_NAB) && (d2 >> 24) == 1u) {   // one task became feasible: its pieces of the two arrays
            const uint32_t bt = d2 & 0xFFFFFFu;
            for (uint32_t sec : {L.ts(), L.tf()}) {
                const uint32_t lo = (sec + 8u * bt) & ~63u, end = sec + 8u * Tn;
                put(lo < sec ? sec : lo, lo + 64u < end ? lo + 64u : end);
            }
        } else if (dm & SimT::DIRTY_TIMES) put(L.ts(), L.marr());         // time_start, time_finish
        // a join (the only thing that dirties arrival rows / member ids without also dirtying the abandonment counts) names its
        // task: the aligned 64-byte pieces of those sections that hold it go back instead of the 8 T-byte sections
        const bool one_task = (dm & SimT::DIRTY_IDS) && !(dm & SimT::DIRTY_NAB) && (dm & SimT::DIRTY_ROWS) != SimT::DIRTY_ROWS;
        if (one_task) {
            const uint32_t jt = dm >> SimT::DIRTY_TASK_SHIFT;
            auto piece = [&](uint32_t sec) {
                const uint32_t lo = (sec + 8u * jt) & ~63u, end = sec + 8u * Tn;
                put(lo < sec ? sec : lo, lo + 64u < end ? lo + 64u : end);
            };
#pragma unroll
            for (int j = 0; j < MC; j++) if (dm & (2u << j)) piece(L.marr() + 8u * Tn * j);
            for (uint32_t w = 0; w < L.idw(); w++) piece(L.mids() + 8u * Tn * w);
        } else {
            if ((dm & SimT::DIRTY_ROWS) == SimT::DIRTY_ROWS) put(L.marr(), L.mids());
            else {
#pragma unroll
                for (int j = 0; j < MC; j++) if (dm & (2u << j)) put(L.marr() + 8u * Tn * j, L.marr() + 8u * Tn * (j + 1));
            }
            if (dm & SimT::DIRTY_IDS) put(L.mids(), L.tinfo());
        }
        put(L.tinfo(), (dm & SimT::DIRTY_NAB) ? L.mut_bytes() : L.tnab());  // status words (+ abandonment counts)
        PHK_MARK(5);                               // write-back (issue)
    }
    const bool want_obs = agents_out || tasks_out || mask_out || leader_out || active_out;
    if (want_obs) {
        WSYNC();
        float* ag = agents_out ? agents_out + (size_t)e * 6 * BA : nullptr;
        float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (BT + 1) : nullptr;
        uint8_t* mk = mask_out ? mask_out + (size_t)e * (BT + 1) : nullptr;
        int leader = -1;
        if (!(h.flags & DCM_FLAG_DONE)) { AMask gm; leader = S.pick_leader(h, lane, -1, key1(h.seed, h.d), gm, (mode & DCM_PARAM_NO_GROUPING) != 0); }
        if (leader >= 0) {
            // The observation rows are built in LDS and leave as contiguous runs.  One lane per row writing its 5 or 6 floats
            // straight to HBM is a 20/24-byte-strided store (24 partial cache lines per wave instruction, 12 instructions);
            // staged, the same bytes are 6 fully coalesced instructions.  The staging area is the member-slot section of the
            // record image (arrival rows + id words): the write-back above has already read it, observe() never does, and
            // LDS operations of a wave execute in order -- so it costs no LDS (a separate 1.5 KB buffer would cost six
            // resident workgroups per CU, which is why round 2 measured staging slower).
            // Only for grids that fill the machine several times over (the HBM-bound regime: 165 -> 158 us at 65 536 envs);
            // a single round of workgroups is latency-bound and the extra LDS round trip costs it 0.8 us of 24 (4096 envs).
            const uint32_t need = 24u * (uint32_t)S.A() + 21u * ((uint32_t)S.T() + 1u) + 16u;
            if (gridDim.x >= 8192u && L.tinfo() - L.marr() >= need) {
                float* sag = (float*)(smem + L.marr());
                float* stk = sag + 6 * S.A();
                uint8_t* smk = (uint8_t*)(stk + 5 * (S.T() + 1));
                S.observe(h, lane, leader, ag ? sag : nullptr, tk ? stk : nullptr, mk ? smk : nullptr, xy);
                WSYNC();
                if (ag) for (int i = lane; i < 6 * S.A(); i += WAVE) __builtin_nontemporal_store(sag[i], ag + i);
                if (tk) for (int i = lane; i < 5 * (S.T() + 1); i += WAVE) __builtin_nontemporal_store(stk[i], tk + i);
                if (mk) for (int i = lane; i <= S.T(); i += WAVE) __builtin_nontemporal_store(smk[i], mk + i);
            } else {
                S.observe(h, lane, leader, ag, tk, mk, xy);
            }
        } else {
            S.write_inactive_obs(lane, ag, tk, mk);
        }
        if constexpr (RS || CA == 0) S.write_pad_obs(lane, BA, BT, ag, tk, mk);
        if (lane == 0) {
            if (leader_out) leader_out[e] = leader;
            if (active_out) active_out[e] = leader >= 0 ? 1 : 0;
        }
        PHK_MARK(4);                               // next leader + observation stores (issue)
    }
    PHK_TOTAL(6);
}

// Config-2 hot path: whole episodes in one persistent launch, record resident in LDS.
// budget (per env: budget_in[e] if given, else budget_all; < 0 = unlimited): an env takes at most that many decisions in
// this launch; when the budget runs out the env stays at the decision point it has reached (a later call -- dcm_rollout_random,
// dcm_observe or dcm_step -- carries on from it) and the observation buffers hold what the last decision TAKEN saw.
// __launch_bounds__(64, 3): at least three waves per SIMD.  The one-chunk shapes need 111 VGPRs anyway (four waves per SIMD); the
// 50A/200T instantiation wants 176 -- two waves per SIMD although its LDS image (10.9 KB with the member arrival times left in
// the HBM record) would let twelve workgroups share a CU -- and with 168 (five spilled) it runs three: 8.65 -> 6.90 ms per
// 8192-env launch.  Four (128 VGPRs, 47 spilled) measured slower again (7.15 ms).
template <int CA, int CT, bool RS, int MC = M>
__global__ __launch_bounds__(WAVE, 3) void k_rollout_random(int A, int T, int PA, int PT, KP P, unsigned char* state, int episodes,
                                                        float* agents_out, float* tasks_out, uint8_t* mask_out,
                                                        int64_t* steps_out, double* summary, uint16_t* ablog,
                                                        const int32_t* sizes, int64_t budget_all, const int64_t* budget_in,
                                                        unsigned char* gscr, double* retlog, int retcap) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    using SimT = Sim<CA, CT, RS, (CT > WAVE) && !RS, MC>;   // member arrival times in the HBM record (MG) for the exact multi-chunk shapes
    SimT S{eA, eT, PA, PT, smem, nullptr};
    using AMask = typename SimT::AMask;
    const Lay L = S.L();
    S.scr = SimT::SCR_IN_LDS ? smem + L.lds_rec() : gscr + (size_t)e * L.scratch_bytes();
    const int BA = S.BA(A), BT = S.BT(T);
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    S.gm = (double*)(rec + L.marr());
    typename SimT::XY xy;
    S.template load_record<true, false>(rec, lane, xy);
    S.set_ablog(ablog, e, BA, BT, lane);
    S.set_retlog(retlog, retcap, e, lane);
    if (lane == 0) S.inc_state()[1] = -1;  // incremental task_update: nothing is known about the last call of the previous launch
    WSYNC();
    HdrRegs h = load_hdr(smem);
    float* ag = agents_out ? agents_out + (size_t)e * 6 * BA : nullptr;
    float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (BT + 1) : nullptr;
    uint8_t* mk = mask_out ? mask_out + (size_t)e * (BT + 1) : nullptr;
    if constexpr (RS || CA == 0) S.write_pad_obs(lane, BA, BT, ag, tk, mk);
    // the usual call gives all three observation buffers: say so once, so that the per-decision null checks of observe() fold
    // (wave-uniform branches otherwise, at every decision)
    const bool all_obs = agents_out && tasks_out && mask_out;
    double* row = summary + (size_t)e * 8;
    // decisions left in this launch: a 32-bit countdown is the only loop-carried counter (steps = budget - left afterwards)
    constexpr int NO_BUDGET = 0x7FFFFFFF;
    int64_t bud = budget_in ? budget_in[e] : budget_all;
    const int left0 = uni((int)((bud < 0 || bud >= NO_BUDGET) ? NO_BUDGET : bud));
    int left = left0;
    PH_DECL;
    // key_1 = mix64(seed + GAMMA (d+1)): the argument is carried and advanced by GAMMA per decision (no 64-bit multiply,
    // and neither seed nor d stay live in the loop: d = d0 + steps afterwards)
    uint64_t gd = h.seed + GAMMA * (h.d + 1);
    const uint64_t d0 = h.d;
    for (int ep = 0; ep < episodes; ep++) {
        if (h.flags & DCM_FLAG_DONE) {  // restart from the loaded instance; d keeps running
            if (h.flags & (DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER | DCM_FLAG_BAD_INSTANCE)) break;
            if (left == 0) break;       // budget spent at an episode boundary: the finished episode's results stay readable
            S.reset_state(h, lane);
            S.advance(h, P, lane, row PH_PASS);
            PH_MARK(10);
        }
        // (the budget test rides on the loop's own scalar branch; testing it between observe and the action pick instead
        //  splits the hot block and was measured 2.3 % slower)
        while (!(h.flags & DCM_FLAG_DONE) && left != 0) {
            AMask gm;
            const uint64_t k1 = mix64(gd);   // (computing the next decision's key early, under the LDS latency of apply, measured
                                             //  0.9 % SLOWER: two more live registers across the whole decision)
            const int leader = S.pick_leader(h, lane, -1, k1, gm);
            if (leader < 0) break;
            PH_MARK(0);
            if (all_obs) { __builtin_assume(ag != nullptr); __builtin_assume(tk != nullptr); __builtin_assume(mk != nullptr); S.observe(h, lane, leader, ag, tk, mk, xy); }
            else S.observe(h, lane, leader, ag, tk, mk, xy);
            PH_MARK(1);
            const int action = S.pick_random_action(lane, k1);
            PH_MARK(2);
            S.template apply_and_advance<true>(h, P, lane, leader, gm, action, k1, -1, nullptr, row PH_PASS, RouteLog{nullptr, nullptr, nullptr, 0}, 0, false, 0, true, false, &xy);
            gd += GAMMA;
            left--;
        }
        if (left == 0) break;
    }
    PH_FLUSH(lane);
    const int64_t steps = (int64_t)(left0 - left);
    if (lane == 0 && steps_out) steps_out[e] = steps;
    h.d = d0 + (uint64_t)steps;   // every decision of this kernel is valid, so apply_and_advance counted exactly `steps`
    {   // Hdr::max_arrival: this kernel only takes valid actions, under which every arrival list is monotone, so the maximum of
        // the agents' last arrivals IS the running maximum of the episode so far -- folded in once per launch for a later
        // dcm_step on the same episode
        double m = 0.0;
        S.for_agents(lane, [&](int a) { const double av = (S.cur()[a] != -2) ? S.arr()[a] : 0.0; m = av > m ? av : m; });
        const double wm = wave_nanmax(m);
        if (lane == 0) { Hdr* q = (Hdr*)smem; if (wm > q->max_arrival) q->max_arrival = wm; }
    }
    WSYNC();
    store_hdr(h, lane);
    WSYNC();
    S.store_record(rec, lane);
}

#include "rollout_fast.hpp"
#include "step_fast.hpp"
#include "rollout_fast_mc.hpp"
#if defined(DCM_TU_G) || !defined(DCM_SPLIT_G)
#include "rollout_fast_g.hpp"
#endif

__global__ __launch_bounds__(WAVE) void k_env_status(int PA, int PT, int PC, const unsigned char* state, int B, uint32_t* flags_out,
                                                    int64_t* dec_out, double* now_out, int32_t* episodes_out) {
    const int e = blockIdx.x * WAVE + threadIdx.x;
    if (e >= B) return;
    const Hdr* h = (const Hdr*)(state + (size_t)e * Lay{PA, PT, PC}.rec_bytes());
    if (episodes_out) episodes_out[e] = (int32_t)h->episodes;
    if (flags_out) flags_out[e] = h->flags;
    if (dec_out) dec_out[e] = (int64_t)h->d;
    if (now_out) now_out[e] = h->now;
}

template <int MC>
__global__ __launch_bounds__(WAVE) void k_get_tasks(int A, int T, int PA, int PT, KP P, unsigned char* state, uint8_t* finished,
                                                   uint8_t* feasible, double* time_start, double* time_finish,
                                                   double* sum_wait, int32_t* status, int32_t* n_members,
                                                   int32_t* n_abandoned, uint16_t* ablog, const int32_t* sizes,
                                                   unsigned char* gscr) {
    const int e = blockIdx.x, lane = threadIdx.x;
    int eA, eT;
    env_dims<0, 0, false>(sizes, e, A, T, eA, eT);
    Sim<0, 0, false, false, MC> S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = gscr + (size_t)e * L.scratch_bytes();
    copy16_in(smem, state + (size_t)e * L.rec_bytes(), L.rec_bytes(), lane);
    S.set_ablog(ablog, e, A, T, lane);
    WSYNC();
    HdrRegs h = load_hdr(smem);
    if (sum_wait) S.compute_waits(h.now, P.mwt, lane);
    for (int t = lane; t < T; t += WAVE) {                                  // rows t >= T_e of a ragged batch read as 0
        const size_t o = (size_t)e * T + t;
        const bool in = t < eT;
        const uint32_t info = in ? S.tinfo()[t] : 0u;
        if (finished) finished[o] = (info & T_FIN) ? 1 : 0;
        if (feasible) feasible[o] = (info & T_FEAS) ? 1 : 0;
        if (time_start) time_start[o] = in ? S.ts()[t] : 0.0;
        if (time_finish) time_finish[o] = in ? S.tf()[t] : 0.0;
        if (sum_wait) sum_wait[o] = in ? S.tw()[t] : 0.0;
        if (status) status[o] = (int)(int8_t)((info >> 8) & 0xFF);
        if (n_members) n_members[o] = (info >> 16) & 0xFF;
        if (n_abandoned) n_abandoned[o] = in ? (int32_t)S.tnab()[t] : 0;
    }
}

template <int MC>
__global__ __launch_bounds__(WAVE) void k_get_agents(int A, int T, int PA, int PT, KP P, unsigned char* state, double* sum_wait,
                                                    double* travel_dist, double* next_decision, double* arrival,
                                                    double* x, double* y, uint8_t* returned, uint8_t* assigned,
                                                    int32_t* current, int32_t* pending, uint16_t* ablog,
                                                    const int32_t* sizes, unsigned char* gscr) {
    const int e = blockIdx.x, lane = threadIdx.x;
    int eA, eT;
    env_dims<0, 0, false>(sizes, e, A, T, eA, eT);
    Sim<0, 0, false, false, MC> S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = gscr + (size_t)e * L.scratch_bytes();
    copy16_in(smem, state + (size_t)e * L.rec_bytes(), L.rec_bytes(), lane);
    S.set_ablog(ablog, e, A, T, lane);
    WSYNC();
    HdrRegs h = load_hdr(smem);
    if (sum_wait) S.compute_waits(h.now, P.mwt, lane);
    for (int a = lane; a < A; a += WAVE) {                                  // rows a >= A_e of a ragged batch: 0 / NaN timer
        const size_t o = (size_t)e * A + a;
        const bool in = a < eA;
        const uint32_t ai = in ? S.ainfo()[a] : 0u;
        if (sum_wait) sum_wait[o] = in ? S.aw()[a] : 0.0;
        if (travel_dist) travel_dist[o] = in ? S.tdist()[a] : 0.0;
        if (next_decision) next_decision[o] = in ? S.nd()[a] : __builtin_nan("");
        if (arrival) arrival[o] = in ? S.arr()[a] : 0.0;
        if (x) x[o] = in ? S.ax()[a] : 0.0;
        if (y) y[o] = in ? S.ay()[a] : 0.0;
        if (returned) returned[o] = (ai & A_RETURNED) ? 1 : 0;
        if (assigned) assigned[o] = (ai & A_ASSIGNED) ? 1 : 0;
        if (current) current[o] = in ? S.cur()[a] : -2;
        if (pending) pending[o] = (int32_t)((ai >> 8) & 0xFFu);
    }
}

// task['members'] of every task, in list order (env/task_env.py:80): ids_out[B][T][member slots of the handle], -1 padded
__global__ void k_get_members(int T, int PA, int PT, int PC, const unsigned char* state, int B, int16_t* ids_out, const int32_t* sizes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int e = (int)(i / T), t = (int)(i % T);
    const int eT = sizes ? sizes[2 * e + 1] : T;
    const Lay L{PA, PT, PC};
    const unsigned char* rec = state + (size_t)e * L.rec_bytes();
    uint64_t ids[2] = {0, 0};
    int n = 0;
    if (t < eT) {
        for (uint32_t w = 0; w < L.idw(); w++) ids[w] = ((const uint64_t*)(rec + L.mids()))[w * PT + t];
        n = (int)((((const uint32_t*)(rec + L.tinfo()))[t] >> 16) & 0xFF);
    }
    for (int j = 0; j < PC; j++) ids_out[i * PC + j] = (j < n) ? (int16_t)((ids[j >> 3] >> (8 * (j & 7))) & 0xFF) : (int16_t)-1;
}

// task['abandoned_agent'] (env/task_env.py:89) as a dense count table: out[B][A][T] = number of times task t moved agent a to
// its abandoned_agent list in the current episode = entries of the agent's abandonment log + the overflow count table
__global__ __launch_bounds__(WAVE) void k_get_abandoned(int A, int T, int PA, int PT, int PC, const unsigned char* state,
                                                       const uint16_t* ablog, uint16_t* out, const int32_t* sizes) {
    const int e = blockIdx.x, lane = threadIdx.x, B = gridDim.x;
    int eA, eT;
    env_dims<0, 0, false>(sizes, e, A, T, eA, eT);
    const Lay L{PA, PT, PC};
    const uint32_t* ainfo = (const uint32_t*)(state + (size_t)e * L.rec_bytes() + L.ainfo());
    const uint16_t* log_e = ablog + (size_t)e * A * AB_CAP;
    const uint16_t* cnt_e = (const uint16_t*)((const uint8_t*)(ablog + (size_t)B * A * AB_CAP) + (size_t)e * abcnt_pitch(A, T));
    uint16_t* o = out + (size_t)e * A * T;
    for (int i = lane; i < A * T; i += WAVE) o[i] = 0;
    __syncthreads();
    for (int a = lane; a < eA; a += WAVE) {                                 // one lane per agent: no two lanes share a row
        const uint32_t nab = ainfo[a] >> 16;
        const int nl = nab < (uint32_t)AB_CAP ? (int)nab : AB_CAP;
        for (int i = 0; i < nl; i++) o[a * T + log_e[a * AB_CAP + i]] += 1;
        if (nab > (uint32_t)AB_CAP)
            for (int t = 0; t < eT; t++) o[a * T + t] += cnt_e[a * eT + t];
    }
}

__global__ void k_distance(const double* ax, const double* ay, const double* bx, const double* by, double* dist_out,
                           double* time_out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = dist2(ax[i], ay[i], bx[i], by[i]);
    if (dist_out) dist_out[i] = d;
    if (time_out) time_out[i] = over_velocity(d);
}

// ---------------------------------------------------------------------------------- host side
// Instantiations: the three BASELINE shapes exactly; <20,50,runtime sizes> for every other shape (uniform or ragged) inside
// the reference's training range A <= 20, T <= 50 (parameters.py:15-16); <64,64,runtime sizes> for the remaining one-chunk
// shapes (A <= 64, T <= 64: one lane per agent / task); <128,256,runtime sizes and layout> for the mid sizes (generate_env
// takes any size, env/task_env.py:57-65); <0,0> for the rest.
#define FOR_EACH_INSTANCE(X) X(20, 50, false); X(20, 50, true); X(64, 64, true); X(50, 200, false); X(100, 500, false); X(128, 256, true); X(0, 0, false)
// one-chunk layouts: the register-resident persistent kernel (rollout_fast.hpp)
#define FOR_EACH_FAST(X) X(20, 50, false); X(20, 50, true); X(64, 64, true)
#define DISPATCH_ENV(env, CALL)                                                                        \
    do {                                                                                               \
        const dcm_env* e_ = (env);                                                                     \
        const bool exact_ = !e_->sizes && e_->A == e_->L.A && e_->T == e_->L.T;                        \
        if (e_->L.C > M) { CALL(0, 0, false, MW); }   /* DCM_PARAM_WIDE_MEMBERS: sixteen member slots (two id words), runtime-size code */ \
        else if (e_->L.A == 20 && e_->L.T == 50) { if (exact_) { CALL(20, 50, false); } else { CALL(20, 50, true); } } \
        else if (e_->L.A == 64 && e_->L.T == 64) { CALL(64, 64, true); }                               \
        else if (exact_ && e_->A == 50 && e_->T == 200) { CALL(50, 200, false); }                      \
        else if (exact_ && e_->A == 100 && e_->T == 500) { CALL(100, 500, false); }                    \
        else if (e_->A <= 128 && e_->T <= 256) { CALL(128, 256, true); }   /* mid sizes: bounded trip counts, own layout */ \
        else { CALL(0, 0, false); }                                                                    \
    } while (0)
// kernel arguments every env kernel starts with: batch dims, layout dims
#define DIMS(env) (env)->A, (env)->T, (env)->L.A, (env)->L.T

// dcm_create / dcm_destroy run on the handle's device but leave the caller's current device as they found it
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

#if defined(DCM_TU_G) || !defined(DCM_SPLIT_G)
void dcm::launch_rollout_fast_g(int nac, int ntc, bool obs, unsigned grid, unsigned lds_bytes, hipStream_t stream, int A, int T, int PA, int PT,
                                KP kp, unsigned char* state, int episodes, float* agents_out, float* tasks_out, uint8_t* mask_out,
                                int64_t* steps_out, double* summary, uint16_t* ablog, const int32_t* sizes, int64_t budget_all,
                                const int64_t* budget_in, unsigned char* gscr, double* retlog, int retcap) {
#define CALLG(NAC, NTC, OBS)                                                                                           \
    hipLaunchKernelGGL((k_rollout_fast_g<NAC, NTC, OBS>), dim3(grid), dim3(WAVE), lds_bytes, stream, A, T, PA, PT, kp, state, episodes, \
                       agents_out, tasks_out, mask_out, steps_out, summary, ablog, sizes, budget_all, budget_in, gscr, retlog, retcap)
#define CALLT(NAC, OBS) do { if (ntc > 3) { CALLG(NAC, 4, OBS); } else if (ntc > 2) { CALLG(NAC, 3, OBS); } else { CALLG(NAC, 2, OBS); } } while (0)
#define CALLA(OBS) do { if (nac > 1) { CALLT(2, OBS); } else { CALLT(1, OBS); } } while (0)
    if (obs) { CALLA(true); } else { CALLA(false); }
#undef CALLA
#undef CALLT
#undef CALLG
}
#endif

#ifndef DCM_DEVICE_ONLY_TU   // (tools/loop_insts.py compiles single kernel instantiations of this file without the host API)
extern "C" {

const char* dcm_last_error(void) { return dcm::g_err; }
int dcm_abi_version(void) { return DCM_ABI_VERSION; }
#ifndef DCM_BUILD_ID
#define DCM_BUILD_ID "unknown"
#endif
const char* dcm_build_id(void) { return DCM_BUILD_ID; }
#ifdef DCM_COUNT_PATHS
// developer build only (tools/path_counts.py): read and clear the dynamic path counts of the register-resident rollout kernel
int dcm_debug_path_counts(unsigned long long* out32) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_path_counts), 32 * sizeof(unsigned long long)));
    static const unsigned long long zero[32] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_path_counts), zero, sizeof(zero)));
    return 0;
}
#endif

int dcm_create(const dcm_params* params, dcm_env** out) {
    if (!params || !out) return fail(DCM_ERR_INVALID, "dcm_create: null argument");
    *out = nullptr;
    if (params->n_envs < 1 || params->n_agents < 1 || params->n_agents > DCM_MAX_AGENTS || params->n_tasks < 1 ||
        params->n_tasks > DCM_MAX_TASKS)
        return fail(DCM_ERR_INVALID, "dcm_create: need 1<=A<=128, 1<=T<=1023, B>=1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(DCM_ERR_NO_DEVICE, "dcm_create: no HIP device (this library has no CPU path)");
    if (params->device < 0 || params->device >= ndev) return fail(DCM_ERR_INVALID, "dcm_create: bad device ordinal");
    DeviceGuard guard(params->device);
    dcm_env* h = new (std::nothrow) dcm_env();
    if (!h) return fail(DCM_ERR_INVALID, "dcm_create: out of host memory");
    h->p = *params;
    h->A = params->n_agents; h->T = params->n_tasks;
    h->L = (params->flags & DCM_PARAM_WIDE_MEMBERS) ? Lay{h->A, h->T, MW}
           : (h->A <= 20 && h->T <= 50) ? Lay{20, 50} : (h->A <= 64 && h->T <= 64) ? Lay{64, 64} : Lay{h->A, h->T};
    h->kp.mwt = params->max_waiting_time;
    h->kp.max_time = params->max_time;
    if (h->L.lds_rec() + align16(4u * (uint32_t)h->L.T) > 160 * 1024) { delete h; return fail(DCM_ERR_INVALID, "dcm_create: env record does not fit the 160 KiB LDS"); }
    const size_t bytes = (size_t)params->n_envs * h->L.rec_bytes();
    hipError_t e1 = hipMalloc((void**)&h->state, bytes);
    hipError_t e2 = hipMalloc((void**)&h->summary, (size_t)params->n_envs * 8 * sizeof(double));
    if (e1 == hipSuccess && e2 == hipSuccess)
        e2 = hipMalloc((void**)&h->ablog, side_bytes(params->n_envs, params->n_agents, params->n_tasks));
    if (e1 == hipSuccess && e2 == hipSuccess)
        e2 = hipMalloc((void**)&h->gscratch, (size_t)params->n_envs * h->L.scratch_bytes());
    if (e1 != hipSuccess || e2 != hipSuccess) {
        if (h->state) (void)hipFree(h->state);
        if (h->summary) (void)hipFree(h->summary);
        if (h->ablog) (void)hipFree(h->ablog);
        if (h->gscratch) (void)hipFree(h->gscratch);
        delete h;
        return fail(DCM_ERR_HIP, "dcm_create: hipMalloc failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    hipError_t e3 = hipMemset(h->state, 0, bytes);
    if (e3 == hipSuccess) e3 = hipMemset(h->ablog, 0, side_bytes(params->n_envs, params->n_agents, params->n_tasks));
    if (e3 != hipSuccess) { (void)hipFree(h->state); (void)hipFree(h->summary); (void)hipFree(h->ablog); (void)hipFree(h->gscratch); delete h; return fail(DCM_ERR_HIP, "hipMemset: %s", hipGetErrorString(e3)); }
    // kernels that keep the record in LDS may need more than the default 64 KiB of dynamic LDS.  The limit is a
    // per-function, per-device attribute shared by every handle of the process, so it only ever grows: a later, smaller
    // env must not lower it under an earlier handle's launches.
    static int lds_limit[64] = {0};
    static std::mutex lds_mutex;                 // handles are created from several host threads (one actor thread per GPU)
    std::lock_guard<std::mutex> lds_guard(lds_mutex);
    const int dev_slot = params->device & 63;
    int lds = (int)(h->L.lds_rec() + align16(4u * (uint32_t)h->L.T));   // (+ the wake-up times of the multi-chunk layouts)
    if (h->L.lds_bytes() <= 10240) lds = (int)h->L.lds_bytes();           // persistent kernel of the small layouts: scratch in LDS
    if (lds + 512 <= 160 * 1024) lds += 512;                               // the register-resident kernels' dummy slots (small layouts)
    // (only when the limit really grows: the attribute calls of one host thread must not keep landing in another thread's
    //  stream capture -- actors create handles for new batch shapes while others capture)
    const bool grow = lds > lds_limit[dev_slot];
    if (!grow) { *out = h; return DCM_OK; }
    lds_limit[dev_slot] = lds;
#define SET_ATTR(CA, CT, RS)                                                                                             \
    (void)hipFuncSetAttribute((const void*)k_reset<CA, CT, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);        \
    (void)hipFuncSetAttribute((const void*)k_observe<CA, CT, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);      \
    (void)hipFuncSetAttribute((const void*)k_step<CA, CT, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);         \
    (void)hipFuncSetAttribute((const void*)k_rollout_random<CA, CT, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
    FOR_EACH_INSTANCE(SET_ATTR);
#undef SET_ATTR
    (void)hipFuncSetAttribute((const void*)k_reset<0, 0, false, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_observe<0, 0, false, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_step<0, 0, false, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_rollout_random<0, 0, false, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#define SET_FAST(CA, CT, RS)                                                                                              \
    (void)hipFuncSetAttribute((const void*)k_rollout_fast<CA, CT, RS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
    (void)hipFuncSetAttribute((const void*)k_rollout_fast<CA, CT, RS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    (void)hipFuncSetAttribute((const void*)k_rollout_fast<CA, CT, RS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
    (void)hipFuncSetAttribute((const void*)k_rollout_fast<CA, CT, RS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    (void)hipFuncSetAttribute((const void*)k_step_fast<CA, CT, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
    FOR_EACH_FAST(SET_FAST);
#undef SET_FAST
    (void)hipFuncSetAttribute((const void*)k_rollout_fast_mc<50, 200, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_rollout_fast_mc<50, 200, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_get_tasks<M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_get_agents<M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_get_tasks<MW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k_get_agents<MW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    *out = h;
    return DCM_OK;
}

int dcm_destroy(dcm_env* env) {
    if (!env) return DCM_OK;
    DeviceGuard guard(env->p.device);
    if (env->state) (void)hipFree(env->state);
    if (env->summary) (void)hipFree(env->summary);
    if (env->ablog) (void)hipFree(env->ablog);
    if (env->gscratch) (void)hipFree(env->gscratch);
    if (env->routes) (void)hipFree(env->routes);
    if (env->route_len) (void)hipFree(env->route_len);
    if (env->rmarr) (void)hipFree(env->rmarr);
    if (env->sizes) (void)hipFree(env->sizes);
    if (env->side) (void)hipFree(env->side);
    if (env->pendq) (void)hipFree(env->pendq);
    if (env->init) (void)hipFree(env->init);
    delete env;
    return DCM_OK;
}


int dcm_load_instances(dcm_env* env, const double* depot, const double* task_xy, const int32_t* req, const double* dur,
                       void* stream) {
    CHECK_ENV(env);
    if (!depot || !task_xy || !req || !dur) return fail(DCM_ERR_INVALID, "dcm_load_instances: null array");
    { const int rc_ = dcm::drop_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    if (env->sizes) { HIP_TRY(hipFree(env->sizes)); env->sizes = nullptr; }   // back to a uniform batch (hipFree synchronises)
    hipLaunchKernelGGL(k_load_instances, GRID(env), 0, (hipStream_t)stream, DIMS(env), env->L.C, env->state, depot, task_xy,
                       req, dur, (const int32_t*)nullptr);
    LAUNCH_OK();
    env->loaded = true;
    env->reset_done = false;
    return DCM_OK;
}

int dcm_load_instances_ragged(dcm_env* env, const double* depot, const double* task_xy, const int32_t* req,
                              const double* dur, const int32_t* n_agents_host, const int32_t* n_tasks_host, void* stream) {
    CHECK_ENV(env);
    { const int rc_ = dcm::drop_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    if (!depot || !task_xy || !req || !dur || !n_agents_host || !n_tasks_host)
        return fail(DCM_ERR_INVALID, "dcm_load_instances_ragged: null array");
    const int B = env->p.n_envs;
    env->sizes_host.resize((size_t)2 * B);
    for (int e = 0; e < B; e++) {
        const int a = n_agents_host[e], t = n_tasks_host[e];
        if (a < 1 || a > env->A || t < 1 || t > env->T)
            return fail(DCM_ERR_INVALID, "dcm_load_instances_ragged: need 1 <= n_agents[e] <= A and 1 <= n_tasks[e] <= T");
        env->sizes_host[2 * (size_t)e] = a; env->sizes_host[2 * (size_t)e + 1] = t;
    }
    if (!env->sizes) HIP_TRY(hipMalloc((void**)&env->sizes, (size_t)2 * B * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(env->sizes, env->sizes_host.data(), (size_t)2 * B * sizeof(int32_t), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
    hipLaunchKernelGGL(k_load_instances, GRID(env), 0, (hipStream_t)stream, DIMS(env), env->L.C, env->state, depot, task_xy,
                       req, dur, (const int32_t*)env->sizes);
    LAUNCH_OK();
    env->loaded = true;
    env->reset_done = false;
    return DCM_OK;
}

}  // extern "C"

namespace dcm {
int flush_pending(dcm_env* env, void* stream) {
    if (!env->maybe_pending) return DCM_OK;
#define CALL(CA, CT, RS)                                                                                             \
    hipLaunchKernelGGL((k_terminal_flush<CA, CT, RS>), GRID(env),                                    \
                       (Sim<CA, CT, RS>::lds_image_bytes(env->L)) + 512u + (step_scratch_in_lds<CA, CT>() ? env->L.scratch_bytes() : 0u), \
                       (hipStream_t)stream, DIMS(env), env->kp, (const unsigned char*)env->side, env->side_pitch, env->pendq,       \
                       env->summary, (const int32_t*)env->sizes, env->gscratch)
    const bool exact_ = !env->sizes && env->A == env->L.A && env->T == env->L.T;
    if (env->L.A == 20 && env->L.T == 50) { if (exact_) { CALL(20, 50, false); } else { CALL(20, 50, true); } }
    else { CALL(64, 64, true); }
#undef CALL
    LAUNCH_OK();
    env->maybe_pending = false;
    env->steps_since_flush = 0;
    return DCM_OK;
}
int drop_pending(dcm_env* env, void* stream) {
    if (!env->maybe_pending) return DCM_OK;
    HIP_TRY(hipMemsetAsync(env->pendq, 0, (size_t)env->p.n_envs * sizeof(uint32_t), (hipStream_t)stream));
    env->maybe_pending = false;
    env->steps_since_flush = 0;
    return DCM_OK;
}
}  // namespace dcm

extern "C" {

int dcm_reset(dcm_env* env, const uint64_t* seeds, void* stream) {
    CHECK_ENV(env);
    { const int rc_ = dcm::drop_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    if (!env->loaded) return fail(DCM_ERR_STATE, "dcm_reset: call dcm_load_instances first");
    if (!seeds) return fail(DCM_ERR_INVALID, "dcm_reset: null seeds");
#define CALL(CA, CT, RS, ...)                                                                                         \
    hipLaunchKernelGGL((k_reset<CA, CT, RS, ##__VA_ARGS__>), GRID(env), (Sim<CA, CT, RS>::lds_image_bytes(env->L)), (hipStream_t)stream, DIMS(env), env->kp, \
                       env->state, seeds, env->summary, env->ablog, env->p.flags, (const int32_t*)env->sizes, env->gscratch)
    DISPATCH_ENV(env, CALL);
#undef CALL
    LAUNCH_OK();
    if (env->log.len)
        HIP_TRY(hipMemsetAsync(env->log.len, 0, (size_t)env->p.n_envs * env->A * sizeof(int32_t), (hipStream_t)stream));
    env->reset_done = true;
    // the restart image of the register-resident lockstep kernel (dcm_env::init): what this reset produced
    env->init_valid = false;
    // (max_time <= 0: the first event already ends the episode -- summary row, episode count, return log -- which a copied image cannot redo)
    if ((env->p.flags & DCM_PARAM_AUTO_RESET) && env->L.C == M && env->L.A <= 64 && env->L.T <= 64 && !env->init_failed && env->kp.max_time > 0.0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
        const size_t sb = (size_t)env->p.n_envs * env->L.rec_bytes();
        if (!env->init && !capturing && hipMalloc((void**)&env->init, sb) != hipSuccess) {
            (void)hipGetLastError();                                         // out of memory: the kernel restarts the long way
            env->init = nullptr;
            env->init_failed = true;
        }
        if (env->init && !capturing) {
            HIP_TRY(hipMemcpyAsync(env->init, env->state, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
            env->init_valid = true;
        }
    }
    return DCM_OK;
}

int dcm_set_route_log(dcm_env* env, int16_t* route_task, double* route_arrival, int32_t* route_len, int32_t cap) {
    CHECK_HANDLE(env);
    const bool off = !route_task && !route_arrival && !route_len;
    if (!off && (!route_task || !route_arrival || !route_len || cap < 1))
        return fail(DCM_ERR_INVALID, "dcm_set_route_log: give all three arrays and cap >= 1, or all NULL");
    env->log = RouteLog{route_task, route_arrival, route_len, off ? 0 : cap};
    return DCM_OK;
}

int dcm_set_return_log(dcm_env* env, double* returns, int32_t cap) {
    CHECK_HANDLE(env);
    if ((returns == nullptr) != (cap == 0) || cap < 0)
        return fail(DCM_ERR_INVALID, "dcm_set_return_log: give a buffer and cap >= 1, or NULL and 0");
    env->retlog = returns; env->retcap = cap;
    return DCM_OK;
}

int dcm_observe(dcm_env* env, float* agents_out, float* tasks_out, uint8_t* mask_out, int32_t* leader_out,
                uint8_t* active_out, const int32_t* leader_in, void* stream) {
    CHECK_ENV(env);
    if (!env->reset_done) return fail(DCM_ERR_STATE, "dcm_observe: call dcm_reset first");
#define CALL(CA, CT, RS, ...)                                                                                           \
    hipLaunchKernelGGL((k_observe<CA, CT, RS, ##__VA_ARGS__>), GRID(env), (Sim<CA, CT, RS>::lds_image_bytes(env->L)), (hipStream_t)stream, DIMS(env),            \
                       env->state, agents_out, tasks_out, mask_out, leader_out, active_out, leader_in,                  \
                       (const int32_t*)env->sizes, env->p.flags)
    DISPATCH_ENV(env, CALL);
#undef CALL
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_step(dcm_env* env, const int32_t* actions, const int32_t* leader_in, const int32_t* nfol_in,
             const int16_t* followers_in, float* agents_out, float* tasks_out, uint8_t* mask_out, int32_t* leader_out,
             uint8_t* active_out, void* stream) {
    CHECK_ENV(env);
    if (!env->reset_done) return fail(DCM_ERR_STATE, "dcm_step: call dcm_reset first");
    if (!actions) return fail(DCM_ERR_INVALID, "dcm_step: null actions");
    if ((nfol_in == nullptr) != (followers_in == nullptr))
        return fail(DCM_ERR_INVALID, "dcm_step: nfol_in and followers_in must be given together");
    // The register-resident kernels skip the task_update pass of a QUIET join on the ground that a member who has just joined
    // has not waited max_waiting_time yet (env/task_env.py:269), which needs max_waiting_time > 0 (the reference's 10 / 100):
    // a handle with max_waiting_time <= 0 (or NaN) takes the general kernels, which evaluate the rule literally.
    const bool quiet_ok = env->kp.mwt > 0.0;
#ifndef DCM_NO_FAST_STEP
    // The plain call shape on a one-chunk layout: the register-resident step (step_fast.hpp); same contract, same results.
    if (quiet_ok && env->L.C == M && env->L.A <= 64 && env->L.T <= 64 && env->T <= 63 && !leader_in && !nfol_in && !env->log.len && agents_out && tasks_out &&
        mask_out && leader_out && active_out && !(env->p.flags & DCM_PARAM_NO_GROUPING)) {
        // Deferred terminal metrics (dcm_env::side) for auto-resetting handles: not under stream capture -- the periodic flush is a
        // host-side decision and the buffers are allocated on first use -- where episode ends keep computing their metrics inline.
        uint32_t* pendq = nullptr;
        if (env->p.flags & DCM_PARAM_AUTO_RESET) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
            if (capturing) {
                if (env->maybe_pending)
                    return fail(DCM_ERR_STATE, "dcm_step under stream capture: deferred episode summaries are waiting; call dcm_summary (or dcm_reset) before the capture");
                env->captured = true;      // a replay of this graph writes summary rows unseen by the host: no snapshots on this handle any more
            } else if (!env->captured) {
                if (!env->side && !env->side_failed) {
                    const uint32_t pitch = dcm::align16(env->L.rec_bytes() + (uint32_t)env->L.A * dcm::AB_CAP * (uint32_t)sizeof(uint16_t));
                    const size_t qn = (size_t)env->p.n_envs;
                    if (hipMalloc((void**)&env->side, (size_t)env->p.n_envs * pitch) != hipSuccess ||
                        hipMalloc((void**)&env->pendq, qn * sizeof(uint32_t)) != hipSuccess ||
                        hipMemset(env->pendq, 0, qn * sizeof(uint32_t)) != hipSuccess) {
                        (void)hipGetLastError();                         // out of memory: keep the inline terminal metrics
                        if (env->side) { (void)hipFree(env->side); env->side = nullptr; }
                        if (env->pendq) { (void)hipFree(env->pendq); env->pendq = nullptr; }
                        env->side_failed = true;
                    } else env->side_pitch = pitch;
                }
                pendq = env->pendq;
            }
        }
#define CALL(CA, CT, RS)                                                                                             \
    hipLaunchKernelGGL((k_step_fast<CA, CT, RS>), GRID(env),                                                           \
                       (Sim<CA, CT, RS>::lds_image_bytes(env->L)) + 512u + (step_scratch_in_lds<CA, CT>() ? env->L.scratch_bytes() : 0u), \
                       (hipStream_t)stream, DIMS(env), env->kp,                                                            \
                       env->state, actions, agents_out, tasks_out, mask_out, leader_out, active_out, env->summary, env->ablog,  \
                       env->p.flags, (const int32_t*)env->sizes, env->gscratch, env->p.auto_reset_episodes, env->retlog, (int)env->retcap, \
                       env->side, env->side_pitch, pendq, (pendq && env->init_valid) ? (const unsigned char*)env->init : nullptr)
        const bool exact_ = !env->sizes && env->A == env->L.A && env->T == env->L.T;
        if (env->L.A == 20 && env->L.T == 50) { if (exact_) { CALL(20, 50, false); } else { CALL(20, 50, true); } }
        else { CALL(64, 64, true); }
#undef CALL
        LAUNCH_OK();
        if (pendq) {
            env->maybe_pending = true;
            if (++env->steps_since_flush >= dcm_env::FLUSH_EVERY) return dcm::flush_pending(env, stream);
        }
        return DCM_OK;
    }
#endif
    { const int rc_ = dcm::flush_pending(env, stream); if (rc_ != DCM_OK) return rc_; }    // (the general kernel writes summary rows itself)
#define CALL(CA, CT, RS, ...)                                                                                        \
    hipLaunchKernelGGL((k_step<CA, CT, RS, ##__VA_ARGS__>), GRID(env), (Sim<CA, CT, RS>::lds_image_bytes(env->L)), (hipStream_t)stream, DIMS(env), env->kp,   \
                       env->state, actions, leader_in, nfol_in, followers_in, agents_out, tasks_out, mask_out, leader_out, \
                       active_out, env->summary, env->log, env->ablog, env->p.flags, (const int32_t*)env->sizes, env->gscratch, \
                       env->p.auto_reset_episodes, env->retlog, (int)env->retcap)
    DISPATCH_ENV(env, CALL);
#undef CALL
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_rollout_random(dcm_env* env, int32_t episodes, int64_t max_decisions, const int64_t* max_decisions_in,
                       float* agents_out, float* tasks_out, uint8_t* mask_out, int64_t* steps_out, void* stream) {
    CHECK_ENV(env);
    if (!env->reset_done) return fail(DCM_ERR_STATE, "dcm_rollout_random: call dcm_reset first");
    if (episodes < 1) return fail(DCM_ERR_INVALID, "dcm_rollout_random: episodes must be >= 1");
    { const int rc_ = dcm::flush_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    const bool quiet_ok = env->kp.mwt > 0.0;     // (see dcm_step)
#ifndef DCM_NO_FAST_ROLLOUT
    // One-chunk layouts (one lane per agent and per task, lane 63 free for the depot) with all three observation buffers or
    // none: the register-resident kernel.  Same contract, same results (tests/test_gpu_rollout.py runs both).
    const bool all_obs = agents_out && tasks_out && mask_out, no_obs = !agents_out && !tasks_out && !mask_out;
    if (quiet_ok && env->L.C == M && env->L.A <= 64 && env->L.T <= 64 && env->T <= 63 && (all_obs || no_obs)) {
#define CALLF(CA, CT, RS, OBS, PRIO)                                                                                  \
    hipLaunchKernelGGL((k_rollout_fast<CA, CT, RS, OBS, PRIO>), GRID(env),                                            \
                       (Sim<CA, CT, RS>::SCR_IN_LDS ? env->L.lds_bytes() : Sim<CA, CT, RS>::lds_image_bytes(env->L)) + 512u, (hipStream_t)stream, DIMS(env), \
                       env->kp, env->state, (int)episodes, agents_out, tasks_out, mask_out, steps_out, env->summary, env->ablog, \
                       (const int32_t*)env->sizes, max_decisions, max_decisions_in, env->gscratch, env->retlog, (int)env->retcap)
        // wave priorities (k_rollout_fast, PRIO) for a launch that fills the machine by itself: 16 workgroups x 256 CUs
        const bool prio = env->p.n_envs >= 4096;
#define CALL(CA, CT, RS) do { if (prio) { if (all_obs) { CALLF(CA, CT, RS, true, true); } else { CALLF(CA, CT, RS, false, true); } }   \
                              else { if (all_obs) { CALLF(CA, CT, RS, true, false); } else { CALLF(CA, CT, RS, false, false); } } } while (0)
        const bool exact_ = !env->sizes && env->A == env->L.A && env->T == env->L.T;
        if (env->L.A == 20 && env->L.T == 50) { if (exact_) { CALL(20, 50, false); } else { CALL(20, 50, true); } }
        else { CALL(64, 64, true); }
#undef CALL
#undef CALLF
        LAUNCH_OK();
        return DCM_OK;
    }
    // BASELINE configs[3], 50A/200T exactly: the multi-chunk register-resident kernel (rollout_fast_mc.hpp)
    if (quiet_ok && env->L.C == M && !env->sizes && env->A == 50 && env->T == 200 && env->L.A == 50 && env->L.T == 200 && (all_obs || no_obs)) {
#define CALLM(OBS)                                                                                                    \
    hipLaunchKernelGGL((k_rollout_fast_mc<50, 200, OBS>), GRID(env), (FastM<50, 200, OBS>::LDS_BYTES), (hipStream_t)stream, \
                       DIMS(env), env->kp, env->state, (int)episodes, agents_out, tasks_out, mask_out, steps_out, env->summary, env->ablog, \
                       (const int32_t*)env->sizes, max_decisions, max_decisions_in, env->gscratch, env->retlog, (int)env->retcap)
        if (all_obs) { CALLM(true); } else { CALLM(false); }
#undef CALLM
        LAUNCH_OK();
        return DCM_OK;
    }
    // Every other batch of the mid-size class (A <= 128, T <= 256; uniform or ragged): rollout_fast_g.hpp, chunk counts from the batch dims
    if (quiet_ok && env->L.C == M && env->A <= 128 && env->T <= 256 && !(env->L.A == 20 && env->L.T == 50) && !(env->L.A == 64 && env->L.T == 64) &&
        (all_obs || no_obs)) {
        dcm::launch_rollout_fast_g(env->A > 64 ? 2 : 1, env->T > 192 ? 4 : (env->T > 128 ? 3 : 2), all_obs, (unsigned)env->p.n_envs,
                                   (unsigned)(Sim<128, 256, true>::lds_image_bytes(env->L)) + 512u, (hipStream_t)stream, DIMS(env), env->kp,
                                   env->state, (int)episodes, agents_out, tasks_out, mask_out, steps_out, env->summary, env->ablog,
                                   (const int32_t*)env->sizes, max_decisions, max_decisions_in, env->gscratch, env->retlog, (int)env->retcap);
        LAUNCH_OK();
        return DCM_OK;
    }
#endif
#define CALL(CA, CT, RS, ...)                                                                                         \
    hipLaunchKernelGGL((k_rollout_random<CA, CT, RS, ##__VA_ARGS__>), GRID(env),                                                     \
                       (Sim<CA, CT, RS>::SCR_IN_LDS ? env->L.lds_bytes() : Sim<CA, CT, RS, ((CT) > WAVE) && !(RS)>::lds_image_bytes(env->L)), (hipStream_t)stream, DIMS(env), \
                       env->kp, env->state, (int)episodes, agents_out, tasks_out, mask_out, steps_out, env->summary, env->ablog, \
                       (const int32_t*)env->sizes, max_decisions, max_decisions_in, env->gscratch, env->retlog, (int)env->retcap)
    DISPATCH_ENV(env, CALL);
#undef CALL
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_summary(dcm_env* env, double* out, void* stream) {
    CHECK_ENV(env);
    if (!out) return fail(DCM_ERR_INVALID, "dcm_summary: null out");
    { const int rc_ = dcm::flush_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    HIP_TRY(hipMemcpyAsync(out, env->summary, (size_t)env->p.n_envs * 8 * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return DCM_OK;
}

int dcm_env_status(dcm_env* env, uint32_t* flags_out, int64_t* decisions_out, double* now_out, void* stream) {
    CHECK_ENV(env);
    const int B = env->p.n_envs;
    hipLaunchKernelGGL(k_env_status, dim3((B + WAVE - 1) / WAVE), dim3(WAVE), 0, (hipStream_t)stream, env->L.A, env->L.T, env->L.C,
                       env->state, B, flags_out, decisions_out, now_out, (int32_t*)nullptr);   // (layout dims: only the record pitch matters)
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_env_episodes(dcm_env* env, int32_t* episodes_out, void* stream) {
    CHECK_ENV(env);
    if (!episodes_out) return fail(DCM_ERR_INVALID, "dcm_env_episodes: null episodes_out");
    const int B = env->p.n_envs;
    hipLaunchKernelGGL(k_env_status, dim3((B + WAVE - 1) / WAVE), dim3(WAVE), 0, (hipStream_t)stream, env->L.A, env->L.T, env->L.C,
                       env->state, B, (uint32_t*)nullptr, (int64_t*)nullptr, (double*)nullptr, episodes_out);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_get_tasks(dcm_env* env, uint8_t* finished, uint8_t* feasible, double* time_start, double* time_finish,
                  double* sum_wait, int32_t* status, int32_t* n_members, int32_t* n_abandoned, void* stream) {
    CHECK_ENV(env);
#define GETT(MCV) hipLaunchKernelGGL(k_get_tasks<MCV>, GRID(env), env->L.lds_rec(), (hipStream_t)stream, DIMS(env), env->kp, \
                       env->state, finished, feasible, time_start, time_finish, sum_wait, status, n_members, n_abandoned, \
                       env->ablog, (const int32_t*)env->sizes, env->gscratch)
    if (env->L.C > M) { GETT(MW); } else { GETT(M); }
#undef GETT
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_get_agents(dcm_env* env, double* sum_wait, double* travel_dist, double* next_decision, double* arrival, double* x,
                   double* y, uint8_t* returned, uint8_t* assigned, int32_t* current, int32_t* pending_group,
                   void* stream) {
    CHECK_ENV(env);
#define GETA(MCV) hipLaunchKernelGGL(k_get_agents<MCV>, GRID(env), env->L.lds_rec(), (hipStream_t)stream, DIMS(env), env->kp, \
                       env->state, sum_wait, travel_dist, next_decision, arrival, x, y, returned, assigned, current, \
                       pending_group, env->ablog, (const int32_t*)env->sizes, env->gscratch)
    if (env->L.C > M) { GETA(MW); } else { GETA(M); }
#undef GETA
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_get_members(dcm_env* env, int16_t* ids_out, void* stream) {
    CHECK_ENV(env);
    if (!ids_out) return fail(DCM_ERR_INVALID, "dcm_get_members: null ids_out");
    const int64_t n = (int64_t)env->p.n_envs * env->T;
    hipLaunchKernelGGL(k_get_members, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, env->T, env->L.A, env->L.T, env->L.C,
                       env->state, env->p.n_envs, ids_out, (const int32_t*)env->sizes);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_get_abandoned(dcm_env* env, uint16_t* counts_out, void* stream) {
    CHECK_ENV(env);
    if (!counts_out) return fail(DCM_ERR_INVALID, "dcm_get_abandoned: null counts_out");
    hipLaunchKernelGGL(k_get_abandoned, GRID(env), 0, (hipStream_t)stream, DIMS(env), env->L.C, env->state, env->ablog, counts_out,
                       (const int32_t*)env->sizes);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_state_bytes(dcm_env* env, size_t* bytes_out) {
    CHECK_HANDLE(env);
    if (!bytes_out) return fail(DCM_ERR_INVALID, "null bytes_out");
    *bytes_out = (size_t)env->p.n_envs * env->L.rec_bytes() + (size_t)env->p.n_envs * 8 * sizeof(double) +
                 side_bytes(env->p.n_envs, env->A, env->T);
    return DCM_OK;
}

int dcm_clone_state(dcm_env* env, void* dst, void* stream) {
    CHECK_ENV(env);
    if (!dst) return fail(DCM_ERR_INVALID, "dcm_clone_state: null dst");
    { const int rc_ = dcm::flush_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    const size_t sb = (size_t)env->p.n_envs * env->L.rec_bytes();
    HIP_TRY(hipMemcpyAsync(dst, env->state, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    const size_t mb = (size_t)env->p.n_envs * 8 * sizeof(double);
    HIP_TRY(hipMemcpyAsync((unsigned char*)dst + sb, env->summary, mb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync((unsigned char*)dst + sb + mb, env->ablog, side_bytes(env->p.n_envs, env->A, env->T),
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DCM_OK;
}

int dcm_restore_state(dcm_env* env, const void* src, void* stream) {
    CHECK_ENV(env);
    if (!src) return fail(DCM_ERR_INVALID, "dcm_restore_state: null src");
    { const int rc_ = dcm::drop_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    const size_t sb = (size_t)env->p.n_envs * env->L.rec_bytes();
    HIP_TRY(hipMemcpyAsync(env->state, src, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    const size_t mb = (size_t)env->p.n_envs * 8 * sizeof(double);
    HIP_TRY(hipMemcpyAsync(env->summary, (const unsigned char*)src + sb, mb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(env->ablog, (const unsigned char*)src + sb + mb,
                           side_bytes(env->p.n_envs, env->A, env->T), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    env->loaded = true;
    env->reset_done = true;
    env->init_valid = false;          // (the restored records may belong to other instances)
    return DCM_OK;
}

int dcm_distance(const double* ax, const double* ay, const double* bx, const double* by, double* dist_out, double* time_out,
                 int64_t n, void* stream) {
    if (!ax || !ay || !bx || !by || n < 0) return fail(DCM_ERR_INVALID, "dcm_distance: bad argument");
    if (n == 0) return DCM_OK;
    hipLaunchKernelGGL(k_distance, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ax, ay, bx, by,
                       dist_out, time_out, n);
    LAUNCH_OK();
    return DCM_OK;
}

#ifdef DCM_PROFILE_PHASES
int dcm_prof_read_fast(unsigned long long* out16, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fast_cycles), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fast_cycles), z, sizeof(z))); }
    return 0;
}
int dcm_prof_read(unsigned long long* out16, int reset) {
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z))); }
    return DCM_OK;
}
int dcm_prof_read_step(unsigned long long* out, int n_envs, int reset) {   // out[n_envs][32]
    if (n_envs > DCM_STEP_PROF_ENVS) n_envs = DCM_STEP_PROF_ENVS;
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_rows), sizeof(unsigned long long) * 32 * (size_t)n_envs));
    if (reset) { void* p = nullptr; HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_step_rows))); HIP_TRY(hipMemset(p, 0, sizeof(unsigned long long) * 32 * (size_t)DCM_STEP_PROF_ENVS)); }
    return DCM_OK;
}
#endif

int dcm_record_bytes(dcm_env* env, size_t* bytes_out) {
    CHECK_HANDLE(env);
    if (!bytes_out) return fail(DCM_ERR_INVALID, "null bytes_out");
    *bytes_out = env->L.rec_bytes();
    return DCM_OK;
}

}  // extern "C"
#endif  // DCM_DEVICE_ONLY_TU
