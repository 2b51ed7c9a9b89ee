// dcmrta_replay.hip -- route replay with optional dynamic task visibility on MI355X (gfx950).
//
// Restates TaskEnv.pre_set_route / execute_by_route (env/task_env.py:562-599) and the reactive branch
// of agent_update (:213-224): every deciding agent acts alone on the next entry of its preset route,
// max_waiting_time = 100, cut-off at t = 200, tasks become visible in batches of 20 every 10 time units
// (capped at 100) when reactive.  This is the deterministic known-answer path of the reference (the
// CTAS-D routes reproduce testSet_20A_50T_CONDET/metrics/metrics.csv:2) and BASELINE config 5.
//
// One wavefront per env, the whole episode in one launch.  LDS holds what the event loop reads on its critical path (agent
// arrays, task status words, member ids); the member arrival times, time_finish, the wake-up times and two per-agent
// accumulators live in a per-env HBM scratch (RLay below), the instance fields in the env record that dcm_load_instances
// filled.  Unlike RL mode a task may collect more members than its requirement, so member slots are sized by `member_cap`
// (<= 32) and walked with runtime loops.
//
// With the whole state in LDS the kernel ran one wave per SIMD (every instruction at its full issue latency, every dependent
// LDS access a full round trip); with 11.7 KB per env at 100A/500T fourteen waves share a CU and the CU's single scalar unit
// is the busiest resource, so what counts is the number of scalar instructions per agent step.  Like task_update,
// agent_update is incremental after an agent_step (only the agents whose inputs the step changed are recomputed), each phase
// issues its LDS reads back to back before it uses any of them, and wave-uniform words are NOT forced into SGPRs.
#include "common.hpp"

using namespace dcm;

namespace {

constexpr int MR_MAX = 32;
constexpr uint32_t R_TYPE_ERROR = DCM_FLAG_TYPE_ERROR;

// LDS layout of the replay state: what the event loop reads on its critical path.  [0, 48 A) is dead once the loop has ended and
// then holds the per-task waiting sums / the time_start copy of the terminal metrics (tsc(): behind the layout if 48 A < 8 T).
struct RLay {
    int A, T, MR;
    __device__ uint32_t ax() const { return 0; }
    __device__ uint32_t ay() const { return 8 * A; }
    __device__ uint32_t arr() const { return 16 * A; }
    __device__ uint32_t nd() const { return 24 * A; }
    __device__ uint32_t nx() const { return 32 * A; }                      // f64[A] x, y of the agent's NEXT preset target
    __device__ uint32_t ny() const { return 40 * A; }
    __device__ uint32_t aw() const { return 48 * A; }
    __device__ uint32_t cur() const { return 56 * A; }
    __device__ uint32_t ainfo() const { return 60 * A; }
    __device__ uint32_t phead() const { return 64 * A; }
    __device__ uint32_t plen() const { return 68 * A; }
    __device__ uint32_t tb() const { return 72 * A; }
    __device__ uint32_t tinfo() const { return tb(); }                     // u32[T]
    __device__ uint32_t mid() const { return tinfo() + 4 * T; }            // u8[MR][T]
    __device__ uint32_t tsc() const { return 48 * A >= 8 * T ? 0u : (uint32_t)((mid() + MR * T + 7) & ~7); }   // f64[T], terminal only
    // the "replay scratch" block (member arrivals, time_finish, travel_dist, max arrival, wake-up times) when it is kept in LDS
    // (grids of at most one wave per SIMD, see replay_lds_bytes): behind everything else
    __device__ uint32_t state() const { return (uint32_t)((mid() + MR * T + 7) & ~7) + (48 * A >= 8 * T ? 0u : 8u * (uint32_t)T); }
};
// What the event loop does not touch, or touches off its critical path, stays out of LDS.  In the HBM record: the read-only
// instance arrays (task x, y, duration: a task's duration is read once, when it becomes feasible; the coordinates of an agent's
// next target are staged per agent when its route is popped).  In the handle's per-env HBM scratch: time_start (written once per
// task, read by the terminal metrics), the abandonment log and the per-task abandonment counts (rare removal path).
// In the replay scratch dcm_load_routes allocates (replay_scratch_bytes per env): the member arrival times f64[T][member_cap]
// (read by the step that joins the task -- one line per task, requested together with the task's LDS words -- and by the task's
// own update), time_finish f64[T] (the step hands the joined task's value on in registers), travel_dist and
// max(arrival_time) f64[A] (loaded and stored by the agent's own step, never waited for).  A wave's global accesses are issued
// and served in order, so the wave sees its own stores (wavefront-scope fences need no cache action).
// The wake-up times f32[T] are there too (read once per event, eight coalesced loads in flight).  100A/500T with member_cap 5: 11.7 KB
// of LDS per env = FOURTEEN resident waves per CU (rounds 1-2: two, with 74 KB; round 3 at first three, then four with 39.3 KB).
__host__ __device__ inline size_t replay_scratch_bytes(int A, int T, int MR) {
    return (size_t)8 * T * MR + (size_t)8 * T + (size_t)16 * A + (((size_t)4 * T + 7) & ~(size_t)7);
}
// state_in_lds: the replay scratch block sits behind the LDS layout instead of in HBM.  With at most four envs per CU (the 8-GPU
// shard of BASELINE config 5: 1024 envs per GPU) every env is resident at once whatever it costs in LDS, there is nothing to
// overlap with, and what counts is the latency of one wave's chain: 35.3 KB per env at 100A/500T, all accesses LDS round trips.
__host__ __device__ inline uint32_t replay_lds_bytes(int A, int T, int MR, bool state_in_lds = false) {
    const uint32_t loop = (((uint32_t)(72 * A + 4 * T + MR * T) + 7u) & ~7u) + (48 * A >= 8 * T ? 0u : 8u * (uint32_t)T);
    return align16(loop + (state_in_lds ? (uint32_t)replay_scratch_bytes(A, T, MR) : 0u));
}

struct RP {
    double mwt;     // 100, env/task_env.py:564
    double cutoff;  // 200, :565
    int reactive;
    // dynamic-arrival schedule (dcm_set_visibility): visible = clip(now // period * batch + initial, initial, cap) :567 and
    // the depot re-arm time (next - 1) // batch * period :221; the reference hard-codes 20 / 20 / 10 / 100
    int vis_initial, vis_batch, vis_period, vis_cap;
};

// Python float floor division (now // 10, env/task_env.py:567)
__device__ double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx), div = (vx - mod) / wx, fl;
    if (mod != 0.0) { if ((wx < 0) != (mod < 0)) { mod += wx; div -= 1.0; } }
    if (div != 0.0) { fl = floor(div); if (div - fl > 0.5) fl += 1.0; } else fl = copysign(0.0, vx / wx);
    return fl;
}

// -DDCM_REPLAY_PHASES (tools/replay_phases.py, a separate build): shader clocks per phase of the event loop, summed per env and
// left in the first eight time_start entries of the env.  The s_memtime marks cost a scalar-memory round trip each: read the
// shares, not the totals.
#ifdef DCM_REPLAY_PHASES
#define RPH_DECL uint64_t rph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, rph_t = __builtin_readcyclecounter()
#define RPH(i) do { const uint64_t t_ = __builtin_readcyclecounter(); rph_acc[i] += t_ - rph_t; rph_t = t_; } while (0)
#define RPH_COUNT(i) (rph_acc[i] += 1)
#else
#define RPH_DECL
#define RPH(i)
#define RPH_COUNT(i)
#endif

__device__ __forceinline__ double rl(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

struct Rep {
    int A, T, MR;
    unsigned char* b;
    RLay L;
    const double *gtx, *gty, *gtd;     // task x, y, duration of this env in its HBM record (read-only)
    double* gts;                       // time_start[T],
    uint16_t* gab;                     // the abandonment log u16[A][AB_CAP] and
    uint32_t* gnab;                    // the abandonment counts u32[T] of this env in the handle's HBM scratch
    double* gmarr;                     // member arrival times f64[T][MR], then time_finish f64[T], travel_dist f64[A] and
                                       // max(arrival_time) f64[A] of this env (the replay scratch, see RLay)
    __device__ double* ax() const { return (double*)(b + L.ax()); }
    __device__ double* ay() const { return (double*)(b + L.ay()); }
    __device__ double* arr() const { return (double*)(b + L.arr()); }
    __device__ double* nd() const { return (double*)(b + L.nd()); }
    __device__ double* tdist() const { return gmarr + (size_t)T * MR + T; }
    __device__ double* aw() const { return (double*)(b + L.aw()); }
    __device__ double* amax() const { return gmarr + (size_t)T * MR + T + A; }
    __device__ int32_t* cur() const { return (int32_t*)(b + L.cur()); }
    __device__ uint32_t* ainfo() const { return (uint32_t*)(b + L.ainfo()); }
    __device__ int32_t* phead() const { return (int32_t*)(b + L.phead()); }
    __device__ int32_t* plen() const { return (int32_t*)(b + L.plen()); }
    __device__ double* ts() const { return gts; }
    __device__ double* tf() const { return gmarr + (size_t)T * MR; }
    __device__ double* nx() const { return (double*)(b + L.nx()); }
    __device__ double* ny() const { return (double*)(b + L.ny()); }
    __device__ double* tw() const { return (double*)(b + L.tsc()); }           // terminal only
    __device__ double& marr(int j, int t) const { return gmarr[t * MR + j]; }
    __device__ uint32_t* tinfo() const { return (uint32_t*)(b + L.tinfo()); }
    __device__ uint32_t* tnab() const { return gnab; }
    __device__ float* wake() const { return (float*)(gmarr + (size_t)T * MR + T + 2 * A); }   // f32[T] earliest time a task_update call can change the task
    __device__ uint8_t* mid() const { return (uint8_t*)(b + L.mid()); }
    __device__ uint16_t* ablog() const { return gab; }

    // tinfo[t]: bits 0-7 requirements, 8-15 status (int8), 16-23 len(members), 24 feasible, 25 finished

    // env/task_env.py:245-281 for ONE task (lane-private).  Bit 0 of the result: the call changed the member list or made the
    // task feasible at a `now` at which it is already over -- the only cases in which an immediate second call at the same `now`
    // can change it again (a Q1-skipped member / stale status after the spread branch / `finished` of a task that has just
    // become feasible, :273).  Bit 1: the task became feasible; bit 2: members were removed -- the two things agent_update reads.
    __device__ uint32_t task_update_one(int t, double now, double mwt) const {
        uint32_t info = tinfo()[t];
        bool touched = false, became = false, removed = false;
        double w = __builtin_inf();          // earliest time at which a later call can change the task if nobody joins it
        if (!(info & T_FEAS)) {                                              // :249
            const int req = info & 0xFF;
            const int n = (info >> 16) & 0xFF;                               // :250
            const int status = req - n;                                      // :252
            uint32_t keep = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
            bool changed = false;
            if (status <= 0) {                                               // :254
                double mx = marr(0, t), mn = mx;
                for (int j = 1; j < n; j++) { const double v = marr(j, t); mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
                if (mx - mn <= mwt) {                                        // :255
                    const double tfin = mx + gtd[t];
                    ts()[t] = mx; tf()[t] = tfin; info |= T_FEAS;            // :256-258
                    became = true;
                    touched = now >= tfin;   // only a task that is already over changes again at this `now` (finished, :273)
                    w = tfin;
                } else {
                    const double thr = mx - mwt;                             // :262
                    for (int j = 0; j < n; j++) if (marr(j, t) <= thr) { keep &= ~(1u << j); changed = true; }
                }
            } else {
                bool skip = false;                                           // :268-271 (quirk Q1)
                double mn = __builtin_inf();
                for (int j = 0; j < n; j++) {
                    const double v = marr(j, t);
                    mn = v < mn ? v : mn;
                    if (skip) { skip = false; continue; }
                    if (now - v >= mwt) { keep &= ~(1u << j); changed = true; skip = true; }  // :269
                }
                w = mn + mwt;                // the earliest member's limit (+inf without members)
            }
            int nn = n;
            if (changed) {
                int k = 0;
                for (int j = 0; j < n; j++) {
                    const uint32_t id = mid()[j * T + t];
                    if (keep & (1u << j)) {
                        if (k != j) { mid()[k * T + t] = (uint8_t)id; marr(k, t) = marr(j, t); }
                        k++;
                    } else {
                        const uint32_t nth = atomicAdd(&ainfo()[id], 1u << 16) >> 16;   // abandoned_agent.append :265/:271
                        if (nth < (uint32_t)AB_CAP) ablog()[id * AB_CAP + nth] = (uint16_t)t;
                        if (cur()[id] == t) atomicAnd(&ainfo()[id], ~A_MEMBER);
                    }
                }
                tnab()[t] += (uint32_t)(n - k);
                nn = k;
                touched = true; removed = true;
                w = -__builtin_inf();        // the next call looks again (stale status byte, a Q1-skipped member)
            }
            info = (info & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)nn << 16);
        } else if (!(info & T_FIN)) {
            const double tfin = tf()[t];
            if (now >= tfin) info |= T_FIN;                                  // :273-274
            else w = tfin;
        }
        tinfo()[t] = info;
        wake()[t] = __double2float_rd(w);
        return (touched ? 1u : 0u) | (became ? 2u : 0u) | (removed ? 4u : 0u);
    }

    // The same for the task an agent_step has just touched, from registers: the whole wave works on task k with what the step has
    // already read -- `info` = the task's word as the step left it, tf_k = its finish time (if it is feasible), v = on lane j the
    // arrival time of member slot j (j < len(members), the new member's included), dur = its duration -- so the common outcomes
    // (still waiting for members; now complete and feasible) cost no LDS round trip at all.  A call that has to remove members
    // (rare) is handed to task_update_one.  Same result bits.
    __device__ __forceinline__ uint32_t task_update_joined(int k, uint32_t info, double tf_k, double v, double dur, double now,
                                                           double mwt, int lane) const {
        if (info & T_FEAS) {                                                 // :273-274
            if (!(info & T_FIN) && now >= tf_k && lane == 0) { tinfo()[k] = info | T_FIN; wake()[k] = __builtin_inff(); }
            return 0;
        }
        const int req = info & 0xFF, n = (info >> 16) & 0xFF, status = req - n;   // :250-252
        bool became = false, touched = false, removal;
        double mx = -__builtin_inf(), mn = __builtin_inf(), w;
        for (int j = 0; j < n; j++) { const double vj = rl(v, j); mx = vj > mx ? vj : mx; mn = vj < mn ? vj : mn; }
        if (status <= 0) {                                                   // :254
            removal = !(mx - mn <= mwt);                                     // :255 / :262 (the earliest arrival is then <= max - mwt)
            if (!removal) {
                const double tfin = mx + dur;
                if (lane == 0) { ts()[k] = mx; tf()[k] = tfin; }             // :256-258
                info |= T_FEAS;
                became = true;
                touched = now >= tfin;       // only a task that is already over changes again at this `now` (finished, :273)
                w = tfin;
            }
        } else {
            removal = __ballot(lane < n && now - v >= mwt) != 0;             // :268-271: the first such member is always removed
            w = mn + mwt;
        }
        if (removal) {
            uint32_t r = 0;
            if (lane == 0) r = task_update_one(k, now, mwt);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        }
        info = (info & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)n << 16);
        if (lane == 0) { tinfo()[k] = info; wake()[k] = __double2float_rd(w); }
        return (touched ? 1u : 0u) | (became ? 2u : 0u);
    }

    // task_update (:245-281).  only == -1: every task, like the reference.  only >= 0 / -2: the call that follows an
    // agent_step at an unchanged `now` when the previous call reported redo == false -- then every task except the one the
    // agent has just joined (`only`; -2 = it went to the depot) is at a fixed point of task_update and is skipped, which
    // turns the T/64 lane passes of this call into one.  n_infeas carries the number of tasks that are not feasible
    // (feasible_assignment never reverts), for np.all(feasible) of the depot check :279.
    // what (out): what the call changed of the things agent_update derives the agents' next decisions from -- TU_FULL after a pass
    // over every task (anything), else TU_BECAME (task `only` became feasible) and / or TU_REMOVED (it lost members), or 0.
    static constexpr uint32_t TU_FULL = 1u, TU_BECAME = 2u, TU_REMOVED = 4u;
    // (info_k, tf_k, slot_v, dur_k: what task_update_joined takes, for only >= 0.)
    __device__ void task_update(double now, double mwt, int lane, int only, bool& redo, int& n_infeas, uint32_t& what,
                                uint32_t info_k = 0, double tf_k = 0.0, double slot_v = 0.0, double dur_k = 0.0) const {
        bool touched = false;
        what = only == -1 ? TU_FULL : 0u;
        if (only == -1) {
            // Every visit of a task leaves in wake()[t] the earliest time at which a later call can change it (time_finish; the
            // earliest member's arrival + max_waiting_time; -inf after a removal: stale status byte, Q1-skipped member; +inf if
            // nothing is pending), rounded DOWN to fp32 -- a margin far above the rounding of the fp64 expressions it stands for.
            // A task with now < wake is at a fixed point of task_update_one, so the pass reads the wake-up times of 512 tasks in one
            // LDS round trip and visits the due ones: one or two lanes per event instead of all T tasks.
            int infeas = n_infeas;
#pragma nounroll
            for (int tb = 0; tb < T; tb += 8 * WAVE) {
                float wk[8];
                const int nc = (T - tb + WAVE - 1) / WAVE < 8 ? (T - tb + WAVE - 1) / WAVE : 8;
#pragma unroll
                for (int c = 0; c < 8; c++) if (c < nc) { const int t = tb + c * WAVE + lane; wk[c] = wake()[t < T ? t : 0]; }
                uint32_t due = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) if (c < nc) { if (tb + c * WAVE + lane < T && now >= (double)wk[c]) due |= 1u << c; }
                // the chunks with a due task: OR of the lanes' masks (DPP row shifts / broadcasts, no scalar loop over all eight)
                uint32_t anyc = due;
                anyc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)anyc, 0x111, 0xF, 0xF, false);   // row_shr:1
                anyc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)anyc, 0x112, 0xF, 0xF, false);   // row_shr:2
                anyc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)anyc, 0x114, 0xF, 0xF, false);   // row_shr:4
                anyc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)anyc, 0x118, 0xF, 0xF, false);   // row_shr:8
                anyc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)anyc, 0x142, 0xA, 0xF, false);   // row_bcast:15
                anyc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)anyc, 0x143, 0xC, 0xF, false);   // row_bcast:31
                uint32_t todo = (uint32_t)__builtin_amdgcn_readlane((int)anyc, 63);
#pragma nounroll
                for (; todo; todo &= todo - 1) {
                    const int c = __ffs((int)todo) - 1;
                    const bool act = (due >> c) & 1u;
                    uint32_t r = 0;
                    if (act) r = task_update_one(tb + c * WAVE + lane, now, mwt);
                    touched = touched || (r & 1u);
                    infeas -= __popcll(__ballot(r & 2u));                    // became feasible
                }
            }
            n_infeas = infeas;
        } else if (only >= 0) {
            const uint32_t r = uni(task_update_joined(only, info_k, tf_k, slot_v, dur_k, now, mwt, lane));
            touched = r & 1u;
            if (r & 2u) { n_infeas -= 1; what |= TU_BECAME; }
            if (r & 4u) what |= TU_REMOVED;
        }
        redo = __any(touched);
        const bool all_feasible = n_infeas == 0;
        WSYNC();
        if (all_feasible) {                                                  // depot :277-280 (uniform; false until the very end)
#pragma nounroll
            for (int a = lane; a < A; a += WAVE) {
                const uint32_t ai = ainfo()[a];
                if ((ai & A_INDEPOT) && now >= arr()[a]) ainfo()[a] = ai | A_RETURNED;
            }
        }
    }

    // env/task_env.py:207-243 including the reactive depot branch :213-224.
    // mode AU_FULL: every agent, like the reference.  The calls that follow an agent_step of agent `single` at an unchanged `now`
    // need less: an agent's branch below reads its own fields, the feasibility flag and times of its current task, its membership
    // flag and all(feasible[:visible]); if the task_update in between changed none of these for anybody else (what == 0) only
    // that agent is recomputed (AU_ONE, on lane 0); if it made the joined task k feasible, its listed members are as well
    // (AU_TASK, one per lane) -- everybody else would read what it read in the previous call and store what it stored.
    // ninf_vis (in/out) = number of infeasible tasks among the visible ones, counted by AU_FULL and maintained by the caller
    // (a flip of all(feasible[:visible]) changes every depot agent: the caller then asks for AU_FULL).
    // (agent['assigned'] :232-240 is not kept: nothing on the replay path reads it.)
    static constexpr int AU_ONE = 0, AU_TASK = 1, AU_FULL = 2;
    // :221-222: when a depot agent whose next preset task `next` is not visible yet decides again -- np.max([arrival at the depot,
    // (next - 1) // batch * period, now])
    // ((next - 1) // batch * period of :221 as a double)
    __device__ __forceinline__ static double rearm_quantum(int next, int vis_batch, int vis_period) {
        // (next - 1) // batch for 0 <= next - 1 < 65536 and 2 <= batch < 65536 is one multiply-high by this constant
        const uint32_t magic = vis_batch >= 2 && vis_batch < 65536 ? 0xFFFFFFFFu / (uint32_t)vis_batch + 1u : 0u;
        const int x = next - 1;
        int q;                                                               // python floor division
        if (magic && x >= 0 && x < 65536) q = (int)__umulhi((uint32_t)x, magic);
        else { q = x / vis_batch; if (x % vis_batch != 0 && x < 0) q--; }
        return (double)(q * vis_period);
    }
    __device__ __forceinline__ static double rearm_time(int next, double arrv, double now, int vis_batch, int vis_period) {
        const double ndt = rearm_quantum(next, vis_batch, vis_period);
        double v = arrv;
        v = ndt > v ? ndt : v;
        v = now > v ? now : v;
        return v;
    }
    __device__ void agent_update(double now, double mwt, int reactive, int visible, int lane, uint32_t& flags, int vis_batch,
                                 int vis_period, int mode, int k, int single, int& ninf_vis) const {
        if (reactive && mode == AU_FULL) {                                   // :214 all(feasible[:visible_length])
            const int lim = visible < T ? visible : T;
            int cnt = 0;
#pragma nounroll
            for (int t0 = 0; t0 < lim; t0 += WAVE) {
                const int t = t0 + lane;
                cnt += __popcll(__ballot(t < lim && !(tinfo()[t < lim ? t : 0] & T_FEAS)));
            }
            ninf_vis = cnt;
        }
        const bool allf_vis = ninf_vis == 0;
        int a_first = lane, a_stop = A;
        if (mode != AU_FULL) {
            const int nk = mode == AU_TASK ? (int)((tinfo()[k] >> 16) & 0xFF) : 0;
            a_first = lane < nk ? (int)mid()[lane * T + k] : (lane == nk ? single : A);
            a_stop = a_first < A ? a_first + 1 : 0;
        }
        bool terr = false;
        for (int a = a_first; a < a_stop; a += WAVE) {
            // what either branch reads about the agent: one LDS round trip; then its current task's words: a second one
            const int c = cur()[a], len = plen()[a], head = phead()[a];
            const uint32_t ai = ainfo()[a];
            const double arrv = arr()[a], nxtd = aw()[a];                    // (aw: the next preset action, staged by the kernel)
            const int cc = c >= 0 ? c : 0;
            const uint32_t info = tinfo()[cc];
            const double tfc = tf()[cc];
            if (c == -2) continue;                                           // :209
            double v;
            if (c == -1) {                                                   // :212
                if (!reactive || allf_vis || (len >= 0 && head >= len)) v = __builtin_nan("");   // :215,:226 / :217-218
                else if (len < 0) { terr = true; continue; }                 // :220 TypeError in the reference
                else {
                    v = rearm_time((int)nxtd, arrv, now, vis_batch, vis_period);   // :221-222
                    ainfo()[a] = ai & ~A_INDEPOT;                            // :223-224 depot['members'].remove
                }
            } else {
                const bool member = (info & T_FEAS) && (ai & A_MEMBER);      // :228-230
                v = member ? tfc : arrv + mwt;                               // :231 / :235,:238
            }
            nd()[a] = v;
        }
        if (__any(terr)) flags |= R_TYPE_ERROR;
    }
};

// <CA, CT, CMR> = the batch's agents / tasks / member slots as compile-time constants (every LDS offset and loop bound folds), or
// <0, 0, 0> = read from the arguments.  SLDS: the replay scratch block in LDS instead of HBM (see replay_lds_bytes).
template <int CA, int CT, int CMR, bool SLDS>
__global__ __launch_bounds__(WAVE) void k_replay(int A_, int T_, int PA, int PT, int MR_, RP P, const unsigned char* state,
                                                const int32_t* routes, const int32_t* route_len, int route_cap,
                                                double* summary, int64_t* steps_out, uint32_t* flags_out,
                                                uint8_t* finished, double* time_start, double* time_finish,
                                                double* task_wait, int32_t* n_members, double* agent_wait,
                                                double* travel_dist, uint8_t* returned, unsigned char* gscr, double* member_arrivals) {
    const int e = blockIdx.x, lane = threadIdx.x;
    const int A = CA ? CA : A_, T = CT ? CT : T_, MR = CMR ? CMR : MR_;
    const Lay EL{PA, PT};                                      // layout dims of the handle's records (>= the batch dims)
    const unsigned char* rec = state + (size_t)e * EL.rec_bytes();
    Rep R{A, T, MR, smem, RLay{A, T, MR}, (const double*)(rec + EL.tx()), (const double*)(rec + EL.ty()),
          (const double*)(rec + EL.tdur()), (double*)(gscr + (size_t)e * EL.scratch_bytes() + EL.s_tw()),
          (uint16_t*)(gscr + (size_t)e * EL.scratch_bytes() + EL.s_absort()),
          (uint32_t*)(gscr + (size_t)e * EL.scratch_bytes() + EL.s_terms()), SLDS ? (double*)(smem + RLay{A, T, MR}.state())
               : (double*)((unsigned char*)member_arrivals + (size_t)e * replay_scratch_bytes(A, T, MR))};
    const Hdr* gh = (const Hdr*)rec;
    const double depot_x = uni(gh->depot_x), depot_y = uni(gh->depot_y);
    const int32_t* my_routes = routes + (size_t)e * A * route_cap;
    // ---- clear_decisions (env/task_env.py:129-140) from the loaded instance
    {
        const uint32_t* gti = (const uint32_t*)(rec + EL.tinfo());
#pragma nounroll
        for (int t = lane; t < T; t += WAVE) {
            const uint32_t req = gti[t] & 0xFF;
            R.tinfo()[t] = req | (req << 8);
            R.tnab()[t] = 0; R.ts()[t] = 0.0; R.tf()[t] = 0.0; R.wake()[t] = __builtin_inff();
        }
#pragma nounroll
        for (int a = lane; a < A; a += WAVE) {
            R.ax()[a] = depot_x; R.ay()[a] = depot_y; R.arr()[a] = 0.0; R.amax()[a] = 0.0; R.nd()[a] = 0.0; R.tdist()[a] = 0.0;
            R.aw()[a] = 0.0; R.cur()[a] = -2; R.ainfo()[a] = 0; R.phead()[a] = 0;
            R.plen()[a] = route_len[(size_t)e * A + a];                      // pre_set_route :595-599 (-1 = None)
        }
    }
    double now = 0.0;
    uint32_t flags = 0;
    bool finished_flag = false, redo = true;
    int visible = 0, guard = 0, n_infeas = T;
    int steps = 0;
    const double mwt = P.mwt;                                                // :564
    // Guard (not in the reference): with reactive planning a depot agent whose next task id is beyond the hard
    // visibility cap of 100 re-decides at the same time forever (env/task_env.py:220-222,578-584); stop such envs.
    const int step_cap = 64 * (A + T) + 4096;
    WSYNC();
    // aw()[a] (scratch until the terminal metrics) holds the next preset action of agent a: staged once for everybody,
    // then refreshed only for the agent that pops its route (its entry is the only one that changes), with the global
    // load issued at the pop and consumed after the updates so that its latency hides behind task_update/agent_update
#pragma nounroll
    for (int a = lane; a < A; a += WAVE) {
        const int len = R.plen()[a];
        const int32_t first = (len > 0) ? my_routes[(size_t)a * route_cap] : 0;
        R.aw()[a] = (double)first;
        const bool task = first >= 1 && first <= T;      // (an out-of-range entry is reported when it is executed)
        R.nx()[a] = task ? R.gtx[first - 1] : depot_x;
        R.ny()[a] = task ? R.gty[first - 1] : depot_y;
    }
    WSYNC();
    uint32_t tu_what = 0;
    int ninf_vis = 0;
    RPH_DECL;
    // next_decision_time of the lane's agents (NaN beyond A) and their minimum: read once per event, by check_finished, and used
    // again by next_decision of the following event (nothing changes them in between)
    double ndv[AW_MAX], tmin;
    auto read_next_decisions = [&]() {
        double lmin = __builtin_nan("");
#pragma unroll
        for (int i = 0; i < AW_MAX; i++) {
            const int a = i * WAVE + lane;
            ndv[i] = R.nd()[a < A ? a : 0];
            if (a >= A) ndv[i] = __builtin_nan("");
            lmin = nanmin2(lmin, ndv[i]);
        }
        tmin = wave_nanmin(lmin);
    };
    auto latest_arrival = [&]() {                                           // max(x) if x else 0 over the whole arrival lists
        double lmax = 0.0;
#pragma nounroll
        for (int a = lane; a < A; a += WAVE) { const double av = R.amax()[a]; lmax = av > lmax ? av : lmax; }
        return wave_nanmax(lmax);
    };
    read_next_decisions();
    // visible = clip(now // period * batch + initial) :567 changes only when `now` leaves [vis_lo, vis_hi) = q * period .. (q + 1) *
    // period (the float floor division is exact for these magnitudes): the fmod / division sequence runs once per window
    double vis_lo = __builtin_inf(), vis_hi = -__builtin_inf();
    while (!finished_flag && now < P.cutoff) {                               // :565
        RPH(5); RPH_COUNT(6);
        if (P.reactive && !(now >= vis_lo && now < vis_hi)) {                // :566-567
            const double q = py_floordiv(now, (double)P.vis_period);
            vis_lo = q * (double)P.vis_period; vis_hi = (q + 1.0) * (double)P.vis_period;
            double v = q * (double)P.vis_batch + (double)P.vis_initial;
            v = v < (double)P.vis_initial ? (double)P.vis_initial : v; v = v > (double)P.vis_cap ? (double)P.vis_cap : v;
            visible = (int)v;
        }
        // next_decision :283-289
        const bool any = (tmin == tmin);
        now = any ? tmin : latest_arrival();                                 // :569
        // deciding set (exact ==), fixed before the updates like the reference's `decision_agents`
        uint64_t dm[AW_MAX];
#pragma unroll
        for (int i = 0; i < AW_MAX; i++) dm[i] = __ballot(any && ndv[i] == tmin);
        WSYNC();
        RPH(0);
        R.task_update(now, mwt, lane, -1, redo, n_infeas, tu_what);          // :570
        WSYNC();
        RPH(1);
        R.agent_update(now, mwt, P.reactive, visible, lane, flags, P.vis_batch, P.vis_period, Rep::AU_FULL, -1, -1, ninf_vis);   // :571
        WSYNC();
        RPH(3);
        if (flags & R_TYPE_ERROR) break;
        if (!any) { if (++guard > 8) { flags |= DCM_FLAG_TRUNCATED; break; } } else guard = 0;
#pragma unroll
        for (int i = 0; i < AW_MAX; i++) {
            uint64_t m = dm[i];
            while (m) {                                                      // :572 for agent in decision_agents
                const int a = i * 64 + __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                // everything the step reads about agent a, in ONE LDS round trip (wave-uniform addresses, broadcast reads)
                const int len_v = R.plen()[a], head_v = R.phead()[a], cur_old = R.cur()[a];
                const double nxt_v = R.aw()[a], sx = R.nx()[a], sy = R.ny()[a], px = R.ax()[a], py = R.ay()[a], td = R.tdist()[a],
                             amx = R.amax()[a];
                uint32_t ai = R.ainfo()[a];
                const int len = len_v, head = head_v;
                int action;
                bool popped = false;
                int32_t upcoming = 0;
                double up_x = depot_x, up_y = depot_y;
                if (len < 0 || head >= len) action = 0;                      // :573-577
                else {
                    const int nxt = (int)nxt_v;                         // == my_routes[a][head]
                    if (P.reactive && nxt > visible) action = 0;             // :578-584
                    else {
                        action = nxt; popped = true;                         // :585 pop(0)
                        if (lane == 0) {
                            R.phead()[a] = head + 1;
                            if (head + 1 < len) {
                                upcoming = my_routes[(size_t)a * route_cap + head + 1];
                                if (upcoming >= 1 && upcoming <= T) { up_x = R.gtx[upcoming - 1]; up_y = R.gty[upcoming - 1]; }
                            }
                        }
                    }
                }
                if (action < 0 || action > T) { flags |= DCM_FLAG_BAD_ACTION; break; }
                // agent_step :300-324.  Second round trip, in flight during the distance arithmetic: the joined task's info word
                // and, one per lane, its member ids.
                const int k = action - 1, kk = k >= 0 ? k : 0;
                uint32_t info = R.tinfo()[kk];
                const int mine = lane < MR ? (int)R.mid()[lane * T + kk] : -1;
                double slot_v = R.marr(lane < MR ? lane : 0, kk);              // lane j: arrival time of member slot j
                const double tf_k = R.tf()[kk];
                const double dur_k = R.gtd[kk];              // (from HBM, needed only if the task becomes feasible in this step)
                // target of a popped action = the coordinates staged for this agent (see `upcoming` below); a forced depot visit
                // (route exhausted / next task not yet visible) leaves them in place for later
                const double tx_ = action ? sx : depot_x, ty_ = action ? sy : depot_y;
                // (distance, sqrt and division on all lanes -- wave-uniform values -- and only the stores on lane 0: gfx950 runs
                //  fp64 VALU instructions with fewer than 16 active lanes 4x slower, profiles/r03_calib)
                const double d = dist2(px, py, tx_, ty_);
                const double arrival = now + over_velocity(d);                      // :315,:318
                const double tdist_new = td + d;                                    // :317
                ai = ai & ~A_MEMBER;
                int pos = -1;
                bool fresh = false;
                if (action == 0) ai |= A_INDEPOT;                            // :321-322
                else {
                    int n = (info >> 16) & 0xFF;
                    const uint64_t hit = __ballot(lane < n && mine == a);
                    pos = hit ? 63 - __clzll((long long)hit) : -1;           // (the last matching slot, like a scan would find)
                    if (pos < 0) {
                        if (n >= MR) flags |= DCM_FLAG_OVERFLOW;
                        else { pos = n++; fresh = true; }
                    }
                    if (pos >= 0) ai |= A_MEMBER;
                    info = (info & ~0x00FF0000u) | ((uint32_t)n << 16);
                }
                const bool joined = pos >= 0;
                if (joined && lane == pos) slot_v = arrival;
                if (lane == 0) {
                    R.tdist()[a] = tdist_new;
                    R.arr()[a] = arrival;
                    // a member released by its task finishing before it arrived re-decides early, so the list is
                    // not monotone in replays with surplus visitors; :286 takes the max over the whole list
                    if (cur_old == -2 || arrival > amx) R.amax()[a] = arrival;
                    R.ax()[a] = tx_; R.ay()[a] = ty_;                        // :320
                    R.cur()[a] = k;                                          // :314
                    R.ainfo()[a] = ai;
                    if (action) {
                        if (fresh) R.mid()[pos * T + k] = (uint8_t)a;
                        if (joined) R.marr(pos, k) = arrival;
                        R.tinfo()[k] = info;
                        R.wake()[k] = -__builtin_inff();                     // its member list changed: the next task_update visits it
                    }
                }
                if (++steps > step_cap) flags |= DCM_FLAG_TRUNCATED | DCM_FLAG_OVERFLOW;
                WSYNC();
                RPH(4); RPH_COUNT(7);
#ifdef DCM_REPLAY_PHASES
                const bool full_ = redo;
#endif
                R.task_update(now, mwt, lane, redo ? -1 : (action > 0 ? action - 1 : -2), redo, n_infeas, tu_what, info, tf_k, slot_v,
                              dur_k);                                        // :575/:582/:586
                if (popped && lane == 0) { R.aw()[a] = (double)upcoming; R.nx()[a] = up_x; R.ny()[a] = up_y; }   // before agent_update: its reactive branch reads aw
                WSYNC();
#ifdef DCM_REPLAY_PHASES
                if (full_) RPH(1); else RPH(2);
#endif
                int au_mode = Rep::AU_ONE;
                if (tu_what & (Rep::TU_FULL | Rep::TU_REMOVED)) au_mode = Rep::AU_FULL;
                else if (tu_what & Rep::TU_BECAME) {
                    au_mode = Rep::AU_TASK;
                    if (P.reactive && action - 1 < visible && --ninf_vis == 0) au_mode = Rep::AU_FULL;   // all(feasible[:visible]) flips
                }
                if (au_mode == Rep::AU_ONE && action > 0) {
                    // the common case written out: only agent a changes, and what its branch of agent_update reads is at hand
                    // (the task_update in between changed nothing about task k -- otherwise tu_what would say so -- so its
                    //  feasibility flag and finish time are the ones the step read)
                    if (lane == 0) R.nd()[a] = ((info & T_FEAS) && joined) ? tf_k : arrival + mwt;        // :229-238
                } else if (au_mode == Rep::AU_ONE && !popped) {
                    // ... and for a forced step to the depot (route exhausted / next task not visible yet: nothing was popped, so
                    // the route position and the staged next action are the ones the step read; :212-226): the agent waits
                    // there for good (nan) when the replay is not reactive, every visible task is feasible or its route is
                    // exhausted; otherwise it is re-armed (:220-224)
                    double v = __builtin_nan("");
                    if (P.reactive && ninf_vis != 0 && !(len >= 0 && head >= len)) {
                        if (len < 0) flags |= R_TYPE_ERROR;                  // :220 TypeError in the reference
                        else {
                            v = Rep::rearm_time((int)nxt_v, arrival, now, P.vis_batch, P.vis_period);
                            if (lane == 0) R.ainfo()[a] &= ~A_INDEPOT;       // :223-224 depot['members'].remove
                        }
                    }
                    if (lane == 0 && !(flags & R_TYPE_ERROR)) R.nd()[a] = v;
                } else
                    R.agent_update(now, mwt, P.reactive, visible, lane, flags, P.vis_batch, P.vis_period, au_mode, k, a,
                                   ninf_vis);                                // :576/:583/:587
                WSYNC();
#ifdef DCM_REPLAY_PHASES   // agent_update after a step, by kind: 8 = written out inline, 9 = one agent (generic), 10 = task / full
                if (au_mode == Rep::AU_ONE && action > 0) RPH(8); else if (au_mode == Rep::AU_ONE && !popped) RPH(9); else { RPH(10); RPH_COUNT(11); }
#endif
                if (flags & (R_TYPE_ERROR | DCM_FLAG_OVERFLOW)) break;
            }
            if (flags & (R_TYPE_ERROR | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_ACTION)) break;
        }
        if (flags & (R_TYPE_ERROR | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_ACTION)) break;
        // check_finished :366-373,:588
        read_next_decisions();
        if (!(tmin == tmin)) {
            now = latest_arrival();
            bool allret = true, allfin = true;
#pragma nounroll
            for (int a = lane; a < A; a += WAVE) allret = allret && (R.ainfo()[a] & A_RETURNED);
#pragma nounroll
            for (int t = lane; t < T; t += WAVE) allfin = allfin && (R.tinfo()[t] & T_FIN);
            finished_flag = __all(allret) && __all(allfin);
        } else finished_flag = false;
    }
    WSYNC();
    if (time_finish) for (int t = lane; t < T; t += WAVE) time_finish[(size_t)e * T + t] = R.tf()[t];
    WSYNC();
    // ---- get_episode_reward: calculate_waiting_time :344-364 (np.sum = pairwise block for n >= 8)
#pragma nounroll
    for (int t = lane; t < T; t += WAVE) {
        const uint32_t info = R.tinfo()[t];
        const int n = (info >> 16) & 0xFF;
        const double ab = (double)R.tnab()[t] * mwt;
        double s = 0.0;
        if (n != 0) {
            double mx = R.marr(0, t);
            for (int j = 1; j < n; j++) { const double v = R.marr(j, t); mx = v > mx ? v : mx; }
            const bool feas = info & T_FEAS;
            auto term = [&](int j) { const double v = R.marr(j, t); return feas ? (mx - v) : (now - v); };
            if (n < 8) { for (int j = 0; j < n; j++) s += term(j); }
            else {
                double r[8];
                for (int j = 0; j < 8; j++) r[j] = term(j);
                int i;
                for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += term(i + j);
                s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < n; i++) s += term(i);
            }
        }
        R.tw()[t] = s + ab;
    }
    // :358-364 per agent in the reference's order: tasks ascending, member term first, then +max_waiting_time per
    // entry of the agent in that task's abandoned_agent list (entries from the abandonment log, sorted by task id)
    bool over = false;
#pragma nounroll
    for (int a = lane; a < A; a += WAVE) {
        const uint32_t nab = R.ainfo()[a] >> 16;
        const int nl = nab < (uint32_t)AB_CAP ? (int)nab : AB_CAP;
        over = over || nab > (uint32_t)AB_CAP;
        uint16_t* my = R.ablog() + a * AB_CAP;
        for (int i = 1; i < nl; i++) {
            const uint16_t v = my[i];
            int j = i;
            while (j > 0 && my[j - 1] > v) { my[j] = my[j - 1]; j--; }
            my[j] = v;
        }
        int p = 0;
        double s = 0.0;
#pragma nounroll
        for (int t = 0; t < T; t++) {
            const uint32_t info = R.tinfo()[t];
            const int n = (info >> 16) & 0xFF;
            int pos = -1;
            for (int j = 0; j < n; j++) if (R.mid()[j * T + t] == a) pos = j;
            if (pos >= 0) {
                const double mine = R.marr(pos, t);
                if (info & T_FEAS) {
                    double mx = R.marr(0, t);
                    for (int j = 1; j < n; j++) { const double v = R.marr(j, t); mx = v > mx ? v : mx; }
                    s += mx - mine;                                          // :360
                } else { const double w = now - mine; s += (w > 0.0) ? w : 0.0; }   // :362
            }
            while (p < nl && my[p] == (uint16_t)t) { s += mwt; p++; }        // :363-364
        }
        s += (double)(nab - (uint32_t)nl) * mwt;
        R.aw()[a] = s;
    }
    if (__any(over)) flags |= DCM_FLAG_WAIT_ORDER;
    WSYNC();
    int nfin = 0;
#pragma nounroll
    for (int t0 = 0; t0 < T; t0 += WAVE) {
        const int t = t0 + lane;
        nfin += __popcll(__ballot(t < T && (R.tinfo()[t < T ? t : 0] & T_FIN)));
    }
    const double Td = (double)T, Ad = (double)A;
    const double m3 = psum<4>(R.aw(), A) / Ad, m4 = psum<4>(R.tdist(), A), m5 = psum<4>(R.tw(), T) / Td;
#pragma nounroll
    for (int t = lane; t < T; t += WAVE) {
        const size_t o = (size_t)e * T + t;
        const uint32_t info = R.tinfo()[t];
        if (finished) finished[o] = (info & T_FIN) ? 1 : 0;
        if (task_wait) task_wait[o] = R.tw()[t];
        if (n_members) n_members[o] = (info >> 16) & 0xFF;
    }
    WSYNC();
    // time_start comes back from the HBM scratch into the LDS scratch the waiting sums have just left (np.nanmean(time_start),
    // worker.py:105, is one serial pairwise sum: it wants its T inputs an LDS read away)
    double* const tsl = R.tw();
#pragma nounroll
    for (int t = lane; t < T; t += WAVE) { const double v = R.ts()[t]; tsl[t] = v; if (time_start) time_start[(size_t)e * T + t] = v; }
    WSYNC();
    const double m2 = psum<4>(tsl, T) / Td;
    if (lane == 0) {
        double* row = summary + (size_t)e * 8;
        row[0] = -now; row[1] = (double)nfin; row[2] = (double)nfin / Td; row[3] = now;
        row[4] = m2; row[5] = m3; row[6] = m4; row[7] = m5;
        if (steps_out) steps_out[e] = steps;
        if (flags_out) flags_out[e] = flags | DCM_FLAG_DONE | (finished_flag ? DCM_FLAG_FINISHED : 0u);
    }
#pragma nounroll
    for (int a = lane; a < A; a += WAVE) {
        const size_t o = (size_t)e * A + a;
        if (agent_wait) agent_wait[o] = R.aw()[a];
        if (travel_dist) travel_dist[o] = R.tdist()[a];
        if (returned) returned[o] = (R.ainfo()[a] & A_RETURNED) ? 1 : 0;
    }
#ifdef DCM_REPLAY_PHASES
    WSYNC();
    if (time_start && lane < 12) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 12; q++) if (lane == q) v = (double)rph_acc[q];
        time_start[(size_t)e * T + lane] = v;
    }
#endif
}

#include "replay_fast.hpp"

}  // namespace

extern "C" {

int dcm_load_routes(dcm_env* env, const int32_t* routes, const int32_t* route_len, int32_t route_cap, int32_t member_cap,
                    void* stream) {
    CHECK_ENV(env);
    if (!routes || !route_len || route_cap < 1) return fail(DCM_ERR_INVALID, "dcm_load_routes: bad argument");
    if (member_cap < 1 || member_cap > MR_MAX) return fail(DCM_ERR_INVALID, "dcm_load_routes: member_cap must be in 1..32");
    if (env->L.C != DCM_MAX_MEMBERS) return fail(DCM_ERR_STATE, "dcm_load_routes: not on a DCM_PARAM_WIDE_MEMBERS handle (route replay has its own member_cap)");
    if (replay_lds_bytes(env->A, env->T, member_cap) > 160 * 1024)
        return fail(DCM_ERR_INVALID, "dcm_load_routes: replay state does not fit the 160 KiB LDS; lower member_cap");
    const size_t nr = (size_t)env->p.n_envs * env->A * route_cap, nl = (size_t)env->p.n_envs * env->A;   // (CHECK_ENV: the handle's device is current)
    if (env->routes) { (void)hipFree(env->routes); env->routes = nullptr; }
    if (env->route_len) { (void)hipFree(env->route_len); env->route_len = nullptr; }
    if (env->rmarr) { (void)hipFree(env->rmarr); env->rmarr = nullptr; }
    HIP_TRY(hipMalloc((void**)&env->rmarr, (size_t)env->p.n_envs * replay_scratch_bytes(env->A, env->T, member_cap)));
    HIP_TRY(hipMalloc((void**)&env->routes, nr * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&env->route_len, nl * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(env->routes, routes, nr * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(env->route_len, route_len, nl * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    env->route_cap = route_cap;
    env->member_cap = member_cap;
    return DCM_OK;
}

int dcm_set_visibility(dcm_env* env, int32_t initial, int32_t batch, int32_t period, int32_t cap) {
    CHECK_HANDLE(env);
    if (initial < 0 || batch < 1 || period < 1 || cap < initial)
        return fail(DCM_ERR_INVALID, "dcm_set_visibility: need initial >= 0, batch >= 1, period >= 1, cap >= initial");
    env->vis[0] = initial; env->vis[1] = batch; env->vis[2] = period; env->vis[3] = cap;
    return DCM_OK;
}

int dcm_set_replay_placement(dcm_env* env, int32_t placement) {
    CHECK_HANDLE(env);
    if (placement < 0 || placement > 2) return fail(DCM_ERR_INVALID, "dcm_set_replay_placement: 0 = auto, 1 = LDS, 2 = HBM");
    env->replay_placement = placement;
    return DCM_OK;
}

int dcm_execute_routes(dcm_env* env, int32_t reactive, int64_t* steps_out, uint32_t* flags_out, uint8_t* finished,
                       double* time_start, double* time_finish, double* task_wait, int32_t* n_members,
                       double* agent_wait, double* travel_dist, uint8_t* returned, void* stream) {
    CHECK_ENV(env);
    { const int rc_ = dcm::flush_pending(env, stream); if (rc_ != DCM_OK) return rc_; }
    if (!env->loaded) return fail(DCM_ERR_STATE, "dcm_execute_routes: call dcm_load_instances first");
    if (!env->routes) return fail(DCM_ERR_STATE, "dcm_execute_routes: call dcm_load_routes first");
    if (env->sizes) return fail(DCM_ERR_STATE, "dcm_execute_routes: route replay needs a uniform batch (dcm_load_instances)");
    // Where the replay scratch block lives: in LDS when the whole batch is resident with at most one wave per SIMD anyway (<= 4
    // envs per CU) and it fits a quarter of the CU's LDS, else in HBM (14 instead of 4 resident waves per CU at 100A/500T).
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, env->p.device));
    const uint32_t lds_in = replay_lds_bytes(env->A, env->T, env->member_cap, true);
    bool slds = lds_in <= 40u * 1024u && env->p.n_envs <= 4 * cus;
    if (env->replay_placement == 1) slds = lds_in <= 160u * 1024u;
    if (env->replay_placement == 2) slds = false;
    const uint32_t lds = slds ? lds_in : replay_lds_bytes(env->A, env->T, env->member_cap, false);

    RP P{100.0, 200.0, reactive ? 1 : 0, env->vis[0], env->vis[1], env->vis[2], env->vis[3]};  // env/task_env.py:564-565,567
    // The register-resident kernel (replay_fast.hpp) for every replay whose agents fit two lane chunks, whose LIVE tasks -- all
    // of them without dynamic arrivals, tasks 1..cap with them (an agent is never sent to a task that is not visible yet, and
    // visible <= cap: env/task_env.py:567,578-584) -- fit two lane chunks and whose member slots fit one id word: BASELINE
    // config 5 (100A/500T at the reference's cap of 100) and every small shape.  An explicit replay placement (1 / 2) asks for
    // the general kernel, whose scratch block it places.
    {
        const int TL = reactive ? (env->T < env->vis[3] ? env->T : env->vis[3]) : env->T;
        const uint32_t flds = replay_fast_lds_bytes(env->A, env->T, env->route_cap);   // (route_cap < 32768: the cursor and the length share a word)
        if (env->replay_placement == 0 && env->A <= 2 * WAVE && TL <= 2 * WAVE && env->member_cap <= 8 && flds <= 64u * 1024u && env->route_cap < 32768) {
#define REPLAYF(CMR, RE)                                                                                                     \
    do {                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)k_replay_fast<2, 2, CMR, RE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds); \
        hipLaunchKernelGGL((k_replay_fast<2, 2, CMR, RE>), GRID(env), flds, (hipStream_t)stream, env->A, env->T, TL, env->L.A, env->L.T, \
                           env->member_cap, P, env->state, env->routes, env->route_len, env->route_cap, env->summary,      \
                           steps_out, flags_out, finished, time_start, time_finish, task_wait, n_members, agent_wait,       \
                           travel_dist, returned, env->gscratch);                                                          \
    } while (0)
            if (env->member_cap <= 5) { if (reactive) REPLAYF(5, true); else REPLAYF(5, false); }
            else { if (reactive) REPLAYF(8, true); else REPLAYF(8, false); }
#undef REPLAYF
            LAUNCH_OK();
            return DCM_OK;
        }
    }
#define REPLAY(CA, CT, CMR, SL)                                                                                              \
    do {                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)k_replay<CA, CT, CMR, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_replay<CA, CT, CMR, SL>), GRID(env), lds, (hipStream_t)stream, env->A, env->T, env->L.A, env->L.T,  \
                           env->member_cap, P, env->state, env->routes, env->route_len, env->route_cap, env->summary,      \
                           steps_out, flags_out, finished, time_start, time_finish, task_wait, n_members, agent_wait,       \
                           travel_dist, returned, env->gscratch, env->rmarr);                                              \
    } while (0)
    const bool base5 = env->A == 100 && env->T == 500 && env->member_cap == 5;   // BASELINE config 5
    if (base5 && slds) REPLAY(100, 500, 5, true);
    else if (base5) REPLAY(100, 500, 5, false);
    else if (slds) REPLAY(0, 0, 0, true);
    else REPLAY(0, 0, 0, false);
#undef REPLAY
    LAUNCH_OK();
    return DCM_OK;
}

}  // extern "C"
