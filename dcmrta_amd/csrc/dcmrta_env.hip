// dcmrta_env.hip -- MI355X (gfx950) batched coalition-formation + routing environment.
//
// One wavefront (64 lanes) per env instance, one env per 64-thread workgroup.  The env's
// canonical record (S = 64 + 48*A + 96*T bytes, SURVEY.md §8d / DESIGN.md) is copied
// HBM -> LDS with 16-byte-per-lane coalesced loads, every phase of the reference state
// machine then runs wave-parallel on the LDS copy (lanes stride over tasks for
// task_update / mask / task observation and over agents for agent_update / agent
// observation / next_decision; wave-wide reductions use ballots and cross-lane shuffles),
// and the mutable part of the record is copied back.  The persistent rollout kernel keeps
// the record in LDS for whole episodes.  All times and positions are fp64 with the
// reference's operation order (compile with -ffp-contract=off); observations are rounded to
// fp32 exactly where the reference casts (worker.py:62,64).  No MFMA: the path is
// elementwise/reduction work, not a dense contraction.
//
// Reference restated: env/task_env.py (TaskEnv) and worker.py:41-112 (the rollout loop).
// Every device function cites the lines it follows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/dcmrta_env.h"

namespace {

constexpr int WAVE = 64;
constexpr int M = DCM_MAX_MEMBERS;
constexpr int AW = (DCM_MAX_AGENTS + 63) / 64;  // 64-bit words of an agent bitmask

// ---------------------------------------------------------------------------------- record
// Header of an env record (64 B).
struct Hdr {
    double now;            // current_time, env/task_env.py:28
    uint64_t seed;         // choice-protocol seed of this env
    uint64_t d;            // running decision counter (key of the choice protocol)
    double depot_x, depot_y;  // depot['location'] :111
    uint32_t flags;        // DCM_FLAG_*
    int32_t cur_group;     // 1-based index of the group now deciding (worker.py:52), 0 = none
    int32_t n_groups;      // groups of the current event (env/task_env.py:291-298)
    int32_t empty_passes;  // consecutive zero-decider events (guard)
    uint32_t ep_steps;     // decisions in the current episode
    uint32_t episodes;     // finished episodes since dcm_reset
};
static_assert(sizeof(Hdr) == 64, "header must be 64 bytes");

// ainfo[a]: bit0 returned, bit1 assigned, bit2 in depot['members'], bits 8-15 pending group id,
//           bits 16-31 number of times the agent was moved to an abandoned_agent list
constexpr uint32_t A_RETURNED = 1u, A_ASSIGNED = 2u, A_INDEPOT = 4u;
// tinfo[t]: bits 0-7 requirements, 8-15 status (int8, may be stale: quirk Q3), 16-23 len(members),
//           bit 24 feasible_assignment, bit 25 finished
constexpr uint32_t T_FEAS = 1u << 24, T_FIN = 1u << 25;

struct Layout {
    int32_t A, T;
    // mutable part
    uint32_t o_ax, o_ay, o_arr, o_nd, o_tdist;  // f64[A]: location, arrival_time[-1], next_decision, travel_dist
    uint32_t o_cur, o_ainfo;                    // i32[A] route[-1] (-2 none, -1 depot), u32[A]
    uint32_t o_ts, o_tf;                        // f64[T] time_start, time_finish
    uint32_t o_marr;                            // f64[M][T] latest arrival of member slot j of task t
    uint32_t o_mids;                            // u64[T]   byte j = agent id of member slot j (ordered, Q1)
    uint32_t o_tinfo, o_tnab;                   // u32[T], u32[T] len(abandoned_agent)
    uint32_t mut_bytes;                         // 16-aligned size of the mutable part
    // constant part (instance)
    uint32_t o_tx, o_ty, o_tdur;                // f64[T] location, time
    uint32_t rec_bytes;                         // 16-aligned record stride
    uint32_t lds_bytes;                         // record + scratch (task_wait[T], agent_wait[A])
    uint32_t o_tw, o_aw;                        // scratch offsets (LDS only)
};

struct KP {
    double mwt;       // max_waiting_time
    double max_time;  // MAX_TIME
};

struct Env {
    double *ax, *ay, *arr, *nd, *tdist, *ts, *tf, *marr, *tx, *ty, *tdur, *tw, *aw;
    uint64_t* mids;
    int32_t* cur;
    uint32_t *ainfo, *tinfo, *tnab;
    int A, T;
};

__device__ __forceinline__ Env make_env(unsigned char* b, const Layout& L) {
    Env E;
    E.A = L.A; E.T = L.T;
    E.ax = (double*)(b + L.o_ax); E.ay = (double*)(b + L.o_ay); E.arr = (double*)(b + L.o_arr);
    E.nd = (double*)(b + L.o_nd); E.tdist = (double*)(b + L.o_tdist);
    E.cur = (int32_t*)(b + L.o_cur); E.ainfo = (uint32_t*)(b + L.o_ainfo);
    E.ts = (double*)(b + L.o_ts); E.tf = (double*)(b + L.o_tf); E.marr = (double*)(b + L.o_marr);
    E.mids = (uint64_t*)(b + L.o_mids); E.tinfo = (uint32_t*)(b + L.o_tinfo); E.tnab = (uint32_t*)(b + L.o_tnab);
    E.tx = (double*)(b + L.o_tx); E.ty = (double*)(b + L.o_ty); E.tdur = (double*)(b + L.o_tdur);
    E.tw = (double*)(b + L.o_tw); E.aw = (double*)(b + L.o_aw);
    return E;
}

#define WSYNC() __syncthreads() /* 64-thread workgroup: lowers to a wave barrier + LDS wait */

// ---------------------------------------------------------------------------------- choice protocol
constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ULL;
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint64_t d, uint32_t slot) {
    return mix64(mix64(seed + GAMMA * (d + 1)) + GAMMA * (uint64_t)(slot + 1));
}
// x % n for 1 <= n < 65536 with 32-bit arithmetic only
__device__ __forceinline__ uint32_t mod_small(uint64_t x, uint32_t n) {
    uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    uint32_t c = (0xFFFFFFFFu % n + 1u) % n;  // 2^32 mod n
    return ((hi % n) * c + (lo % n)) % n;
}

// ---------------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o); v = (w < v) ? w : v; }
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o); v = (w > v) ? w : v; }
    return v;
}
// position of the idx-th (0-based) set bit of a wave-uniform mask; all 64 lanes must call
__device__ __forceinline__ int nth_set_bit(uint64_t m, int idx, int lane) {
    const bool b = (m >> lane) & 1ull;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const uint64_t sel = __ballot(b && rank == idx);
    return __ffsll((unsigned long long)sel) - 1;
}

struct AMask { uint64_t w[AW]; };
__device__ __forceinline__ int amask_count(const AMask& m) { int n = 0;
#pragma unroll
    for (int i = 0; i < AW; i++) n += __popcll(m.w[i]);
    return n; }
__device__ __forceinline__ bool amask_test(const AMask& m, int a) { return (m.w[a >> 6] >> (a & 63)) & 1ull; }
__device__ __forceinline__ void amask_clear(AMask& m, int a) {
#pragma unroll
    for (int i = 0; i < AW; i++) if (i == (a >> 6)) m.w[i] &= ~(1ull << (a & 63));
}
__device__ __forceinline__ int amask_nth(const AMask& m, int idx, int lane) {
    int base = 0, res = -1;
#pragma unroll
    for (int i = 0; i < AW; i++) {
        const int c = __popcll(m.w[i]);
        const int p = nth_set_bit(m.w[i], idx - base, lane);  // -1 when idx-base is out of this word's range
        if (res < 0 && idx - base >= 0 && idx - base < c) res = i * 64 + p;
        base += c;
    }
    return res;
}
// agents whose pending group id equals g
__device__ __forceinline__ AMask group_mask(const Env& E, int g, int lane) {
    AMask m;
#pragma unroll
    for (int i = 0; i < AW; i++) {
        const int a = i * 64 + lane;
        const bool in = (a < E.A) && (int)((E.ainfo[a < E.A ? a : 0] >> 8) & 0xFFu) == g;
        m.w[i] = __ballot(in);
    }
    return m;
}

// ---------------------------------------------------------------------------------- numpy add.reduce
// np.sum / np.mean use pairwise summation (8 accumulators per <=128-element block, recursive
// halving above); restated so the perf metrics of worker.py:103-108 are bit-identical.
__device__ double psum_block(const double* a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
template <int DEPTH>
__device__ double psum(const double* a, int n) {
    if constexpr (DEPTH == 0) {
        return psum_block(a, n);
    } else {
        if (n <= 128) return psum_block(a, n);
        int n2 = n / 2;
        n2 -= n2 % 8;
        return psum<DEPTH - 1>(a, n2) + psum<DEPTH - 1>(a + n2, n - n2);
    }
}

// ---------------------------------------------------------------------------------- primitives
// env/task_env.py:161-163 -- np.linalg.norm of a 2-vector == sqrt(fma(dy,dy,dx*dx)) on the
// reference machine (tests/golden/distance_kat.npz).
__device__ __forceinline__ double dist2(double ax, double ay, double bx, double by) {
    const double dx = ax - bx, dy = ay - by;
    return sqrt(__builtin_fma(dy, dy, dx * dx));
}

// ---------------------------------------------------------------------------------- task_update
// env/task_env.py:245-281.  Lanes stride over tasks; each lane walks its task's <=M ordered
// member slots.  The "coalition capability-vs-requirement reduction" is status = req - len(members).
__device__ void task_update(Env& E, const Hdr& h, const KP& P, int lane) {
    const double now = h.now, mwt = P.mwt;
    bool allf = true;
    for (int t = lane; t < E.T; t += WAVE) {
        uint32_t info = E.tinfo[t];
        if (!(info & T_FEAS)) {                                            // :249
            const int req = info & 0xFF;
            const int n = (info >> 16) & 0xFF;                             // :250
            const uint64_t ids = E.mids[t];
            double av[M];
#pragma unroll
            for (int j = 0; j < M; j++) av[j] = (j < n) ? E.marr[j * E.T + t] : 0.0;  // :251
            const int status = req - n;                                    // :252
            uint32_t keep = 0;                                             // bit j: slot j stays a member
            bool changed = false;
            if (status <= 0) {                                             // :254
                double mx = av[0], mn = av[0];
#pragma unroll
                for (int j = 1; j < M; j++) if (j < n) { mx = av[j] > mx ? av[j] : mx; mn = av[j] < mn ? av[j] : mn; }
                if (mx - mn <= mwt) {                                      // :255
                    E.ts[t] = mx;                                          // :256
                    E.tf[t] = mx + E.tdur[t];                              // :257
                    info |= T_FEAS;                                        // :258
                    keep = (1u << n) - 1u;
                } else {
                    const double thr = mx - mwt;                           // :262
#pragma unroll
                    for (int j = 0; j < M; j++) if (j < n) { if (av[j] <= thr) changed = true; else keep |= 1u << j; }
                }
            } else {
                // :268-271 iterates task['members'] while removing from it: after a removal the
                // element that slides into the freed slot is skipped by the list iterator (Q1).
                bool skip = false;
#pragma unroll
                for (int j = 0; j < M; j++) if (j < n) {
                    if (skip) { keep |= 1u << j; skip = false; }
                    else if (now - av[j] >= mwt) { changed = true; skip = true; }   // :269
                    else keep |= 1u << j;
                }
            }
            int nn = n;
            if (changed) {
                uint64_t nids = 0;
                int k = 0, dropped = 0;
#pragma unroll
                for (int j = 0; j < M; j++) if (j < n) {
                    const uint32_t id = (uint32_t)((ids >> (8 * j)) & 0xFF);
                    if (keep & (1u << j)) {
                        nids |= (uint64_t)id << (8 * k);
                        E.marr[k * E.T + t] = av[j];
                        k++;
                    } else {
                        atomicAdd(&E.ainfo[id], 1u << 16);                 // :265/:271 abandoned_agent.append(member)
                        dropped++;
                    }
                }
                E.mids[t] = nids;
                E.tnab[t] += (uint32_t)dropped;
                nn = k;
            }
            info = (info & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)nn << 16);
        } else {
            if (now >= E.tf[t]) info |= T_FIN;                             // :273-274
        }
        E.tinfo[t] = info;
        allf = allf && (info & T_FEAS);
    }
    const bool all_feasible = __all(allf);
    WSYNC();
    // depot :277-280
    for (int a = lane; a < E.A; a += WAVE) {
        uint32_t ai = E.ainfo[a];
        if ((ai & A_INDEPOT) && now >= E.arr[a] && all_feasible) E.ainfo[a] = ai | A_RETURNED;
    }
}

// ---------------------------------------------------------------------------------- agent_update
// env/task_env.py:207-243 (non-reactive branch :226)
__device__ void agent_update(Env& E, const Hdr& h, const KP& P, int lane) {
    const double now = h.now;
    for (int a = lane; a < E.A; a += WAVE) {
        const int c = E.cur[a];
        if (c == -2) continue;                                             // :209 no arrival yet
        if (c == -1) { E.nd[a] = __builtin_nan(""); continue; }            // :212,:226
        const uint32_t info = E.tinfo[c];                                  // :228
        uint32_t ai = E.ainfo[a];
        bool member = false;
        if (info & T_FEAS) {                                               // :229
            const int n = (info >> 16) & 0xFF;
            const uint64_t ids = E.mids[c];
#pragma unroll
            for (int j = 0; j < M; j++) if (j < n && (int)((ids >> (8 * j)) & 0xFF) == a) member = true;  // :230
        }
        if (member) {
            E.nd[a] = E.tf[c];                                             // :231
            if (now >= E.ts[c]) ai |= A_ASSIGNED;                          // :232-233
        } else {
            E.nd[a] = E.arr[a] + P.mwt;                                    // :235 / :238
            ai &= ~A_ASSIGNED;                                             // :236 / :240
        }
        E.ainfo[a] = ai;
    }
}

// ---------------------------------------------------------------------------------- terminal
// calculate_waiting_time (env/task_env.py:344-364) into LDS scratch tw[T], aw[A].
__device__ void compute_waits(Env& E, const Hdr& h, const KP& P, int lane) {
    const double now = h.now, mwt = P.mwt;
    for (int t = lane; t < E.T; t += WAVE) {
        const uint32_t info = E.tinfo[t];
        const int n = (info >> 16) & 0xFF;
        const double ab = (double)E.tnab[t] * mwt;
        double s = 0.;
        if (n != 0) {                                                      // :349
            double mx = E.marr[t];
            for (int j = 1; j < n; j++) { const double v = E.marr[j * E.T + t]; mx = v > mx ? v : mx; }
            if (info & T_FEAS) { for (int j = 0; j < n; j++) s += mx - E.marr[j * E.T + t]; }    // :351
            else { for (int j = 0; j < n; j++) s += now - E.marr[j * E.T + t]; }                // :354
        }
        E.tw[t] = s + ab;                                                  // :351-357
    }
    // :358-364, per agent in task order.  The +max_waiting_time terms of abandoned entries are added
    // as count*mwt after the member terms (the reference interleaves them in task order): equal to
    // within a few ulp, see DESIGN.md "documented deviations".
    for (int a = lane; a < E.A; a += WAVE) {
        double s = 0.;
        for (int t = 0; t < E.T; t++) {
            const uint32_t info = E.tinfo[t];
            const int n = (info >> 16) & 0xFF;
            if (n == 0) continue;
            const uint64_t ids = E.mids[t];
            int pos = -1;
            for (int j = 0; j < n; j++) if ((int)((ids >> (8 * j)) & 0xFF) == a) pos = j;
            if (pos < 0) continue;
            const double mine = E.marr[pos * E.T + t];
            if (info & T_FEAS) {
                double mx = E.marr[t];
                for (int j = 1; j < n; j++) { const double v = E.marr[j * E.T + t]; mx = v > mx ? v : mx; }
                s += mx - mine;                                            // :360
            } else {
                const double w = now - mine;
                s += (w > 0.) ? w : 0.;                                    // :362
            }
        }
        s += (double)(E.ainfo[a] >> 16) * mwt;                             // :363-364
        E.aw[a] = s;
    }
    WSYNC();
}

// get_episode_reward + perf metrics (env/task_env.py:420-425, worker.py:87,103-108) -> row[8]
__device__ void terminal(Env& E, Hdr& h, const KP& P, int lane, double* __restrict__ row) {
    WSYNC();
    compute_waits(E, h, P, lane);
    int nfin = 0;
    for (int t0 = 0; t0 < E.T; t0 += WAVE) {
        const int t = t0 + lane;
        nfin += __popcll(__ballot(t < E.T && (E.tinfo[t < E.T ? t : 0] & T_FIN)));
    }
    // :422 check_finished() once more can only re-assign the same `now`
    const double T_ = (double)E.T, A_ = (double)E.A;
    const double m2 = psum<3>(E.ts, E.T) / T_;     // np.nanmean(time_start)      worker.py:105
    const double m3 = psum<3>(E.aw, E.A) / A_;     // np.mean(agent sum_waiting)  :106
    const double m4 = psum<3>(E.tdist, E.A);       // np.sum(travel_dist)         :107
    const double m5 = psum<3>(E.tw, E.T) / T_;     // np.mean(task sum_waiting)   :108
    if (lane == 0 && row) {
        row[0] = -h.now;                           // reward, env/task_env.py:424
        row[1] = (double)nfin;
        row[2] = (double)nfin / T_;                // success_rate :103
        row[3] = h.now;                            // makespan :104
        row[4] = m2; row[5] = m3; row[6] = m4; row[7] = m5;
    }
    h.flags |= DCM_FLAG_DONE;
    h.episodes += 1;
    h.cur_group = 0;
}

// ---------------------------------------------------------------------------------- event loop
// Boxes D + A of SURVEY.md Appendix B: check_finished (worker.py:85, env/task_env.py:366-373), loop
// test (worker.py:45), next_decision (:283-289), get_unique_group (:291-298), task_update,
// agent_update (worker.py:50-51).  Returns at the next decision point or after terminal().
__device__ void advance(Env& E, Hdr& h, const KP& P, int lane, double* __restrict__ row) {
    const double INF = __builtin_inf();
    for (;;) {
        WSYNC();
        // ---- D: check_finished
        double tmin = INF, maxarr = 0.0;
        bool allret = true;
        for (int a = lane; a < E.A; a += WAVE) {
            const double v = E.nd[a];
            if (v == v) tmin = v < tmin ? v : tmin;                        // np.nanmin :287
            const double av = (E.cur[a] != -2) ? E.arr[a] : 0.0;           // max(arrival_time) or 0 :286
            maxarr = av > maxarr ? av : maxarr;
            allret = allret && (E.ainfo[a] & A_RETURNED);
        }
        tmin = wave_min(tmin);
        const bool any = tmin < INF;
        bool finished = false;
        if (!any) {                                                        // :368
            h.now = wave_max(maxarr);                                      // :369
            bool allfin = true;
            for (int t = lane; t < E.T; t += WAVE) allfin = allfin && (E.tinfo[t] & T_FIN);
            finished = __all(allret) && __all(allfin);                     // :370
        }
        if (finished) h.flags |= DCM_FLAG_FINISHED;
        if (finished || h.now >= P.max_time) { terminal(E, h, P, lane, row); return; }   // worker.py:45
        // ---- A: new event
        h.n_groups = 0;
        if (any) {
            h.now = tmin;                                                  // worker.py:49
            // deciding set: exact equality with the minimum (env/task_env.py:288)
            bool dec[AW];
            uint64_t dm[AW];
#pragma unroll
            for (int i = 0; i < AW; i++) {
                const int a = i * 64 + lane;
                dec[i] = (a < E.A) && (E.nd[a < E.A ? a : 0] == tmin);
                dm[i] = __ballot(dec[i]);
            }
            // fast path: every deciding agent stands on the same (x,y) -> one group
            int first = -1;
#pragma unroll
            for (int i = AW - 1; i >= 0; i--) if (dm[i]) first = i * 64 + __ffsll((unsigned long long)dm[i]) - 1;
            const double x0 = E.ax[first], y0 = E.ay[first];
            bool same = true;
#pragma unroll
            for (int i = 0; i < AW; i++) {
                const int a = i * 64 + lane;
                if (dec[i]) same = same && (E.ax[a] == x0) && (E.ay[a] == y0);
            }
            if (__all(same)) {
#pragma unroll
                for (int i = 0; i < AW; i++) {
                    const int a = i * 64 + lane;
                    if (a < E.A) E.ainfo[a] = (E.ainfo[a] & ~0xFF00u) | (dec[i] ? (1u << 8) : 0u);
                }
                h.n_groups = 1;
            } else {
                // general: groups in ascending (x, then y) order == rows of np.unique(axis=0) :293
#pragma unroll
                for (int i = 0; i < AW; i++) {
                    const int a = i * 64 + lane;
                    if (a < E.A) E.ainfo[a] = (E.ainfo[a] & ~0xFF00u) | (dec[i] ? 0xFF00u : 0u);
                }
                int g = 0;
                for (;;) {
                    double mx = INF;
#pragma unroll
                    for (int i = 0; i < AW; i++) {
                        const int a = i * 64 + lane;
                        if (a < E.A && (E.ainfo[a] & 0xFF00u) == 0xFF00u) mx = E.ax[a] < mx ? E.ax[a] : mx;
                    }
                    mx = wave_min(mx);
                    if (!(mx < INF)) break;
                    double my = INF;
#pragma unroll
                    for (int i = 0; i < AW; i++) {
                        const int a = i * 64 + lane;
                        if (a < E.A && (E.ainfo[a] & 0xFF00u) == 0xFF00u && E.ax[a] == mx) my = E.ay[a] < my ? E.ay[a] : my;
                    }
                    my = wave_min(my);
                    g++;
#pragma unroll
                    for (int i = 0; i < AW; i++) {
                        const int a = i * 64 + lane;
                        if (a < E.A && (E.ainfo[a] & 0xFF00u) == 0xFF00u && E.ax[a] == mx && E.ay[a] == my)
                            E.ainfo[a] = (E.ainfo[a] & ~0xFF00u) | ((uint32_t)g << 8);
                    }
                }
                h.n_groups = g;
            }
        }
        WSYNC();
        task_update(E, h, P, lane);                                        // worker.py:50
        WSYNC();
        agent_update(E, h, P, lane);                                       // worker.py:51
        if (!any) {
            if (++h.empty_passes > 4) { h.flags |= DCM_FLAG_TRUNCATED; terminal(E, h, P, lane, row); return; }
            continue;
        }
        h.empty_passes = 0;
        h.cur_group = 1;
        WSYNC();
        return;
    }
}

// reset + clear_decisions (env/task_env.py:116-140); keeps seed, d, episodes
__device__ void reset_state(Env& E, Hdr& h, int lane) {
    for (int t = lane; t < E.T; t += WAVE) {
        const uint32_t req = E.tinfo[t] & 0xFF;
        E.tinfo[t] = req | (req << 8);       // status = requirements :131, members [] , not feasible/finished
        E.tnab[t] = 0;
        E.mids[t] = 0;
        E.ts[t] = 0.0; E.tf[t] = 0.0;
    }
    for (int a = lane; a < E.A; a += WAVE) {
        E.ax[a] = h.depot_x; E.ay[a] = h.depot_y;  // :134
        E.arr[a] = 0.0; E.nd[a] = 0.0; E.tdist[a] = 0.0;  // :135
        E.cur[a] = -2; E.ainfo[a] = 0;
    }
    h.now = 0.0; h.flags = 0; h.cur_group = 0; h.n_groups = 0; h.empty_passes = 0; h.ep_steps = 0;  // :139-140
}

// ---------------------------------------------------------------------------------- decisions
// worker.py:54 -- the deciding agent of the current group (protocol slot 0), or the injected one
__device__ int pick_leader(const Env& E, Hdr& h, int lane, int leader_in, AMask& gm) {
    gm = group_mask(E, h.cur_group, lane);
    const int glen = amask_count(gm);
    if (glen == 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return -1; }  // unreachable: groups are never empty
    if (leader_in >= 0) {
        if (leader_in >= E.A || !amask_test(gm, leader_in)) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return -1; }
        return leader_in;
    }
    const int idx = (int)mod_small(draw(h.seed, h.d, 0), (uint32_t)glen);
    return amask_nth(gm, idx, lane);
}

// worker.py:57-68: mask + both observation tensors relative to `leader`, straight into the policy's
// input tensors (fp32 casts of worker.py:62,64).  Returns true when every task is masked.
__device__ bool observe(const Env& E, const Hdr& h, int lane, int leader, float* __restrict__ ag,
                        float* __restrict__ tk, uint8_t* __restrict__ mask) {
    const double now = h.now;
    const double lx = E.ax[leader], ly = E.ay[leader];
    // get_current_agent_status, env/task_env.py:165-180
    if (ag) {
        for (int a = lane; a < E.A; a += WAVE) {
            double travel = 0., waiting = 0., remaining = 0.;
            const int c = E.cur[a];
            if (c >= 0) {                                                  // :168
                const double av = E.arr[a], tsK = E.ts[c];
                const double x = av - now; travel = x > 0. ? x : 0.;       // :169
                if (now <= tsK) { const double w = now - av; waiting = w > 0. ? w : 0.; }               // :170
                if (now >= tsK) { const double r = tsK + E.tdur[c] - now; remaining = r > 0. ? r : 0.; } // :171
            }
            float* row = ag + 6 * a;                                       // :176-177
            row[0] = (float)travel; row[1] = (float)remaining; row[2] = (float)waiting;
            row[3] = (float)(lx - E.ax[a]); row[4] = (float)(ly - E.ay[a]);
            row[5] = (E.ainfo[a] & A_ASSIGNED) ? 1.f : 0.f;
        }
    }
    // get_current_task_status :182-190 and get_unfinished_task_mask :192-200
    bool allmasked = true;
    for (int t = lane; t < E.T; t += WAVE) {
        const uint32_t info = E.tinfo[t];
        const int status = (int)(int8_t)((info >> 8) & 0xFF);
        const bool unfinished = !(info & T_FEAS) && status > 0;            // :199
        allmasked = allmasked && !unfinished;
        if (mask) mask[t + 1] = unfinished ? 0 : 1;                        // :193
        if (tk) {
            float* row = tk + 5 * (t + 1);                                 // :185-186
            row[0] = (float)status; row[1] = (float)(info & 0xFF); row[2] = (float)E.tdur[t];
            row[3] = (float)(E.tx[t] - lx); row[4] = (float)(E.ty[t] - ly);
        }
    }
    allmasked = __all(allmasked);
    if (lane == 0) {
        if (mask) mask[0] = allmasked ? 0 : 1;                             // worker.py:58-61
        if (tk) { tk[0] = 0.f; tk[1] = 0.f; tk[2] = 0.f; tk[3] = (float)(h.depot_x - lx); tk[4] = (float)(h.depot_y - ly); }  // :188
    }
    return allmasked;
}

// uniform-random valid action (protocol slot 1): valid = ascending unmasked action ids
__device__ int pick_random_action(const Env& E, const Hdr& h, int lane) {
    int nv = 0;
    for (int t0 = 0; t0 < E.T; t0 += WAVE) {
        const int t = t0 + lane;
        const uint32_t info = E.tinfo[t < E.T ? t : 0];
        const bool un = (t < E.T) && !(info & T_FEAS) && ((int)(int8_t)((info >> 8) & 0xFF) > 0);
        nv += __popcll(__ballot(un));
    }
    if (nv == 0) return 0;  // only the depot is unmasked
    int idx = (int)mod_small(draw(h.seed, h.d, 1), (uint32_t)nv);
    int action = 0;
    for (int t0 = 0; t0 < E.T; t0 += WAVE) {
        const int t = t0 + lane;
        const uint32_t info = E.tinfo[t < E.T ? t : 0];
        const bool un = (t < E.T) && !(info & T_FEAS) && ((int)(int8_t)((info >> 8) & 0xFF) > 0);
        const uint64_t bm = __ballot(un);
        const int c = __popcll(bm);
        const int p = nth_set_bit(bm, idx, lane);
        if (action == 0 && idx >= 0 && idx < c) action = t0 + p + 1;
        idx -= c;
    }
    return action;
}

// TaskEnv.step (env/task_env.py:326-342) + agent_step (:300-324) for leader + followers, then
// task_update / agent_update (worker.py:74-76) and the move to the next decision point.
__device__ void apply_and_advance(Env& E, Hdr& h, const KP& P, int lane, int leader, const AMask& gm0, int action,
                                  int nfol_in, const int16_t* __restrict__ fol_in, double* __restrict__ row) {
    if (action < 0 || action > E.T) { h.flags |= DCM_FLAG_BAD_ACTION | DCM_FLAG_DONE; return; }
    AMask rest = gm0;
    amask_clear(rest, leader);                                             // :328 group.remove(leader)
    int rlen = amask_count(rest);
    AMask mm;                                                              // members of this step
#pragma unroll
    for (int i = 0; i < AW; i++) mm.w[i] = 0;
    mm.w[leader >> 6] |= 1ull << (leader & 63);
    int ml[M];                                                             // ordered: leader, followers
#pragma unroll
    for (int j = 0; j < M; j++) ml[j] = -1;
    ml[0] = leader;
    int nm = 1;
    double tx_, ty_;
    if (action == 0) {
        // vacancy = len(group) (:327): every co-located agent returns with the leader (Q9);
        // the draw order of the followers does not change any state, so no draws are spent.
#pragma unroll
        for (int i = 0; i < AW; i++) mm.w[i] |= rest.w[i];
        nm += rlen; rlen = 0;
        tx_ = h.depot_x; ty_ = h.depot_y;
    } else {
        const int k = action - 1;
        const int vacancy = (int)(int8_t)((E.tinfo[k] >> 8) & 0xFF);       // :327 task status (may be stale)
        int nf = (vacancy > 1) ? ((vacancy - 1 < rlen) ? vacancy - 1 : rlen) : 0;  // :330-331
        if (nfol_in >= 0) nf = nfol_in;
        if (nf > M - 1 || nf > rlen) { h.flags |= DCM_FLAG_OVERFLOW | DCM_FLAG_DONE; return; }
        for (int j = 0; j < nf; j++) {                                     // :331 choice without replacement
            int f;
            if (nfol_in >= 0) {
                f = fol_in[j];
                if (f < 0 || f >= E.A || !amask_test(rest, f)) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return; }
            } else {
                f = amask_nth(rest, (int)mod_small(draw(h.seed, h.d, 2 + j), (uint32_t)rlen), lane);
            }
            amask_clear(rest, f); rlen--;                                  // :332-333
#pragma unroll
            for (int i = 0; i < AW; i++) if (i == (f >> 6)) mm.w[i] |= 1ull << (f & 63);
#pragma unroll
            for (int q = 1; q < M; q++) if (q == nm) ml[q] = f;
            nm++;
        }
        tx_ = E.tx[k]; ty_ = E.ty[k];
    }
    // agent_step for every member (:300-324); independent per agent
#pragma unroll
    for (int i = 0; i < AW; i++) {
        const int a = i * 64 + lane;
        if (a < E.A && ((mm.w[i] >> lane) & 1ull)) {
            const double d = dist2(E.ax[a], E.ay[a], tx_, ty_);
            const double travel_time = d / 0.2;                            // :315 velocity 0.2 (:99)
            E.tdist[a] += d;                                               // :317
            E.arr[a] = h.now + travel_time;                                // :318
            E.ax[a] = tx_; E.ay[a] = ty_;                                  // :320
            E.cur[a] = action - 1;                                         // :314 route.append
            uint32_t ai = E.ainfo[a] & ~0xFF00u;                           // leaves the pending group
            if (action == 0) ai |= A_INDEPOT;                              // :321-322 depot['members']
            E.ainfo[a] = ai;
        }
    }
    WSYNC();
    if (action > 0) {
        // :321-322 members.append unless already listed; a re-joining agent keeps its slot but
        // get_arrival_time (:202-205) now returns the new, later arrival (Q4)
        const int k = action - 1;
        uint32_t info = E.tinfo[k];
        uint64_t ids = E.mids[k];
        int n = (info >> 16) & 0xFF;
        bool ovf = false;
#pragma unroll
        for (int j = 0; j < M; j++) if (j < nm) {
            const int m = ml[j];
            int pos = -1;
#pragma unroll
            for (int q = 0; q < M; q++) if (q < n && (int)((ids >> (8 * q)) & 0xFF) == m) pos = q;
            if (pos < 0) {
                if (n >= M) { ovf = true; }
                else { pos = n++; ids = (ids & ~(0xFFull << (8 * pos))) | ((uint64_t)m << (8 * pos)); }
            }
            if (pos >= 0 && lane == 0) E.marr[pos * E.T + k] = E.arr[m];
        }
        if (ovf) { h.flags |= DCM_FLAG_OVERFLOW | DCM_FLAG_DONE; return; }
        if (lane == 0) { E.mids[k] = ids; E.tinfo[k] = (info & ~0x00FF0000u) | ((uint32_t)n << 16); }
    }
    h.d += 1; h.ep_steps += 1;
    WSYNC();
    task_update(E, h, P, lane);                                            // worker.py:74
    WSYNC();
    agent_update(E, h, P, lane);                                           // worker.py:76
    WSYNC();
    if (rlen > 0) return;                                                  // worker.py:53 same group, next leader
    if (h.cur_group < h.n_groups) { h.cur_group++; return; }               // worker.py:52 next group
    advance(E, h, P, lane, row);                                           // worker.py:85 -> :45
}

// ---------------------------------------------------------------------------------- record I/O
__device__ __forceinline__ void copy16(unsigned char* dst, const unsigned char* src, uint32_t bytes, int lane) {
    const uint4* s = (const uint4*)src;
    uint4* d = (uint4*)dst;
    for (uint32_t i = lane; i < bytes / 16; i += WAVE) d[i] = s[i];
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// ---------------------------------------------------------------------------------- kernels
__global__ __launch_bounds__(WAVE) void k_load_instances(Layout L, unsigned char* state, const double* depot,
                                                        const double* task_xy, const int32_t* req, const double* dur) {
    const int e = blockIdx.x, lane = threadIdx.x;
    unsigned char* rec = state + (size_t)e * L.rec_bytes;
    Env E = make_env(rec, L);
    for (int t = lane; t < L.T; t += WAVE) {
        E.tx[t] = task_xy[((size_t)e * L.T + t) * 2];
        E.ty[t] = task_xy[((size_t)e * L.T + t) * 2 + 1];
        E.tdur[t] = dur[(size_t)e * L.T + t];
        E.tinfo[t] = (uint32_t)req[(size_t)e * L.T + t] & 0xFF;
    }
    if (lane == 0) {
        Hdr* h = (Hdr*)rec;
        h->depot_x = depot[2 * (size_t)e]; h->depot_y = depot[2 * (size_t)e + 1];
        h->flags = DCM_FLAG_DONE; h->episodes = 0; h->d = 0; h->seed = 0;
    }
}

__global__ __launch_bounds__(WAVE) void k_reset(Layout L, KP P, unsigned char* state, const uint64_t* seeds,
                                               double* summary) {
    const int e = blockIdx.x, lane = threadIdx.x;
    unsigned char* rec = state + (size_t)e * L.rec_bytes;
    copy16(smem, rec, L.rec_bytes, lane);
    WSYNC();
    Env E = make_env(smem, L);
    Hdr h = *(Hdr*)smem;
    h.seed = seeds[e]; h.d = 0; h.episodes = 0;
    reset_state(E, h, lane);
    if (lane < 8) summary[(size_t)e * 8 + lane] = __builtin_nan("");
    advance(E, h, P, lane, summary + (size_t)e * 8);
    WSYNC();
    if (lane == 0) *(Hdr*)smem = h;
    WSYNC();
    copy16(rec, smem, L.mut_bytes, lane);
}

__device__ void write_inactive_obs(const Env& E, int lane, float* ag, float* tk, uint8_t* mask) {
    if (ag) for (int i = lane; i < 6 * E.A; i += WAVE) ag[i] = 0.f;
    if (tk) for (int i = lane; i < 5 * (E.T + 1); i += WAVE) tk[i] = 0.f;
    if (mask) for (int i = lane; i <= E.T; i += WAVE) mask[i] = (i == 0) ? 0 : 1;
}

__global__ __launch_bounds__(WAVE) void k_observe(Layout L, unsigned char* state, float* agents_out, float* tasks_out,
                                                 uint8_t* mask_out, int32_t* leader_out, uint8_t* active_out,
                                                 const int32_t* leader_in) {
    const int e = blockIdx.x, lane = threadIdx.x;
    unsigned char* rec = state + (size_t)e * L.rec_bytes;
    copy16(smem, rec, L.rec_bytes, lane);
    WSYNC();
    Env E = make_env(smem, L);
    Hdr h = *(Hdr*)smem;
    float* ag = agents_out ? agents_out + (size_t)e * 6 * L.A : nullptr;
    float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (L.T + 1) : nullptr;
    uint8_t* mk = mask_out ? mask_out + (size_t)e * (L.T + 1) : nullptr;
    int leader = -1;
    if (!(h.flags & DCM_FLAG_DONE)) {
        AMask gm;
        leader = pick_leader(E, h, lane, leader_in ? leader_in[e] : -1, gm);
    }
    if (leader >= 0) observe(E, h, lane, leader, ag, tk, mk);
    else write_inactive_obs(E, lane, ag, tk, mk);
    if (lane == 0) {
        if (leader_out) leader_out[e] = leader;
        if (active_out) active_out[e] = leader >= 0 ? 1 : 0;
        if (leader < 0 && !(((Hdr*)rec)->flags & DCM_FLAG_DONE)) ((Hdr*)rec)->flags = h.flags;  // injected-leader error
    }
}

__global__ __launch_bounds__(WAVE) void k_step(Layout L, KP P, unsigned char* state, const int32_t* actions,
                                              const int32_t* leader_in, const int32_t* nfol_in, const int16_t* fol_in,
                                              float* agents_out, float* tasks_out, uint8_t* mask_out,
                                              int32_t* leader_out, uint8_t* active_out, double* summary) {
    const int e = blockIdx.x, lane = threadIdx.x;
    unsigned char* rec = state + (size_t)e * L.rec_bytes;
    copy16(smem, rec, L.rec_bytes, lane);
    WSYNC();
    Env E = make_env(smem, L);
    Hdr h = *(Hdr*)smem;
    const bool was_active = !(h.flags & DCM_FLAG_DONE);
    if (was_active) {
        AMask gm;
        const int leader = pick_leader(E, h, lane, leader_in ? leader_in[e] : -1, gm);
        if (leader >= 0) {
            const int nf = nfol_in ? nfol_in[e] : -1;
            apply_and_advance(E, h, P, lane, leader, gm, actions[e], nf, fol_in ? fol_in + (size_t)e * DCM_FOLLOWER_COLS : nullptr,
                              summary + (size_t)e * 8);
        }
    }
    const bool want_obs = agents_out || tasks_out || mask_out || leader_out || active_out;
    if (want_obs) {
        WSYNC();
        float* ag = agents_out ? agents_out + (size_t)e * 6 * L.A : nullptr;
        float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (L.T + 1) : nullptr;
        uint8_t* mk = mask_out ? mask_out + (size_t)e * (L.T + 1) : nullptr;
        int leader = -1;
        if (!(h.flags & DCM_FLAG_DONE)) { AMask gm; leader = pick_leader(E, h, lane, -1, gm); }
        if (leader >= 0) observe(E, h, lane, leader, ag, tk, mk);
        else write_inactive_obs(E, lane, ag, tk, mk);
        if (lane == 0) {
            if (leader_out) leader_out[e] = leader;
            if (active_out) active_out[e] = leader >= 0 ? 1 : 0;
        }
    }
    if (was_active) {
        WSYNC();
        if (lane == 0) *(Hdr*)smem = h;
        WSYNC();
        copy16(rec, smem, L.mut_bytes, lane);
    }
}

// Config-2 hot path: whole episodes in one persistent launch, record resident in LDS.
__global__ __launch_bounds__(WAVE) void k_rollout_random(Layout L, KP P, unsigned char* state, int episodes,
                                                        float* agents_out, float* tasks_out, uint8_t* mask_out,
                                                        int64_t* steps_out, double* summary) {
    const int e = blockIdx.x, lane = threadIdx.x;
    unsigned char* rec = state + (size_t)e * L.rec_bytes;
    copy16(smem, rec, L.rec_bytes, lane);
    WSYNC();
    Env E = make_env(smem, L);
    Hdr h = *(Hdr*)smem;
    float* ag = agents_out ? agents_out + (size_t)e * 6 * L.A : nullptr;
    float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (L.T + 1) : nullptr;
    uint8_t* mk = mask_out ? mask_out + (size_t)e * (L.T + 1) : nullptr;
    double* row = summary + (size_t)e * 8;
    int64_t steps = 0;
    for (int ep = 0; ep < episodes; ep++) {
        if (h.flags & DCM_FLAG_DONE) {  // restart from the loaded instance; d keeps running
            const uint32_t err = h.flags & (DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER);
            if (err) break;
            reset_state(E, h, lane);
            advance(E, h, P, lane, row);
        }
        while (!(h.flags & DCM_FLAG_DONE)) {
            AMask gm;
            const int leader = pick_leader(E, h, lane, -1, gm);
            if (leader < 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; break; }
            observe(E, h, lane, leader, ag, tk, mk);
            const int action = pick_random_action(E, h, lane);
            apply_and_advance(E, h, P, lane, leader, gm, action, -1, nullptr, row);
            steps++;
        }
    }
    if (lane == 0 && steps_out) steps_out[e] = steps;
    WSYNC();
    if (lane == 0) *(Hdr*)smem = h;
    WSYNC();
    copy16(rec, smem, L.mut_bytes, lane);
}

__global__ __launch_bounds__(WAVE) void k_env_status(Layout L, const unsigned char* state, int B, uint32_t* flags_out,
                                                    int64_t* dec_out, double* now_out) {
    const int e = blockIdx.x * WAVE + threadIdx.x;
    if (e >= B) return;
    const Hdr* h = (const Hdr*)(state + (size_t)e * L.rec_bytes);
    if (flags_out) flags_out[e] = h->flags;
    if (dec_out) dec_out[e] = (int64_t)h->d;
    if (now_out) now_out[e] = h->now;
}

__global__ __launch_bounds__(WAVE) void k_get_tasks(Layout L, KP P, unsigned char* state, uint8_t* finished,
                                                   uint8_t* feasible, double* time_start, double* time_finish,
                                                   double* sum_wait, int32_t* status, int32_t* n_members,
                                                   int32_t* n_abandoned) {
    const int e = blockIdx.x, lane = threadIdx.x;
    copy16(smem, state + (size_t)e * L.rec_bytes, L.rec_bytes, lane);
    WSYNC();
    Env E = make_env(smem, L);
    Hdr h = *(Hdr*)smem;
    if (sum_wait) compute_waits(E, h, P, lane);
    for (int t = lane; t < L.T; t += WAVE) {
        const size_t o = (size_t)e * L.T + t;
        const uint32_t info = E.tinfo[t];
        if (finished) finished[o] = (info & T_FIN) ? 1 : 0;
        if (feasible) feasible[o] = (info & T_FEAS) ? 1 : 0;
        if (time_start) time_start[o] = E.ts[t];
        if (time_finish) time_finish[o] = E.tf[t];
        if (sum_wait) sum_wait[o] = E.tw[t];
        if (status) status[o] = (int)(int8_t)((info >> 8) & 0xFF);
        if (n_members) n_members[o] = (info >> 16) & 0xFF;
        if (n_abandoned) n_abandoned[o] = (int32_t)E.tnab[t];
    }
}

__global__ __launch_bounds__(WAVE) void k_get_agents(Layout L, KP P, unsigned char* state, double* sum_wait,
                                                    double* travel_dist, double* next_decision, double* arrival,
                                                    double* x, double* y, uint8_t* returned, uint8_t* assigned,
                                                    int32_t* current) {
    const int e = blockIdx.x, lane = threadIdx.x;
    copy16(smem, state + (size_t)e * L.rec_bytes, L.rec_bytes, lane);
    WSYNC();
    Env E = make_env(smem, L);
    Hdr h = *(Hdr*)smem;
    if (sum_wait) compute_waits(E, h, P, lane);
    for (int a = lane; a < L.A; a += WAVE) {
        const size_t o = (size_t)e * L.A + a;
        const uint32_t ai = E.ainfo[a];
        if (sum_wait) sum_wait[o] = E.aw[a];
        if (travel_dist) travel_dist[o] = E.tdist[a];
        if (next_decision) next_decision[o] = E.nd[a];
        if (arrival) arrival[o] = E.arr[a];
        if (x) x[o] = E.ax[a];
        if (y) y[o] = E.ay[a];
        if (returned) returned[o] = (ai & A_RETURNED) ? 1 : 0;
        if (assigned) assigned[o] = (ai & A_ASSIGNED) ? 1 : 0;
        if (current) current[o] = E.cur[a];
    }
}

__global__ void k_distance(const double* ax, const double* ay, const double* bx, const double* by, double* dist_out,
                           double* time_out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = dist2(ax[i], ay[i], bx[i], by[i]);
    if (dist_out) dist_out[i] = d;
    if (time_out) time_out[i] = d / 0.2;
}

// ---------------------------------------------------------------------------------- host side
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(DCM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

Layout make_layout(int A, int T) {
    Layout L{};
    L.A = A; L.T = T;
    uint32_t o = sizeof(Hdr);
    L.o_ax = o; o += 8 * A; L.o_ay = o; o += 8 * A; L.o_arr = o; o += 8 * A; L.o_nd = o; o += 8 * A; L.o_tdist = o; o += 8 * A;
    L.o_cur = o; o += 4 * A; L.o_ainfo = o; o += 4 * A;
    o = align_up(o, 8);
    L.o_ts = o; o += 8 * T; L.o_tf = o; o += 8 * T;
    L.o_marr = o; o += 8 * M * T;
    L.o_mids = o; o += 8 * T;
    L.o_tinfo = o; o += 4 * T; L.o_tnab = o; o += 4 * T;
    L.mut_bytes = align_up(o, 16);
    o = L.mut_bytes;
    L.o_tx = o; o += 8 * T; L.o_ty = o; o += 8 * T; L.o_tdur = o; o += 8 * T;
    L.rec_bytes = align_up(o, 16);
    o = L.rec_bytes;
    L.o_tw = o; o += 8 * T; L.o_aw = o; o += 8 * A;
    L.lds_bytes = align_up(o, 16);
    return L;
}

}  // namespace

struct dcm_env {
    dcm_params p;
    Layout L;
    KP kp;
    unsigned char* state = nullptr;  // [B][rec_bytes]
    double* summary = nullptr;       // [B][8]
    bool loaded = false, reset_done = false;
};

extern "C" {

const char* dcm_last_error(void) { return g_err; }
int dcm_abi_version(void) { return DCM_ABI_VERSION; }

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
    HIP_TRY(hipSetDevice(params->device));
    dcm_env* h = new (std::nothrow) dcm_env();
    if (!h) return fail(DCM_ERR_INVALID, "dcm_create: out of host memory");
    h->p = *params;
    h->L = make_layout(params->n_agents, params->n_tasks);
    h->kp.mwt = params->max_waiting_time;
    h->kp.max_time = params->max_time;
    if (h->L.lds_bytes > 160 * 1024) { delete h; return fail(DCM_ERR_INVALID, "dcm_create: env record does not fit the 160 KiB LDS"); }
    const size_t bytes = (size_t)params->n_envs * h->L.rec_bytes;
    hipError_t e1 = hipMalloc((void**)&h->state, bytes);
    hipError_t e2 = hipMalloc((void**)&h->summary, (size_t)params->n_envs * 8 * sizeof(double));
    if (e1 != hipSuccess || e2 != hipSuccess) {
        if (h->state) (void)hipFree(h->state);
        if (h->summary) (void)hipFree(h->summary);
        delete h;
        return fail(DCM_ERR_HIP, "dcm_create: hipMalloc failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    hipError_t e3 = hipMemset(h->state, 0, bytes);
    if (e3 != hipSuccess) { (void)hipFree(h->state); (void)hipFree(h->summary); delete h; return fail(DCM_ERR_HIP, "hipMemset: %s", hipGetErrorString(e3)); }
    // kernels that keep the record in LDS may need more than the default 64 KiB of dynamic LDS
    const void* ks[] = {(const void*)k_reset, (const void*)k_observe, (const void*)k_step, (const void*)k_rollout_random,
                        (const void*)k_get_tasks, (const void*)k_get_agents};
    for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->L.lds_bytes);
    *out = h;
    return DCM_OK;
}

int dcm_destroy(dcm_env* env) {
    if (!env) return DCM_OK;
    (void)hipSetDevice(env->p.device);
    if (env->state) (void)hipFree(env->state);
    if (env->summary) (void)hipFree(env->summary);
    delete env;
    return DCM_OK;
}

#define CHECK_ENV(env) \
    if (!(env)) return fail(DCM_ERR_INVALID, "null env handle")
#define LAUNCH_OK() HIP_TRY(hipGetLastError())

int dcm_load_instances(dcm_env* env, const double* depot, const double* task_xy, const int32_t* req, const double* dur,
                       void* stream) {
    CHECK_ENV(env);
    if (!depot || !task_xy || !req || !dur) return fail(DCM_ERR_INVALID, "dcm_load_instances: null array");
    hipLaunchKernelGGL(k_load_instances, dim3(env->p.n_envs), dim3(WAVE), 0, (hipStream_t)stream, env->L, env->state, depot,
                       task_xy, req, dur);
    LAUNCH_OK();
    env->loaded = true;
    env->reset_done = false;
    return DCM_OK;
}

int dcm_reset(dcm_env* env, const uint64_t* seeds, void* stream) {
    CHECK_ENV(env);
    if (!env->loaded) return fail(DCM_ERR_STATE, "dcm_reset: call dcm_load_instances first");
    if (!seeds) return fail(DCM_ERR_INVALID, "dcm_reset: null seeds");
    hipLaunchKernelGGL(k_reset, dim3(env->p.n_envs), dim3(WAVE), env->L.lds_bytes, (hipStream_t)stream, env->L, env->kp,
                       env->state, seeds, env->summary);
    LAUNCH_OK();
    env->reset_done = true;
    return DCM_OK;
}

int dcm_observe(dcm_env* env, float* agents_out, float* tasks_out, uint8_t* mask_out, int32_t* leader_out,
                uint8_t* active_out, const int32_t* leader_in, void* stream) {
    CHECK_ENV(env);
    if (!env->reset_done) return fail(DCM_ERR_STATE, "dcm_observe: call dcm_reset first");
    hipLaunchKernelGGL(k_observe, dim3(env->p.n_envs), dim3(WAVE), env->L.lds_bytes, (hipStream_t)stream, env->L, env->state,
                       agents_out, tasks_out, mask_out, leader_out, active_out, leader_in);
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
    hipLaunchKernelGGL(k_step, dim3(env->p.n_envs), dim3(WAVE), env->L.lds_bytes, (hipStream_t)stream, env->L, env->kp,
                       env->state, actions, leader_in, nfol_in, followers_in, agents_out, tasks_out, mask_out, leader_out,
                       active_out, env->summary);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_rollout_random(dcm_env* env, int32_t episodes, float* agents_out, float* tasks_out, uint8_t* mask_out,
                       int64_t* steps_out, void* stream) {
    CHECK_ENV(env);
    if (!env->reset_done) return fail(DCM_ERR_STATE, "dcm_rollout_random: call dcm_reset first");
    if (episodes < 1) return fail(DCM_ERR_INVALID, "dcm_rollout_random: episodes must be >= 1");
    hipLaunchKernelGGL(k_rollout_random, dim3(env->p.n_envs), dim3(WAVE), env->L.lds_bytes, (hipStream_t)stream, env->L,
                       env->kp, env->state, (int)episodes, agents_out, tasks_out, mask_out, steps_out, env->summary);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_summary(dcm_env* env, double* out, void* stream) {
    CHECK_ENV(env);
    if (!out) return fail(DCM_ERR_INVALID, "dcm_summary: null out");
    HIP_TRY(hipMemcpyAsync(out, env->summary, (size_t)env->p.n_envs * 8 * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return DCM_OK;
}

int dcm_env_status(dcm_env* env, uint32_t* flags_out, int64_t* decisions_out, double* now_out, void* stream) {
    CHECK_ENV(env);
    const int B = env->p.n_envs;
    hipLaunchKernelGGL(k_env_status, dim3((B + WAVE - 1) / WAVE), dim3(WAVE), 0, (hipStream_t)stream, env->L, env->state, B,
                       flags_out, decisions_out, now_out);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_get_tasks(dcm_env* env, uint8_t* finished, uint8_t* feasible, double* time_start, double* time_finish,
                  double* sum_wait, int32_t* status, int32_t* n_members, int32_t* n_abandoned, void* stream) {
    CHECK_ENV(env);
    hipLaunchKernelGGL(k_get_tasks, dim3(env->p.n_envs), dim3(WAVE), env->L.lds_bytes, (hipStream_t)stream, env->L, env->kp,
                       env->state, finished, feasible, time_start, time_finish, sum_wait, status, n_members, n_abandoned);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_get_agents(dcm_env* env, double* sum_wait, double* travel_dist, double* next_decision, double* arrival, double* x,
                   double* y, uint8_t* returned, uint8_t* assigned, int32_t* current, void* stream) {
    CHECK_ENV(env);
    hipLaunchKernelGGL(k_get_agents, dim3(env->p.n_envs), dim3(WAVE), env->L.lds_bytes, (hipStream_t)stream, env->L, env->kp,
                       env->state, sum_wait, travel_dist, next_decision, arrival, x, y, returned, assigned, current);
    LAUNCH_OK();
    return DCM_OK;
}

int dcm_state_bytes(dcm_env* env, size_t* bytes_out) {
    CHECK_ENV(env);
    if (!bytes_out) return fail(DCM_ERR_INVALID, "null bytes_out");
    *bytes_out = (size_t)env->p.n_envs * env->L.rec_bytes + (size_t)env->p.n_envs * 8 * sizeof(double);
    return DCM_OK;
}

int dcm_clone_state(dcm_env* env, void* dst, void* stream) {
    CHECK_ENV(env);
    if (!dst) return fail(DCM_ERR_INVALID, "dcm_clone_state: null dst");
    const size_t sb = (size_t)env->p.n_envs * env->L.rec_bytes;
    HIP_TRY(hipMemcpyAsync(dst, env->state, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync((unsigned char*)dst + sb, env->summary, (size_t)env->p.n_envs * 8 * sizeof(double),
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DCM_OK;
}

int dcm_restore_state(dcm_env* env, const void* src, void* stream) {
    CHECK_ENV(env);
    if (!src) return fail(DCM_ERR_INVALID, "dcm_restore_state: null src");
    const size_t sb = (size_t)env->p.n_envs * env->L.rec_bytes;
    HIP_TRY(hipMemcpyAsync(env->state, src, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(env->summary, (const unsigned char*)src + sb, (size_t)env->p.n_envs * 8 * sizeof(double),
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    env->loaded = true;
    env->reset_done = true;
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

int dcm_record_bytes(dcm_env* env, size_t* bytes_out) {
    CHECK_ENV(env);
    if (!bytes_out) return fail(DCM_ERR_INVALID, "null bytes_out");
    *bytes_out = env->L.rec_bytes;
    return DCM_OK;
}

}  // extern "C"
