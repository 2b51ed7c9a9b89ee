// rollout_fast.hpp -- the persistent rollout kernel of the ONE-CHUNK layouts (A <= 64, T <= 63: one lane per agent and per task):
// lane-owned state lives in REGISTERS for the whole fast path, LDS is only the cross-lane exchange.
// Included by dcmrta_env.hip inside its anonymous namespace, after Sim<>.
//
// Why: k_rollout_random keeps the record in LDS and re-reads every field in every phase of a decision -- ~60 LDS instructions in
// ~12 DEPENDENT round trips, each phase fenced from the next, and (round-3 profile) 0.39 of the wave time parked on s_waitcnt,
// 347 scalar + 89 branch instructions per decision, most of them the compiler's exec-mask bookkeeping around guarded lane loops
// and error returns.  With four waves per SIMD (4096 envs on 1024 SIMDs) the kernel is bound by ONE wave's critical path, so the
// cure is fewer dependent steps, not more parallelism:
//   * lane a owns agent a (x, y, arrival, next_decision, travel_dist, route[-1], flags), lane t owns task t (status word,
//     time_start, time_finish, ordered member ids, and the read-only x, y, duration); lane 63 owns the depot as a pseudo-task
//     (its observation row 0 and mask byte 0 leave in the same store instructions as the task rows);
//   * wave-uniform reads of a chosen lane's state (the leader's position, the chosen task's status word / ids / position) are
//     v_readlane -- no LDS round trip; per-lane reads of own state cost nothing;
//   * LDS carries only what another lane must read by INDEX: the member arrival slots f64[M][T] (written by the joining agents'
//     lanes, read by the task's lane), and the status words and the two time arrays (written through by the task lanes, gathered
//     by the agents' lanes in agent_update);
//   * member removal needs no scatter from a task's lane to its members: the task's lane compacts its own slots and publishes a
//     `gone` agent bitmask, and a wave-uniform pass over the dropping tasks lets each agent's lane take its own abandonment
//     (v_readlane of the mask) -- the agents' counters and flags never leave their registers;
//   * the lane code is select-based (no per-lane branches except the member-removal compaction), error returns do not exist on
//     this path (the device policy only takes valid actions on a validated instance), the observation pointers are a template
//     parameter, and every loop / branch condition is wave-uniform in SGPRs.
// Everything that happens once per episode or less -- reset, the terminal metrics, events at which nobody can decide, the first
// event of an episode -- runs the verified Sim<> code on the LDS image: the fast path flushes its registers, calls it, reloads.
//
// Round 6 (profiles/r06_budget.md): the kernel is bound by the length of ONE wave's instruction stream (a lone wave needs 0.78 ms of
// the 1.24 ms a 4096-env launch takes), so the stream itself was cut: the choice-protocol keys of 64 decisions are computed one per
// lane; decide() has no exit in its middle (an exit there keeps a second copy of all 15 lane-owned fields alive: 15 v_mov per
// decision); task_update touches the latest arrival / time_start only when a coalition is complete / a task becomes feasible.
//
// Reference restated: worker.py:45-87 (loop), env/task_env.py:161-342 (the functions named at each block below).
#pragma once
#include <type_traits>

// TRK (lockstep kernel): the fast path records which task sections of the record it has changed (Sim::DIRTY_* bits), so that
// the write-back can skip the others
template <int CA, int CT, bool RS, bool OBS, bool TRK = false>
struct Fast {
    using SimT = Sim<CA, CT, RS, false>;
    using AMask = typename SimT::AMask;
    static_assert(CA >= 1 && CA <= 64 && CT >= 1 && CT <= 64, "one lane per agent / task; lane 63 is the depot's (the host dispatches T <= 63 only)");
    static constexpr int DL = 63;                     // depot lane
    static constexpr Lay L{CA, CT};

    SimT S;
    double* dummy;                                     // 64 doubles of LDS nobody reads (the removal path's discarded writes)
    mutable uint32_t dirty = 0;                        // TRK only
    mutable uint64_t achg = ~0ull;                     // TRK only: the agents whose fields the step has changed (set by flush)
    mutable uint64_t dt_join = 0, dt_times = 0;        // TRK only: the tasks whose ids / arrival rows (a join) and time_start / time_finish
                                                       // (became feasible) the fast path has changed: the write-back sends their 64-byte
                                                       // pieces of those sections instead of the sections
    mutable bool calm = false;                         // the last task_update call left every task at a fixed point for its `now`
    FPH_MEMBERS;
    uint64_t am, tm;                                   // lanes that own an agent / a task (wave-uniform masks)
    // per-lane constants
    bool inA, inT, isD;
    int la, lt;                                        // clamped own agent / task index (0 for the lanes that own none)

    struct R {
        double ax, ay, arr, nd, td;                    // agent: location, arrival_time[-1], next_decision, travel_dist
        int32_t cur; uint32_t ai;                      //        route[-1], ainfo word
        double cts, cdur;                              //        time_start / duration of route[-1] as of the last agent_update
        uint32_t ti; double ts, tf; uint64_t ids;      // task:  tinfo word, time_start, time_finish, ordered member ids
        uint64_t lm; uint32_t nab;                     //        the listed members as an agent bitmask; len(abandoned_agent)
        double tx, ty, dur;                            //        instance (depot lane: depot x, y, 0)
        uint16_t* ab;                                  // agent: its row of the abandonment log (HBM side table)
    };

    __device__ __forceinline__ void init(int lane) {
        const int A_ = S.A(), T_ = S.T();
        am = A_ >= 64 ? ~0ull : ((1ull << A_) - 1ull);
        tm = (1ull << T_) - 1ull;
        inA = lane < A_; inT = lane < T_; isD = lane == DL;
        la = inA ? lane : 0; lt = inT ? lane : 0;
    }
    // position of the idx-th (0-based) set bit of m, idx < popcount(m): the lanes up to and including it are exactly those with
    // fewer than idx + 1 set bits below them
    __device__ __forceinline__ static int nth(uint64_t m, int idx) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        return __popcll(__ballot(rank <= idx)) - 1;
    }
    __device__ __forceinline__ static double rl(double v, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    }
    __device__ __forceinline__ static uint64_t rl(uint64_t v, int src) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    }

    // ------------------------------------------------------------------------------ registers <-> LDS image
    __device__ __forceinline__ void load_consts(R& r) const {
        r.tx = isD ? ((const Hdr*)S.base)->depot_x : S.tx()[lt];
        r.ty = isD ? ((const Hdr*)S.base)->depot_y : S.ty()[lt];
        r.dur = isD ? 0.0 : S.tdur()[lt];
        r.ab = S.ablog() + la * AB_CAP;
    }
    __device__ __forceinline__ void reload(R& r) const {
        calm = false;                                  // nothing is known about the general code's last call
        r.ax = S.ax()[la]; r.ay = S.ay()[la]; r.arr = S.arr()[la]; r.nd = S.nd()[la]; r.td = S.tdist()[la];
        r.cur = S.cur()[la]; r.ai = S.ainfo()[la];
        const int K = r.cur < 0 ? 0 : r.cur;
        r.cts = S.ts()[K]; r.cdur = S.tdur()[K];
        r.ti = S.tinfo()[lt]; r.ts = S.ts()[lt]; r.tf = S.tf()[lt]; r.ids = S.mids()[lt]; r.nab = S.tnab()[lt];
        r.lm = 0ull;
        const int n = (r.ti >> 16) & 0xFF;
#pragma unroll
        for (int j = 0; j < M; j++) r.lm |= (j < n) ? (1ull << ((r.ids >> (8 * j)) & 0xFF)) : 0ull;
    }
    // what the fast path keeps in registers only (everything else is written through when it changes)
    __device__ __forceinline__ void flush(const R& r) const {
        if constexpr (TRK) { if (gridDim.x >= 8192u) {
            // which agents differ from the image the record was loaded into (bit patterns: next_decision may be NaN)?  The
            // write-back sends only their 16-byte pieces of the agent arrays -- for grids that fill the machine several times over,
            // where HBM traffic is what the launch costs (a single round of workgroups is latency-bound: the compare would only add
            // to the chain, 15.4 vs 15.2 us at 4096 envs).
            auto ne = [](double a, double b) { return __double_as_longlong(a) != __double_as_longlong(b); };
            const int ch_ = (int)ne(r.ax, S.ax()[la]) | (int)ne(r.ay, S.ay()[la]) | (int)ne(r.arr, S.arr()[la]) | (int)ne(r.nd, S.nd()[la]) |
                            (int)ne(r.td, S.tdist()[la]) | (int)(r.cur != S.cur()[la]) | (int)(r.ai != S.ainfo()[la]);
            const bool ch = inA && ch_ != 0;
            achg = __ballot(ch);
        } }
        if (inA) {
            S.ax()[la] = r.ax; S.ay()[la] = r.ay; S.arr()[la] = r.arr; S.nd()[la] = r.nd; S.tdist()[la] = r.td;
            S.cur()[la] = r.cur; S.ainfo()[la] = r.ai;
        }
        if (inT) { S.mids()[lt] = r.ids; S.tnab()[lt] = r.nab; }
        WSYNC();
    }

    // ------------------------------------------------------------------------------ task_update, env/task_env.py:245-281
    // One lane per task, inputs from the lane's registers + the member arrival slots in LDS.  Returns with tinfo / time_start /
    // time_finish written through for agent_update's gather.
    // (site: which call this is -- path counters of the developer builds only)
    __device__ __forceinline__ void task_update(R& r, double now, double mwt, int lane, int site = 0) const {
        uint32_t info = r.ti;
        const bool feas0 = info & T_FEAS;
        const int req = info & 0xFF, n = (info >> 16) & 0xFF;                    // :250
        double av[M];
#pragma unroll
        for (int j = 0; j < M; j++) av[j] = S.marr()[j * CT + lt];               // :251 (unused slots hold NaN)
        const double tfin = r.tf, dur = r.dur;
        const int status = req - n;                                              // :252
        // The latest arrival only matters for a task whose coalition is complete (status <= 0 :254): the spread test, time_start and
        // the spread rule's threshold (:255-265).  Such a task exists at the one call after the completing join (or while a stale
        // status lingers, Q3); every other call -- wave-uniform test -- skips the four v_max_f64 and the tests that need them.
        double mn = av[0];
#pragma unroll
        for (int j = 1; j < M; j++) mn = nanmin2(mn, av[j]);
        const bool le0 = status <= 0;                                            // :254
        double mx = 0.0, thr = 0.0;
        bool ok = false;
        if (__ballot(inT && !feas0 && le0)) {
            mx = av[0];
#pragma unroll
            for (int j = 1; j < M; j++) mx = nanmax2(mx, av[j]);
            ok = le0 && (mx - mn <= mwt);                                        // :255
            thr = mx - mwt;                                                      // :262
        }
        // does any member leave?  (see Sim::task_update: decided on the earliest arrival alone)
        const bool any_drop = inT && !feas0 && (le0 ? (!ok && mn <= thr) : (now - mn >= mwt));
        const bool becomes = !feas0 && ok;                                       // :256-258
        // time_start / time_finish change -- and are written through for agent_update's gather -- only when a task becomes feasible
        // (:256-257): wave-uniform test
        const uint64_t bec = __ballot(becomes && inT);
        double nts = r.ts, ntf = tfin;
        if (bec) { nts = becomes ? mx : nts; ntf = becomes ? mx + dur : ntf; }
        int nn = n;
        const uint64_t dmask = __ballot(any_drop);
        if (dmask) {
            CNT(6 + site);
            FPM(20);
            // Members leave (:262-265 spread branch, :268-271 waiting branch with its remove-while-iterating skip, Q1).  The
            // task's lane compacts its own slots; the agents' lanes then take their abandonment from the task's `gone` mask --
            // no scatter through LDS: an agent's counters live in its own lane.
            uint64_t gone = 0ull;
            // (which of the two removal rules is in play is wave-uniform almost always -- the spread rule fires ~3 times per
            //  episode -- so the per-slot tests of the other one are skipped by a scalar branch)
            const bool any_spread = __ballot(any_drop && le0) != 0ull, any_wait = __ballot(any_drop && !le0) != 0ull;
            uint32_t spread = 0, q1 = 0;
            if (any_spread) {
#pragma unroll
                for (int j = 0; j < M; j++) spread |= (av[j] <= thr) ? (1u << j) : 0u;       // :262-265
            }
            if (any_wait) {
                bool prev = false;
#pragma unroll
                for (int j = 0; j < M; j++) {
                    const bool e = !prev && (now - av[j] >= mwt);                // :269, skipping the element after a removal (Q1)
                    q1 |= e ? (1u << j) : 0u;
                    prev = e;
                }
            }
            if (any_drop) {
                const uint32_t drop = le0 ? spread : q1;                         // only listed slots can be set: unused ones hold NaN
                const uint32_t keep = ((1u << n) - 1u) & ~drop;
                // Compact the survivors in order, without a branch: vacate all slots, then every surviving member moves down to
                // its rank among the survivors (a target never lies above its source, so earlier writes are never clobbered);
                // the leavers' writes go to a per-lane dummy slot.  At least one listed member leaves, so at most four survive:
                // their ids fit the low word.  (Writing each slot straight to its final place -- survivors to their rank, leavers
                // as NaN behind them: five writes instead of ten, no dummy slot -- was tried in round 6: +20 VALU for the target
                // and value selects, not faster.)
                const uint32_t idl = (uint32_t)r.ids, idh = (uint32_t)(r.ids >> 32);
                uint32_t nids = 0;
                using gone_t = typename std::conditional<(CA <= 32), uint32_t, uint64_t>::type;   // agent ids below 32: one word
                gone_t g = 0;
                double* const row0 = S.marr() + lt;
                double* const dump = dummy + lane;
#pragma unroll
                for (int j = 0; j < M; j++) row0[j * CT] = __builtin_nan("");
#pragma unroll
                for (int j = 0; j < M; j++) {
                    const bool kp = (keep >> j) & 1u, lv = (drop >> j) & 1u;
                    const int kj = __popc(keep & ((1u << j) - 1u));
                    const uint32_t id = (j < 4 ? (idl >> (8 * j)) : idh) & 0xFFu;
                    nids |= kp ? (id << (8 * kj)) : 0u;
                    g |= lv ? ((gone_t)1 << id) : (gone_t)0;
                    *(kp ? row0 + kj * CT : dump) = av[j];
                }
                gone = (uint64_t)g;
                r.ids = (uint64_t)nids; r.lm &= ~gone;
                r.nab += (uint32_t)__popc(drop);                                 // abandoned_agent.append :265/:271
                nn = __popc(keep);
            }
            if constexpr (TRK) dirty |= SimT::DIRTY_ALL & ~SimT::DIRTY_TIMES;   // slots compacted: every arrival row, ids, counts
            uint64_t todo = dmask;
            do {
                CNT(7);
                const int t = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1ull;
                const uint64_t g = rl(gone, t);
                if ((g >> lane) & 1ull) {                                        // this lane's agent was dropped by task t
                    const uint32_t nth_ = r.ai >> 16;
                    r.ai += 1u << 16;
                    if (nth_ < (uint32_t)AB_CAP) r.ab[nth_] = (uint16_t)t;
                    else { const uint32_t ci = (uint32_t)(la * S.T() + t); atomicAdd((uint32_t*)S.abcnt() + (ci >> 1), 1u << (16 * (ci & 1))); }
                    if (r.cur == t) r.ai &= ~A_MEMBER;                           // no longer `agent in current_task['members']` :230
                }
            } while (todo);
            FPM(21);
        }
        const uint32_t info_i = ((info | (ok ? T_FEAS : 0u)) & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)nn << 16);
        const uint32_t info_f = info | ((now >= tfin) ? T_FIN : 0u);             // :273-274
        info = feas0 ? info_f : info_i;
        r.ti = info;
        if (inT) S.tinfo()[lt] = info;
        bool over_already = false;
        if (bec) {
            r.ts = nts; r.tf = ntf;
            if (inT) { S.ts()[lt] = nts; S.tf()[lt] = ntf; }
            if constexpr (TRK) { dirty |= SimT::DIRTY_TIMES; dt_times |= bec; }
            over_already = __ballot(becomes && inT && now >= ntf) != 0ull;
        }
        const bool all_feasible = (__ballot(!(info & T_FEAS)) & tm) == 0ull;
        // (see Sim::task_update: a call can only change a task again at the same `now` if this one removed members or made a task
        //  feasible that is already over)
        calm = dmask == 0ull && !over_already;
        WSYNC();
        if (all_feasible) {                                                      // depot :277-280
            CNT(16);
            if ((r.ai & A_INDEPOT) && now >= r.arr) r.ai |= A_RETURNED;
        }
    }

    // ------------------------------------------------------------------------------ agent_update, env/task_env.py:207-243
    __device__ __forceinline__ void agent_update(R& r, double now, double mwt) const {
        const int c = r.cur;
        const int K = c < 0 ? 0 : c;
        const uint32_t gi = S.tinfo()[K];                                        // :228
        const double gts = S.ts()[K], gtf = S.tf()[K], gdur = S.tdur()[K];
        const bool member = (gi & T_FEAS) && (r.ai & A_MEMBER);                  // :229-230
        const double ndv = (c == -1) ? __builtin_nan("") : (member ? gtf : r.arr + mwt);   // :226,:231,:235,:238
        const uint32_t as = member ? ((r.ai & A_ASSIGNED) | ((now >= gts) ? A_ASSIGNED : 0u)) : 0u;   // :232-240
        r.nd = (c != -2) ? ndv : r.nd;                                           // :209
        r.ai = (c >= 0) ? ((r.ai & ~A_ASSIGNED) | as) : r.ai;                    // depot leaves `assigned` untouched (Q6)
        r.cts = gts; r.cdur = gdur;
    }

    // ------------------------------------------------------------------------------ observation, worker.py:57-68
    // env/task_env.py:165-200 relative to the leader; rows straight into the policy's input tensors.  Returns the ballot of the
    // unmasked tasks (the random policy picks from it).
    __device__ __forceinline__ uint64_t observe(const R& r, double now, int leader, float* __restrict__ agrow, float* __restrict__ tkrow,
                                                uint8_t* __restrict__ mkp) const {
        const double lx = rl(r.ax, leader), ly = rl(r.ay, leader);
        if constexpr (OBS) {
            const bool on = r.cur >= 0;                                          // :168
            const double x = r.arr - now, w = now - r.arr, rem = r.cts + r.cdur - now;
            const double travel = (on && x > 0.) ? x : 0.;                       // :169
            const double waiting = (on && now <= r.cts && w > 0.) ? w : 0.;      // :170
            const double remaining = (on && now >= r.cts && rem > 0.) ? rem : 0.;   // :171
            const float f0 = (float)travel, f1 = (float)remaining, f2 = (float)waiting;
            const float f3 = (float)(lx - r.ax), f4 = (float)(ly - r.ay), f5 = (r.ai & A_ASSIGNED) ? 1.f : 0.f;
            if (inA) { agrow[0] = f0; agrow[1] = f1; agrow[2] = f2; agrow[3] = f3; agrow[4] = f4; agrow[5] = f5; }   // :176-177
        }
        const uint32_t info = isD ? 0u : r.ti;
        const int status = (int)(int8_t)((info >> 8) & 0xFF);
        const bool unfinished = !(info & T_FEAS) && status > 0;                  // :199
        const uint64_t bm = __ballot(unfinished) & tm;
        if constexpr (OBS) {
            // :193 per task; the depot's byte is False iff every task is masked (worker.py:58-61)
            const bool zero = isD ? (bm == 0ull) : unfinished;
            const uint8_t mv = zero ? 0 : 1;
            const float g0 = (float)status, g1 = (float)(info & 0xFF), g2 = (float)r.dur;
            const float g3 = (float)(r.tx - lx), g4 = (float)(r.ty - ly);        // :185-188
            if (inT || isD) { *mkp = mv; tkrow[0] = g0; tkrow[1] = g1; tkrow[2] = g2; tkrow[3] = g3; tkrow[4] = g4; }
        }
        return bm;
    }

    // ------------------------------------------------------------------------------ one decision
    // worker.py:54-76: leader, observation, uniform-random valid action, TaskEnv.step + agent_step (env/task_env.py:300-342),
    // task_update, agent_update.  Returns the number of agents of the group that have not acted yet.
    // worker.py:54 -- the deciding agent of the current group (protocol slot 0); -1: the group is empty (unreachable)
    __device__ __forceinline__ int pick_leader(const R& r, const HdrRegs& h, uint64_t k1, uint64_t& gm) const {
        gm = __ballot((int)((r.ai >> 8) & 0xFFu) == h.cur_group) & am;
        const int glen = __popcll(gm);
        if (glen == 0) return -1;
        return nth(gm, below((uint32_t)(k1 >> 32), glen));
    }
    // k2p: the decision's second key mix64(k1 + GAMMA) (the first two follower draws) if the caller has it, else nullptr
    // nvp: receives the number of unmasked tasks the decision saw (the launch's wave-priority estimate of the work left)
    __device__ __forceinline__ int decide(R& r, HdrRegs& h, const KP& P, int lane, uint64_t k1, float* agrow, float* tkrow, uint8_t* mkp,
                                          const uint64_t* k2p = nullptr, int* nvp = nullptr) const {
        uint64_t gm;
        // (no early return: an exit from the middle of a decision would keep the whole register set of the agent / task state alive
        //  in a second copy -- 15 v_mov per decision at the loop latch.  An empty group is unreachable; if it ever happened the env
        //  is flagged and stops at the loop's ordinary exit test, its state no longer meaningful.)
        int leader = pick_leader(r, h, k1, gm);
        if (leader < 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; leader = 0; gm = 1ull; }
        FPH(0);
        const uint64_t bm = observe(r, h.now, leader, agrow, tkrow, mkp);
        FPH(1);
        // uniform-random valid action (protocol slot 1)
        const int nv = __popcll(bm);
        if (nvp) *nvp = nv;
        const int action = nv ? nth(bm, below((uint32_t)k1, nv)) + 1 : 0;
        FPH(2);
        return apply(r, h, P, lane, k1, gm, leader, action, k2p);
    }
    // TaskEnv.step :326-342 with the leader's (valid: unmasked task or depot) action, then task_update / agent_update
    __device__ __forceinline__ int apply(R& r, const HdrRegs& h, const KP& P, int lane, uint64_t k1, uint64_t gm, int leader, int action,
                                         const uint64_t* k2p = nullptr) const {
        const double now = h.now;
        uint64_t rest = gm & ~(1ull << leader);                                  // :328
        int rlen = __popcll(gm) - 1;
        uint64_t mm = 1ull << leader, mlist = (uint64_t)(uint32_t)leader;
        int nm = 1;
        int mypos = 0;                                                           // this lane's position in the step's member list
        const int tl = action ? action - 1 : DL;                                 // lane that owns the target
        if (action == 0) {                                                       // vacancy = len(group) :327 (Q9)
            CNT(2);
            mm |= rest; nm += rlen; rlen = 0;
        } else if (rlen != 0) {
            const int vacancy = (int)(int8_t)(((uint32_t)__builtin_amdgcn_readlane((int)r.ti, tl) >> 8) & 0xFF);   // :327 (may be stale)
            const int nf = (vacancy > 1) ? ((vacancy - 1 < rlen) ? vacancy - 1 : rlen) : 0;   // :330-331
            uint64_t kk = k1;
            for (int j = 0; j < nf; j++) {                                       // :331 choice without replacement
                CNT(1);
                FPM(22);
                if ((j & 1) == 0) kk = (j == 0 && k2p) ? *k2p : mix64(kk + GAMMA);
                const uint32_t rr = (j & 1) ? (uint32_t)kk : (uint32_t)(kk >> 32);
                const int f = nth(rest, below(rr, rlen));
                rest &= ~(1ull << f); rlen--;                                    // :332-333
                mm |= 1ull << f;
                mlist |= (uint64_t)(uint32_t)f << (8 * nm);
                // lane f takes its list position
                mypos = lane == f ? nm : mypos;
                nm++;
                FPM(23);
            }
        }
        FPH(3);
        const double tx_ = rl(r.tx, tl), ty_ = rl(r.ty, tl);
        // agent_step :300-324 on ALL lanes (fp64 VALU work with fewer than 16 active lanes is 4x slower on gfx950; the asm
        // statement keeps the compiler from sinking the chain into the members-only block below)
        double d = dist2(r.ax, r.ay, tx_, ty_);
        double arrv = now + over_velocity(d);                                    // :315,:318
        asm volatile("" : "+v"(d), "+v"(arrv));
        FPH(4);
        const bool mem = (mm >> lane) & 1ull;
        uint64_t ids = 0ull; uint32_t kinfo = 0; int n = 0;
        int slot = 0;
        if (action) {
            // :321-322 members.append unless already listed (Q4: a re-joining agent keeps its slot, its arrival is overwritten)
            kinfo = (uint32_t)__builtin_amdgcn_readlane((int)r.ti, tl);
            ids = rl(r.ids, tl);
            n = (kinfo >> 16) & 0xFF;
            slot = n + mypos;
            if constexpr (TRK) {
                dirty |= SimT::DIRTY_IDS | ((rl(r.lm, tl) & mm) ? SimT::DIRTY_ROWS : ((((1u << nm) - 1u) << (n + 1)) & SimT::DIRTY_ROWS));   // ids + the arrival rows written
                dt_join |= 1ull << tl;
            }
            if (rl(r.lm, tl) & mm) {
                CNT(3);
                // rare (Q4): walk the members in order as the reference does; every member's lane learns its own slot
                for (int j = 0; j < nm; j++) {
                    const int m = (int)((mlist >> (8 * j)) & 0xFF);
                    const uint64_t x = ids ^ (0x0101010101010101ull * (uint64_t)(uint32_t)m);
                    uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
                    z &= (n >= 8) ? ~0ull : ((1ull << (8 * n)) - 1ull);
                    int pos;
                    if (z) pos = (__ffsll((unsigned long long)z) - 1) >> 3;
                    else { pos = n++; ids |= (uint64_t)(uint32_t)m << (8 * pos); }
                    if (lane == m) slot = pos;
                }
            } else {
                ids |= mlist << (8 * n);                                         // bytes above n are always zero
                n += nm;
            }
        }
        if (mem) {
            r.td += d;                                                           // :317
            r.arr = arrv;
            r.ax = tx_; r.ay = ty_;                                              // :320
            r.cur = action - 1;                                                  // :314
            r.ai = (r.ai & ~(A_GRP | A_MEMBER)) | (action == 0 ? A_INDEPOT : A_MEMBER);
            if (action) S.marr()[slot * CT + tl] = arrv;
        }
        // A QUIET join: the task still lacks members after it (status = requirement - len(members) > 0) and the previous
        // task_update call -- at this same `now` -- left every task at a fixed point.  task_update (:245-281) then changes
        // nothing but this task's status: not enough members -> not feasible (:254); the joining members have not waited
        // (arrival >= now, :269); everybody else was evaluated at this `now` by the previous call.  The pass over the tasks is
        // skipped and the task's lane takes the new status itself.
        const int status_k = (int)(kinfo & 0xFFu) - n;
        const bool quiet = action && calm && status_k > 0;
        if (action && lane == tl) {
            r.ids = ids; r.lm |= mm;
            r.ti = quiet ? ((kinfo & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status_k & 0xFF) << 8) | ((uint32_t)n << 16))
                         : ((kinfo & ~0x00FF0000u) | ((uint32_t)n << 16));
            if (quiet) S.tinfo()[lt] = r.ti;                                     // (written through like task_update does)
        }
        WSYNC();
        FPH(5);
        if (!quiet) { CNT(5); task_update(r, now, P.mwt, lane); } else CNT(4);         // worker.py:74
        FPH(6);
        agent_update(r, now, P.mwt);                                                 // worker.py:76
        FPH(7);
        return rlen;
    }

    // ------------------------------------------------------------------------------ next event (common case)
    // Boxes D + A of the loop when somebody can decide and MAX_TIME has not passed: next_decision (env/task_env.py:283-289),
    // get_unique_group (:291-298), task_update, agent_update (worker.py:49-51).  Returns false -- with nothing changed -- when the
    // event needs the general code (nobody can decide: check_finished :366-373; or the loop test of worker.py:45 ends the episode).
    __device__ __forceinline__ bool next_event(R& r, HdrRegs& h, const KP& P, int lane) const {
        FPH(8);
        if (h.now >= P.max_time) return false;
        const double ndv = inA ? r.nd : __builtin_nan("");
        const double tmin = wave_nanmin_n<CA>(ndv);                              // :287
        if (!(tmin == tmin)) return false;
        CNT(10);
        h.now = tmin;                                                            // worker.py:49
        const bool dec = (ndv == tmin);                                          // :288 exact ==
        const uint64_t dm = __ballot(dec);
        const int first = __ffsll((unsigned long long)dm) - 1;
        bool same = true;
        if (dm & (dm - 1ull)) {                                                  // more than one decider: all on one point?
            CNT(12);
            const double x0 = rl(r.ax, first), y0 = rl(r.ay, first);
            same = __ballot(dec && !(r.ax == x0 && r.ay == y0)) == 0ull;
        }
        if (same) {
            r.ai = (r.ai & ~A_GRP) | (dec ? (1u << 8) : 0u);
            h.n_groups = 1;
        } else {
            CNT(13);
            // groups in ascending (x, then y) order == rows of np.unique(axis=0) :293
            bool todo = dec;
            uint32_t gid = 0;
            int g = 0;
            for (;;) {
                CNT(14);
                const double mxv = wave_nanmin_n<CA>(todo ? r.ax : __builtin_nan(""));
                if (!(mxv == mxv)) break;
                const double myv = wave_nanmin_n<CA>((todo && r.ax == mxv) ? r.ay : __builtin_nan(""));
                g++;
                if (todo && r.ax == mxv && r.ay == myv) { gid = (uint32_t)g; todo = false; }
            }
            r.ai = (r.ai & ~A_GRP) | (gid << 8);
            h.n_groups = g;
        }
        FPH(9);
        task_update(r, tmin, P.mwt, lane, 11);                                       // worker.py:50
        FPH(10);
        agent_update(r, tmin, P.mwt);                                                // worker.py:51
        FPH(11);
        h.empty_passes = 0;
        h.cur_group = 1;
        return true;
    }
};

// Same contract as k_rollout_random (see there); OBS: all three observation buffers given / none of them.
// PRIO: wave priorities, longest remaining work first (see the main loop) -- the host picks it for launches that fill the machine alone
template <int CA, int CT, bool RS, bool OBS, bool PRIO = false>
__global__ __launch_bounds__(WAVE, 4) void k_rollout_fast(int A, int T, int PA, int PT, KP P, unsigned char* state, int episodes,
                                                      float* agents_out, float* tasks_out, uint8_t* mask_out,
                                                      int64_t* steps_out, double* summary, uint16_t* ablog,
                                                      const int32_t* sizes, int64_t budget_all, const int64_t* budget_in,
                                                      unsigned char* gscr, double* retlog, int retcap) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    using F = Fast<CA, CT, RS, OBS>;
    using SimT = typename F::SimT;
    SimT S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = SimT::SCR_IN_LDS ? smem + L.lds_rec() : gscr + (size_t)e * L.scratch_bytes();
    const int BA = S.BA(A), BT = S.BT(T);
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    typename SimT::XY xy;
    S.template load_record<true, false>(rec, lane, xy);
    S.set_ablog(ablog, e, BA, BT, lane);
    S.set_retlog(retlog, retcap, e, lane);
    WSYNC();
    HdrRegs h = load_hdr(smem);
    // (the launch asks for 512 bytes of LDS behind everything the general code uses: the dummy slots)
    F f{S, (double*)(smem + (SimT::SCR_IN_LDS ? L.lds_bytes() : SimT::lds_image_bytes(L)))};
    f.init(lane);
    float* agrow = nullptr; float* tkrow = nullptr; uint8_t* mkp = nullptr;
    if constexpr (OBS) {
        float* ag = agents_out + (size_t)e * 6 * BA;
        float* tk = tasks_out + (size_t)e * 5 * (BT + 1);
        uint8_t* mk = mask_out + (size_t)e * (BT + 1);
        if constexpr (RS) S.write_pad_obs(lane, BA, BT, ag, tk, mk);
        agrow = ag + 6 * f.la;
        tkrow = tk + (f.inT ? 5 * (lane + 1) : 0);
        mkp = mk + (f.inT ? lane + 1 : 0);
    }
    double* row = summary + (size_t)e * 8;
    constexpr int NO_BUDGET = 0x7FFFFFFF;
    int64_t bud = budget_in ? budget_in[e] : budget_all;
    const int left0 = uni((int)((bud < 0 || bud >= NO_BUDGET) ? NO_BUDGET : bud));
    int left = left0;
    uint64_t gd = h.seed + GAMMA * (h.d + 1);
    // the choice-protocol keys of the next 64 decisions, one per lane (25 VALU instructions per 64 decisions instead of a dependent
    // chain of 20 scalar ones at the head of every decision); ki = the lane that holds the current decision's key
    uint64_t kv = mix64(gd + GAMMA * (uint64_t)lane);
    uint64_t kv2 = mix64(kv + GAMMA);                  // ... and their second keys (follower draws 0 and 1)
    int ki = 0;
    const uint64_t d0 = h.d;
    typename F::R r;
    f.load_consts(r);
    FPH_START(f);
    constexpr uint32_t ERR = DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER | DCM_FLAG_BAD_INSTANCE;
    PH_DECL;
    int ep = 0;
    bool need_adv = false;       // the general event code has to run on the (flushed) LDS image before the next decision
    // Wave priority = longest remaining work first (s_setprio).  A launch that fills the machine by itself ends with its slowest env
    // (430 of a mean 360 decisions at 20A/50T x 3 episodes) while the SIMDs it shares with envs that finished early idle; the waves
    // with the most tasks still to serve -- episodes to come x T + the unmasked tasks of the last decision, in sixths of the launch's
    // total: 4/6, 2/6, 1/6 -- win the instruction arbiter, so the four waves of a SIMD finish together: one 4096-env launch 1.237 ->
    // 1.091 ms (3 episodes), 0.447 -> 0.426 (1 episode), 8192 envs 2.23 -> 2.05.  Re-evaluated at episode ends and at the key refill
    // (every 64 decisions): nothing per decision.  Not for sub-batches that share the SIMDs with other launches (grid < 4096: several
    // streams, whose tails already overlap the others' bodies; priorities across launches measured -1 ... -5 % there, also with a
    // common deadline clock), nor in the multi-chunk kernels (their launches run in several rounds of workgroups: +0.5 / -3 %).
    // A template parameter, not a run-time test: the bookkeeping alone (one more live scalar, the refill path) cost the
    // unprioritised four-stream line 0.45 %.
    constexpr bool use_prio = PRIO;
    int nv_last = S.T(), prio_lv = 3;
    auto set_prio = [&](int ep_now) {
        const int rem6 = 6 * ((episodes - ep_now - 1) * S.T() + nv_last), tot = episodes * S.T();
        const int lv = rem6 >= 4 * tot ? 3 : (rem6 >= 2 * tot ? 2 : (rem6 >= tot ? 1 : 0));
        if (lv != prio_lv) {
            prio_lv = lv;
            if (lv == 3) __builtin_amdgcn_s_setprio(3); else if (lv == 2) __builtin_amdgcn_s_setprio(2);
            else if (lv == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
    };
    if constexpr (use_prio) __builtin_amdgcn_s_setprio(3);
    for (;;) {
        if (!need_adv) {         // head of an episode slot (the `for ep` of k_rollout_random)
            if (ep >= episodes) break;
            if (h.flags & DCM_FLAG_DONE) {   // restart from the loaded instance; d keeps running
                if (h.flags & ERR) break;
                if (left == 0) break;        // budget spent at an episode boundary: the finished episode's results stay readable
                S.reset_state(h, lane);
                need_adv = true;
            }
        }
        if (need_adv) {
            CNT(15);
            S.advance(h, P, lane, row PH_PASS);
            need_adv = false;
            // wave-uniform by construction; tell the compiler so (scalar branches in the fast loop)
            h.now = uni(h.now); h.flags = uni(h.flags); h.cur_group = uni(h.cur_group); h.n_groups = uni(h.n_groups);
            h.empty_passes = uni(h.empty_passes);
        }
        if (!(h.flags & DCM_FLAG_DONE) && left != 0) {
            WSYNC();
            f.reload(r);
            FPHK(f, 13);
            for (;;) {
                FPHK(f, 12);
                CNT(0);
                const uint64_t k1 = F::rl(kv, ki), k2 = F::rl(kv2, ki);
                const int rlen = f.decide(r, h, P, lane, k1, agrow, tkrow, mkp, &k2, use_prio ? &nv_last : nullptr);
                if (h.flags & DCM_FLAG_DONE) break;
                gd += GAMMA;
                if (++ki == WAVE) { kv = mix64(gd + GAMMA * (uint64_t)lane); kv2 = mix64(kv + GAMMA); ki = 0; if constexpr (use_prio) set_prio(ep); }
                left--;
                if (rlen == 0) {                                                  // worker.py:53 else same group, next leader
                    CNT(8);
                    if (h.cur_group < h.n_groups) { CNT(9); h.cur_group++; }                  // worker.py:52 next group
                    else if (!f.next_event(r, h, P, lane)) { need_adv = true; break; }   // worker.py:85 -> :45
                }
                if (left == 0) break;
            }
            f.flush(r);
            FPHK(f, 12);
            if (need_adv) continue;
        }
        if (left == 0) break;
        ep++;
        if constexpr (use_prio) { if (ep < episodes) { nv_last = S.T(); set_prio(ep); } }
    }
    PH_FLUSH(lane);
    FPHK(f, 13);
    FPH_FLUSH(f, lane);
    const int64_t steps = (int64_t)(left0 - left);
    if (lane == 0 && steps_out) steps_out[e] = steps;
    h.d = d0 + (uint64_t)steps;
    {   // Hdr::max_arrival (see k_rollout_random)
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
