// rollout_fast_g.hpp -- the register-resident persistent rollout kernel (see rollout_fast.hpp, rollout_fast_mc.hpp) for the MID-SIZE
// class: any batch with A <= 128 agents and T <= 256 tasks that is not one of the exact shapes, uniform or ragged
// (env/task_env.py:57-65 draws the sizes from ranges).  Included by dcmrta_env.hip after rollout_fast_mc.hpp.
//
// The template bounds the number of 64-lane chunks -- NAC agent chunks (1 or 2), NTC task chunks (2, 3 or 4) -- the sizes themselves
// are runtime values (this env's own rA / rT from the ragged size table, the batch's layout dims for the record).  Lane l owns the
// agents l, 64 + l and the tasks l, 64 + l, ...: one register set per chunk, every loop over chunks unrolled, chunks beyond the
// env's own sizes never visited (their lane masks are empty).
//
// What differs from rollout_fast_mc.hpp:
//   * two agent chunks: agent bitmasks (group, members, `gone`) are one 64-bit word per chunk, a wave-uniform agent id a is the
//     lane a & 63 of chunk a >> 6 (one scalar branch per chunk where a lane is read or written);
//   * the depot is not a pseudo-task in a free lane (T = 64, 128, ... have none): its coordinates are wave-uniform scalars and lane 0
//     writes its observation row;
//   * the general code is Sim<128,256,true>, whose LDS image has the batch's own layout and holds the member-arrival slots
//     f64[M][pT] itself: the fast path uses them in place (no parking in the HBM record), and time_start, the wake-up times and the
//     abandonment counts stay in their image sections as well.  Registers: everything a decision reads of an agent or a task.
#pragma once
#include <utility>
#ifndef DCM_G_WAVES
#define DCM_G_WAVES 2
#endif

template <int NAC, int NTC, bool OBS>
struct FastG {
    static_assert(NAC >= 1 && NAC <= 2 && NTC >= 1 && NTC <= 4, "chunk bounds of the <128,256> class");
    using SimT = Sim<128, 256, true>;                  // the general code: bounded sizes, runtime layout Lay{pA,pT}

    SimT S;
    double* dummy;                                     // 64 doubles of LDS nobody reads (the removal path's discarded writes)
    double depx, depy;                                 // depot (wave-uniform)
    // incremental task_update state (wave-uniform)
    mutable uint32_t touched = 0;                      // lane chunks the previous task_update call touched
    mutable int n_infeas = 0;                          // tasks that are not feasible

    struct R {
        double ax[NAC], ay[NAC], arr[NAC], nd[NAC];            // agent: location, arrival_time[-1], next_decision
        int32_t cur[NAC]; uint32_t ai[NAC];                    //        route[-1], ainfo word
        double cts[NAC], cend[NAC]; bool cfeas[NAC];           //        of route[-1]: time_start, time_start + duration (== time_finish
                                                               //        once it is feasible; 0.0 + duration before), feasible flag
        uint32_t ti[NTC];                                      // task (per chunk): tinfo word
        uint64_t ids[NTC];                                     //        ordered member ids
        double tx[NTC], ty[NTC]; float durf[NTC];              //        instance; the duration as the observation holds it
    };

    __device__ __forceinline__ bool in_agent(int c, int lane) const { return c * 64 + lane < S.rA; }
    __device__ __forceinline__ int aidx(int c, int lane) const { return in_agent(c, lane) ? c * 64 + lane : 0; }     // clamped
    __device__ __forceinline__ bool in_task(int c, int lane) const { return c * 64 + lane < S.rT; }
    __device__ __forceinline__ int tidx(int c, int lane) const { return in_task(c, lane) ? c * 64 + lane : 0; }      // clamped
    __device__ __forceinline__ static uint64_t cmask(int n, int c) {      // lanes of chunk c below n
        const int m = n - 64 * c;
        return m >= 64 ? ~0ull : (m <= 0 ? 0ull : ((1ull << m) - 1ull));
    }
    __device__ __forceinline__ uint64_t amask(int c) const { return cmask(S.rA, c); }
    __device__ __forceinline__ uint64_t tmask(int c) const { return cmask(S.rT, c); }
    __device__ __forceinline__ double* slots() const { return S.marr(); }
    __device__ __forceinline__ static int nth(uint64_t m, int idx) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        return __popcll(__ballot(rank <= idx)) - 1;
    }
    // agent id of the idx-th set bit of a per-chunk agent mask
    __device__ __forceinline__ static int nth_agent(const uint64_t (&m)[NAC], int idx) {
        if constexpr (NAC == 1) return nth(m[0], idx);
        else {
            const int n0 = __popcll(m[0]);
            return idx < n0 ? nth(m[0], idx) : 64 + nth(m[1], idx - n0);
        }
    }
    __device__ __forceinline__ static double rl(double v, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    }
    __device__ __forceinline__ static uint64_t rl(uint64_t v, int src) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    }
    __device__ __forceinline__ static uint32_t rl(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
    // value of a per-chunk register array at the wave-uniform position (chunk, lane): every chunk's lane is read, scalar selects pick
    // (branches around two v_readlane cost more than the reads)
    __device__ __forceinline__ static uint32_t sel(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
    __device__ __forceinline__ static uint64_t sel(bool c, uint64_t a, uint64_t b) { return c ? a : b; }
    __device__ __forceinline__ static double sel(bool c, double a, double b) {
        return __longlong_as_double((long long)sel(c, (uint64_t)__double_as_longlong(a), (uint64_t)__double_as_longlong(b)));
    }
    template <class V, int N>
    __device__ __forceinline__ static V rlc(const V (&v)[N], int chunk, int src) {
        V out = rl(v[0], src);
#pragma unroll
        for (int c = 1; c < N; c++) out = sel(c == chunk, rl(v[c], src), out);
        return out;
    }

    // ------------------------------------------------------------------------------ registers <-> LDS image
    __device__ __forceinline__ void init() {
        depx = uni(((const Hdr*)S.base)->depot_x); depy = uni(((const Hdr*)S.base)->depot_y);
    }
    __device__ __forceinline__ void load_consts(R& r, int lane) const {
#pragma unroll
        for (int c = 0; c < NTC; c++) {
            const int t = tidx(c, lane);
            r.tx[c] = S.tx()[t]; r.ty[c] = S.ty()[t]; r.durf[c] = (float)S.tdur()[t];
        }
    }
    __device__ __forceinline__ void reload(R& r, int lane) const {
#pragma unroll
        for (int c = 0; c < NAC; c++) {
            const int a = aidx(c, lane);
            r.ax[c] = S.ax()[a]; r.ay[c] = S.ay()[a]; r.arr[c] = S.arr()[a]; r.nd[c] = S.nd()[a];
            r.cur[c] = S.cur()[a]; r.ai[c] = S.ainfo()[a];
            const int K = r.cur[c] < 0 ? 0 : r.cur[c];
            r.cts[c] = S.ts()[K]; r.cend[c] = r.cts[c] + S.tdur()[K]; r.cfeas[c] = S.tinfo()[K] & T_FEAS;
        }
        int ninf = 0;
        uint32_t all = 0;
#pragma unroll
        for (int c = 0; c < NTC; c++) {
            const int t = tidx(c, lane);
            r.ti[c] = S.tinfo()[t]; r.ids[c] = S.mids()[t];
            if (in_task(c, lane)) S.wake()[t] = -__builtin_inff();               // every chunk is due at the next new event
            const uint64_t tm = tmask(c);
            ninf += __popcll(__ballot(!(r.ti[c] & T_FEAS)) & tm);
            all |= tm ? (1u << c) : 0u;
        }
        n_infeas = ninf;
        touched = all;
        WSYNC();
    }
    __device__ __forceinline__ void flush(const R& r, int lane) const {
        WSYNC();
#pragma unroll
        for (int c = 0; c < NAC; c++) if (in_agent(c, lane)) {
            const int a = c * 64 + lane;
            S.ax()[a] = r.ax[c]; S.ay()[a] = r.ay[c]; S.arr()[a] = r.arr[c]; S.nd()[a] = r.nd[c];
            S.cur()[a] = r.cur[c]; S.ainfo()[a] = r.ai[c];
        }
#pragma unroll
        for (int c = 0; c < NTC; c++) if (in_task(c, lane)) {
            const int t = c * 64 + lane;
            S.tinfo()[t] = r.ti[c]; S.mids()[t] = r.ids[c];
        }
        if (lane == 0) S.inc_state()[1] = -1;          // the general code's own incremental state: next call visits every task
        WSYNC();
    }

    // ------------------------------------------------------------------------------ task_update, env/task_env.py:245-281
    // One lane chunk (see Fast::task_update for the lane code; Sim::task_update for the incremental visiting rules).
    template <int C>
    __device__ __forceinline__ void tu_chunk(R& r, double now, double mwt, int lane, uint32_t& touched_out) const {
        const bool inT = in_task(C, lane);
        const int t = tidx(C, lane);
        const int PT_ = S.PT();
        uint32_t info = r.ti[C];
        const bool feas0 = info & T_FEAS;
        const int req = info & 0xFF, n = (info >> 16) & 0xFF;                    // :250
        double av[M];
#pragma unroll
        for (int j = 0; j < M; j++) av[j] = slots()[j * PT_ + t];                // :251 (unused slots hold NaN)
        const double tfin = S.tf()[t], dur = S.tdur()[t];
        const int status = req - n;                                              // :252
        double mx = av[0], mn = av[0];
#pragma unroll
        for (int j = 1; j < M; j++) { mx = nanmax2(mx, av[j]); mn = nanmin2(mn, av[j]); }
        const bool le0 = status <= 0;                                            // :254
        const bool ok = le0 && (mx - mn <= mwt);                                 // :255
        const double thr = mx - mwt;                                             // :262
        const bool any_drop = inT && !feas0 && (le0 ? (!ok && mn <= thr) : (now - mn >= mwt));
        const bool becomes = inT && !feas0 && ok;                                // :256-258
        const double ntf = becomes ? mx + dur : tfin;
        if (becomes) { S.ts()[t] = mx; S.tf()[t] = ntf; }                        // time_start, time_finish :256-257
        int nn = n;
        const uint64_t dmask = __ballot(any_drop);
        if (dmask) {
            uint64_t gone[NAC];
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) gone[ac] = 0ull;
            // (see Fast::task_update: the rule that is not in play is skipped by a scalar branch; leavers write to a dummy slot)
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
                const uint32_t idl = (uint32_t)r.ids[C], idh = (uint32_t)(r.ids[C] >> 32);
                uint32_t nids = 0;
                double* const row0 = slots() + t;
                double* const dump = dummy + lane;
#pragma unroll
                for (int j = 0; j < M; j++) row0[j * PT_] = __builtin_nan("");
#pragma unroll
                for (int j = 0; j < M; j++) {
                    const bool kp = (keep >> j) & 1u, lv = (drop >> j) & 1u;
                    const int kj = __popc(keep & ((1u << j) - 1u));
                    const uint32_t id = (j < 4 ? (idl >> (8 * j)) : idh) & 0xFFu;
                    nids |= kp ? (id << (8 * kj)) : 0u;
#pragma unroll
                    for (int ac = 0; ac < NAC; ac++) gone[ac] |= (lv && (int)(id >> 6) == ac) ? (1ull << (id & 63u)) : 0ull;
                    *(kp ? row0 + kj * PT_ : dump) = av[j];
                }
                r.ids[C] = (uint64_t)nids;
                S.tnab()[t] += (uint32_t)__popc(drop);                           // abandoned_agent.append :265/:271
                nn = __popc(keep);
            }
            uint64_t todo = dmask;
            do {
                const int b = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1ull;
                const int tk_ = C * 64 + b;
#pragma unroll
                for (int ac = 0; ac < NAC; ac++) {
                    const uint64_t g = rl(gone[ac], b);
                    if ((g >> lane) & 1ull) {                                    // this lane's agent of chunk ac was dropped by task tk_
                        const int a = ac * 64 + lane;
                        const uint32_t nth_ = r.ai[ac] >> 16;
                        r.ai[ac] += 1u << 16;
                        if (nth_ < (uint32_t)AB_CAP) S.ablog()[a * AB_CAP + nth_] = (uint16_t)tk_;
                        else { const uint32_t ci = (uint32_t)(a * S.T() + tk_); atomicAdd((uint32_t*)S.abcnt() + (ci >> 1), 1u << (16 * (ci & 1))); }
                        if (r.cur[ac] == tk_) r.ai[ac] &= ~A_MEMBER;
                    }
                }
            } while (todo);
        }
        const uint32_t info_i = ((info | (ok ? T_FEAS : 0u)) & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)nn << 16);
        const uint32_t info_f = info | ((now >= tfin) ? T_FIN : 0u);             // :273-274
        info = feas0 ? info_f : info_i;
        r.ti[C] = info;
        // when can the time alone change this task next?  (see Sim::task_update)
        double w = (info & T_FEAS) ? ((info & T_FIN) ? __builtin_inf() : (feas0 ? tfin : mx + dur)) : mn + mwt;
        w = (w == w) ? w : __builtin_inf();
        if (inT) S.wake()[t] = any_drop ? -__builtin_inff() : __double2float_rd(w);
        // agents whose current task has just become feasible refresh their cache of it (wave-uniform pass, rare)
        uint64_t bmask = __ballot(becomes);
        n_infeas -= __popcll(bmask);
        while (bmask) {
            const int b = __ffsll((unsigned long long)bmask) - 1;
            bmask &= bmask - 1ull;
            const double ts_k = rl(mx, b), tf_k = rl(ntf, b);
#pragma unroll
            for (int ac = 0; ac < NAC; ac++)
                if (r.cur[ac] == C * 64 + b) { r.cfeas[ac] = true; r.cts[ac] = ts_k; r.cend[ac] = tf_k; }
        }
        // a freshly feasible task only changes again at this `now` if it is already over (:273 is evaluated one call later)
        if (__ballot(any_drop || (becomes && now >= mx + dur))) touched_out |= 1u << C;
    }

    template <int... Cs>
    __device__ __forceinline__ void tu_chunks(R& r, double now, double mwt, int lane, uint32_t todo, uint32_t& t_out,
                                              std::integer_sequence<int, Cs...>) const {
        ((((todo >> Cs) & 1u) ? tu_chunk<Cs>(r, now, mwt, lane, t_out) : (void)0), ...);   // one wave-uniform branch per chunk
    }
    // kc: chunk of the task the agents have just joined (-1: depot), -3: the call of a new event (the time has moved)
    __device__ __forceinline__ void task_update(R& r, double now, double mwt, int lane, int kc) const {
        uint32_t todo = touched;
        if (kc == -3) {
#pragma unroll
            for (int c = 0; c < NTC; c++) todo |= (__ballot(in_task(c, lane) && now >= (double)S.wake()[tidx(c, lane)]) != 0ull) ? (1u << c) : 0u;
        } else if (kc >= 0) todo |= 1u << kc;
        uint32_t t_out = 0;
        tu_chunks(r, now, mwt, lane, todo, t_out, std::make_integer_sequence<int, NTC>{});
        touched = t_out;
        if (n_infeas == 0) {                                                     // depot :277-280 (np.all(feasible))
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) if ((r.ai[ac] & A_INDEPOT) && now >= r.arr[ac]) r.ai[ac] |= A_RETURNED;
        }
    }

    // ------------------------------------------------------------------------------ agent_update, env/task_env.py:207-243
    __device__ __forceinline__ void agent_update(R& r, double now, double mwt) const {
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) {
            const int c = r.cur[ac];
            const bool member = r.cfeas[ac] && (r.ai[ac] & A_MEMBER);            // :229-230
            const double ndv = (c == -1) ? __builtin_nan("") : (member ? r.cend[ac] : r.arr[ac] + mwt);   // :226,:231,:235,:238
            const uint32_t as = member ? ((r.ai[ac] & A_ASSIGNED) | ((now >= r.cts[ac]) ? A_ASSIGNED : 0u)) : 0u;   // :232-240
            r.nd[ac] = (c != -2) ? ndv : r.nd[ac];                               // :209
            r.ai[ac] = (c >= 0) ? ((r.ai[ac] & ~A_ASSIGNED) | as) : r.ai[ac];    // depot leaves `assigned` untouched (Q6)
        }
    }

    // ------------------------------------------------------------------------------ observation, worker.py:57-68
    struct BM { uint64_t w[NTC]; };
    // ag / tk / mk: the env's rows in the three output tensors (wave-uniform)
    __device__ __forceinline__ BM observe(const R& r, double now, int leader, int lane, float* __restrict__ ag, float* __restrict__ tk,
                                          uint8_t* __restrict__ mk) const {
        const double lx = rlc(r.ax, leader >> 6, leader & 63), ly = rlc(r.ay, leader >> 6, leader & 63);
        if constexpr (OBS) {
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) {
                const bool on = r.cur[ac] >= 0;                                  // :168
                const double x = r.arr[ac] - now, w = now - r.arr[ac], rem = r.cend[ac] - now;
                const double travel = (on && x > 0.) ? x : 0.;                   // :169
                const double waiting = (on && now <= r.cts[ac] && w > 0.) ? w : 0.;      // :170
                const double remaining = (on && now >= r.cts[ac] && rem > 0.) ? rem : 0.;   // :171
                const float f0 = (float)travel, f1 = (float)remaining, f2 = (float)waiting;
                const float f3 = (float)(lx - r.ax[ac]), f4 = (float)(ly - r.ay[ac]), f5 = (r.ai[ac] & A_ASSIGNED) ? 1.f : 0.f;
                if (in_agent(ac, lane)) {                                        // :176-177
                    float* agrow = (ag + 6 * lane) + 6 * 64 * ac;                // (one per-lane base; the chunk offset is an immediate)
                    agrow[0] = f0; agrow[1] = f1; agrow[2] = f2; agrow[3] = f3; agrow[4] = f4; agrow[5] = f5;
                }
            }
        }
        BM bm;
        bool unf[NTC];
#pragma unroll
        for (int c = 0; c < NTC; c++) {
            const uint32_t info = r.ti[c];
            unf[c] = !(info & T_FEAS) && (int)(int8_t)((info >> 8) & 0xFF) > 0;  // :199
            bm.w[c] = __ballot(unf[c]) & tmask(c);
        }
        if constexpr (OBS) {
            uint64_t any = 0ull;
#pragma unroll
            for (int c = 0; c < NTC; c++) any |= bm.w[c];
#pragma unroll
            for (int c = 0; c < NTC; c++) {
                const uint32_t info = r.ti[c];
                const uint8_t mv = unf[c] ? 0 : 1;                               // :193
                const float g0 = (float)(int)(int8_t)((info >> 8) & 0xFF), g1 = (float)(info & 0xFF), g2 = r.durf[c];
                const float g3 = (float)(r.tx[c] - lx), g4 = (float)(r.ty[c] - ly);   // :185-188
                if (in_task(c, lane)) {
                    (mk + lane)[c * 64 + 1] = mv;
                    float* row = (tk + 5 * lane) + 5 * (c * 64 + 1);
                    row[0] = g0; row[1] = g1; row[2] = g2; row[3] = g3; row[4] = g4;
                }
            }
            if (lane == 0) {                                                     // depot row; its mask byte is False iff every task is masked
                mk[0] = (any == 0ull) ? 0 : 1;
                tk[0] = 0.f; tk[1] = 0.f; tk[2] = 0.f; tk[3] = (float)(depx - lx); tk[4] = (float)(depy - ly);
            }
        }
        return bm;
    }

    // ------------------------------------------------------------------------------ one decision (see Fast::decide / apply)
    __device__ __forceinline__ int decide(R& r, HdrRegs& h, const KP& P, int lane, uint64_t k1, float* ag, float* tk, uint8_t* mk) const {
        uint64_t gm[NAC];
        int glen = 0;
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) {
            gm[ac] = __ballot((int)((r.ai[ac] >> 8) & 0xFFu) == h.cur_group) & amask(ac);
            glen += __popcll(gm[ac]);
        }
        // (unreachable: groups are never empty.  No early return -- see Fast::decide: an exit from the middle of a decision keeps a
        //  second copy of the whole lane-owned state alive)
        if (glen == 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; glen = 1; gm[0] = 1ull; }
        const int leader = nth_agent(gm, below((uint32_t)(k1 >> 32), glen));
        const double now = h.now;
        const BM bm = observe(r, now, leader, lane, ag, tk, mk);
        // uniform-random valid action (protocol slot 1): valid = ascending unmasked action ids
        int nv = 0;
#pragma unroll
        for (int c = 0; c < NTC; c++) nv += __popcll(bm.w[c]);
        int kc = -1, tl = 0;                                                     // chunk / lane of the target task (kc < 0: depot)
        if (nv) {
            int idx = below((uint32_t)k1, nv);
#pragma unroll
            for (int c = 0; c < NTC; c++) {
                const int n_c = __popcll(bm.w[c]);
                if (kc < 0 && idx < n_c) { kc = c; tl = nth(bm.w[c], idx); }
                idx -= n_c;
            }
        }
        const int k = kc < 0 ? -1 : kc * 64 + tl;                                // task id, -1 = depot
        double tx_ = depx, ty_ = depy, dur_k = 0.;
        uint32_t kinfo = 0; uint64_t ids = 0ull;
        if (kc >= 0) {
            tx_ = rlc(r.tx, kc, tl); ty_ = rlc(r.ty, kc, tl);
            kinfo = rlc(r.ti, kc, tl); ids = rlc(r.ids, kc, tl);
        }
        // TaskEnv.step :326-342
        uint64_t rest[NAC], mm[NAC];
        int mypos[NAC];
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) {
            const uint64_t lb = (ac == (leader >> 6)) ? (1ull << (leader & 63)) : 0ull;
            rest[ac] = gm[ac] & ~lb;                                             // :328
            mm[ac] = lb;
            mypos[ac] = 0;
        }
        int rlen = glen - 1;
        uint64_t mlist = (uint64_t)(uint32_t)leader;
        int nm = 1;
        if (kc < 0) {                                                            // vacancy = len(group) :327 (Q9)
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) mm[ac] |= rest[ac];
            nm += rlen; rlen = 0;
        } else if (rlen != 0) {
            const int vacancy = (int)(int8_t)((kinfo >> 8) & 0xFF);              // :327 (may be stale)
            const int nf = (vacancy > 1) ? ((vacancy - 1 < rlen) ? vacancy - 1 : rlen) : 0;   // :330-331
            uint64_t kk = k1;
            for (int j = 0; j < nf; j++) {                                       // :331 choice without replacement
                if ((j & 1) == 0) kk = mix64(kk + GAMMA);
                const uint32_t rr = (j & 1) ? (uint32_t)kk : (uint32_t)(kk >> 32);
                const int f = nth_agent(rest, below(rr, rlen));
                const int fl = f & 63;
                rlen--;                                                          // :332-333
                mlist |= (uint64_t)(uint32_t)f << (8 * nm);
#pragma unroll
                for (int ac = 0; ac < NAC; ac++) if (ac == (f >> 6)) {
                    rest[ac] &= ~(1ull << fl);
                    mm[ac] |= 1ull << fl;
                    mypos[ac] = lane == fl ? nm : mypos[ac];
                }
                nm++;
            }
        }
        // agent_step :300-324 on ALL lanes (fp64 VALU work with fewer than 16 active lanes is 4x slower on gfx950)
        double d[NAC], arrv[NAC];
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) {
            d[ac] = dist2(r.ax[ac], r.ay[ac], tx_, ty_);
            arrv[ac] = now + over_velocity(d[ac]);                               // :315,:318
            asm volatile("" : "+v"(d[ac]), "+v"(arrv[ac]));
        }
        int n = 0, slot[NAC];
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) slot[ac] = 0;
        if (kc >= 0) {
            // :321-322 members.append unless already listed (Q4: a re-joining agent keeps its slot, its arrival is overwritten)
            n = (kinfo >> 16) & 0xFF;
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) slot[ac] = n + mypos[ac];
            dur_k = S.tdur()[k];
            uint64_t listed[NAC];                                                // the task's members as agent bitmasks (wave-uniform)
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) listed[ac] = 0ull;
            for (int j = 0; j < n; j++) {
                const uint32_t id = (uint32_t)(ids >> (8 * j)) & 0xFFu;
#pragma unroll
                for (int ac = 0; ac < NAC; ac++) listed[ac] |= ((int)(id >> 6) == ac) ? (1ull << (id & 63u)) : 0ull;
            }
            uint64_t again = 0ull;
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) again |= listed[ac] & mm[ac];
            if (again) {
                for (int j = 0; j < nm; j++) {
                    const int m = (int)((mlist >> (8 * j)) & 0xFF);
                    const uint64_t x = ids ^ (0x0101010101010101ull * (uint64_t)(uint32_t)m);
                    uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
                    z &= (n >= 8) ? ~0ull : ((1ull << (8 * n)) - 1ull);
                    int pos;
                    if (z) pos = (__ffsll((unsigned long long)z) - 1) >> 3;
                    else { pos = n++; ids |= (uint64_t)(uint32_t)m << (8 * pos); }
#pragma unroll
                    for (int ac = 0; ac < NAC; ac++) if (ac * 64 + lane == m) slot[ac] = pos;
                }
            } else {
                ids |= mlist << (8 * n);                                         // bytes above n are always zero
                n += nm;
            }
        }
        const int PT_ = S.PT();
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) {
            const bool mem = (mm[ac] >> lane) & 1ull;
            if (mem) {
                // :317 travel_dist += d: read by the terminal metrics only, so an LDS atomic (same IEEE addition, no round trip to wait for)
                __hip_atomic_fetch_add(&S.tdist()[ac * 64 + lane], d[ac], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                r.arr[ac] = arrv[ac];
                r.ax[ac] = tx_; r.ay[ac] = ty_;                                  // :320
                r.cur[ac] = k;                                                   // :314
                r.ai[ac] = (r.ai[ac] & ~(A_GRP | A_MEMBER)) | (kc < 0 ? A_INDEPOT : A_MEMBER);
                if (kc >= 0) {
                    slots()[slot[ac] * PT_ + k] = arrv[ac];
                    // the new current task as agent_update / the observation read it: a task the device policy can pick is not
                    // feasible yet, and time_start / time_finish of such a task are still the 0.0 of clear_decisions (:131, :256-257)
                    r.cfeas[ac] = false; r.cts[ac] = 0.0; r.cend[ac] = 0.0 + dur_k;
                }
            }
        }
        // A QUIET join (see FastM::decide): the task still lacks members and its lane chunk was at a fixed point for this `now` --
        // the chunk visit is skipped, the task's lane takes the new status and pulls its wake-up time forward itself.
        const int status_k = (int)(kinfo & 0xFFu) - n;
        const bool quiet = kc >= 0 && status_k > 0 && !((touched >> (kc < 0 ? 0 : kc)) & 1u);
        const float wj = __double2float_rd(rlc(arrv, leader >> 6, leader & 63) + P.mwt);   // (wave-uniform: a group arrives together)
        if (kc >= 0 && lane == tl) {
            const uint32_t nti = quiet ? ((kinfo & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(status_k & 0xFF) << 8) | ((uint32_t)n << 16))
                                       : ((kinfo & ~0x00FF0000u) | ((uint32_t)n << 16));
#pragma unroll
            for (int c = 0; c < NTC; c++) if (c == kc) { r.ids[c] = ids; r.ti[c] = nti; }
            if (quiet) {
                float* wp = &S.wake()[k];
                *wp = fminf(*wp, wj);
            }
        }
        WSYNC();
        task_update(r, now, P.mwt, lane, quiet ? -1 : kc);                       // worker.py:74
        agent_update(r, now, P.mwt);                                             // worker.py:76
        return rlen;
    }

    // ------------------------------------------------------------------------------ next event (see Fast::next_event)
    __device__ __forceinline__ bool next_event(R& r, HdrRegs& h, const KP& P, int lane) const {
        if (h.now >= P.max_time) return false;
        double ndv[NAC];
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) ndv[ac] = in_agent(ac, lane) ? r.nd[ac] : __builtin_nan("");
        double m = ndv[0];
#pragma unroll
        for (int ac = 1; ac < NAC; ac++) m = nanmin2(m, ndv[ac]);
        const double tmin = wave_nanmin(m);                                      // :287
        if (!(tmin == tmin)) return false;
        h.now = tmin;                                                            // worker.py:49
        bool dec[NAC];
        uint64_t dm[NAC];
        int ndec = 0, first = -1;
#pragma unroll
        for (int ac = 0; ac < NAC; ac++) {
            dec[ac] = (ndv[ac] == tmin);                                         // :288 exact ==
            dm[ac] = __ballot(dec[ac]);
            if (first < 0 && dm[ac]) first = ac * 64 + __ffsll((unsigned long long)dm[ac]) - 1;
            ndec += __popcll(dm[ac]);
        }
        bool same = true;
        if (ndec > 1) {
            const double x0 = rlc(r.ax, first >> 6, first & 63), y0 = rlc(r.ay, first >> 6, first & 63);
            uint64_t diff = 0ull;
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) diff |= __ballot(dec[ac] && !(r.ax[ac] == x0 && r.ay[ac] == y0));
            same = diff == 0ull;
        }
        if (same) {
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) r.ai[ac] = (r.ai[ac] & ~A_GRP) | (dec[ac] ? (1u << 8) : 0u);
            h.n_groups = 1;
        } else {
            bool todo[NAC];                                                      // groups in ascending (x, then y) order :293
            uint32_t gid[NAC];
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) { todo[ac] = dec[ac]; gid[ac] = 0; }
            int g = 0;
            for (;;) {
                double vx = todo[0] ? r.ax[0] : __builtin_nan("");
#pragma unroll
                for (int ac = 1; ac < NAC; ac++) vx = nanmin2(vx, todo[ac] ? r.ax[ac] : __builtin_nan(""));
                const double mxv = wave_nanmin(vx);
                if (!(mxv == mxv)) break;
                double vy = (todo[0] && r.ax[0] == mxv) ? r.ay[0] : __builtin_nan("");
#pragma unroll
                for (int ac = 1; ac < NAC; ac++) vy = nanmin2(vy, (todo[ac] && r.ax[ac] == mxv) ? r.ay[ac] : __builtin_nan(""));
                const double myv = wave_nanmin(vy);
                g++;
#pragma unroll
                for (int ac = 0; ac < NAC; ac++)
                    if (todo[ac] && r.ax[ac] == mxv && r.ay[ac] == myv) { gid[ac] = (uint32_t)g; todo[ac] = false; }
            }
#pragma unroll
            for (int ac = 0; ac < NAC; ac++) r.ai[ac] = (r.ai[ac] & ~A_GRP) | (gid[ac] << 8);
            h.n_groups = g;
        }
        task_update(r, tmin, P.mwt, lane, -3);                                   // worker.py:50
        agent_update(r, tmin, P.mwt);                                            // worker.py:51
        h.empty_passes = 0;
        h.cur_group = 1;
        return true;
    }
};

// Same contract as k_rollout_random (see there); OBS: all three observation buffers given / none of them.
template <int NAC, int NTC, bool OBS>
__global__ __launch_bounds__(WAVE, DCM_G_WAVES) void k_rollout_fast_g(int A, int T, int PA, int PT, KP P, unsigned char* state, int episodes,
                                                        float* agents_out, float* tasks_out, uint8_t* mask_out,
                                                        int64_t* steps_out, double* summary, uint16_t* ablog,
                                                        const int32_t* sizes, int64_t budget_all, const int64_t* budget_in,
                                                        unsigned char* gscr, double* retlog, int retcap) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<128, 256, true>(sizes, e, A, T, eA, eT);
    using F = FastG<NAC, NTC, OBS>;
    using SimT = typename F::SimT;
    SimT S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = gscr + (size_t)e * L.scratch_bytes();
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    typename SimT::XY xy;
    S.template load_record<true, false>(rec, lane, xy);
    S.set_ablog(ablog, e, A, T, lane);
    S.set_retlog(retlog, retcap, e, lane);
    if (lane == 0) S.inc_state()[1] = -1;
    WSYNC();
    HdrRegs h = load_hdr(smem);
    // (the launch asks for 512 bytes of LDS behind everything the general code uses: the dummy slots)
    F f{S, (double*)(smem + SimT::lds_image_bytes(L))};
    f.init();
    float* ag = nullptr; float* tk = nullptr; uint8_t* mk = nullptr;
    if constexpr (OBS) {
        ag = agents_out + (size_t)e * 6 * A;
        tk = tasks_out + (size_t)e * 5 * (T + 1);
        mk = mask_out + (size_t)e * (T + 1);
        S.write_pad_obs(lane, A, T, ag, tk, mk);
    }
    double* row = summary + (size_t)e * 8;
    constexpr int NO_BUDGET = 0x7FFFFFFF;
    int64_t bud = budget_in ? budget_in[e] : budget_all;
    const int left0 = uni((int)((bud < 0 || bud >= NO_BUDGET) ? NO_BUDGET : bud));
    int left = left0;
    uint64_t gd = h.seed + GAMMA * (h.d + 1);
    const uint64_t d0 = h.d;
    typename F::R r;
    f.load_consts(r, lane);
    constexpr uint32_t ERR = DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER | DCM_FLAG_BAD_INSTANCE;
    PH_DECL;
    int ep = 0;
    bool need_adv = false;
    for (;;) {
        if (!need_adv) {         // head of an episode slot (the `for ep` of k_rollout_random)
            if (ep >= episodes) break;
            if (h.flags & DCM_FLAG_DONE) {
                if (h.flags & ERR) break;
                if (left == 0) break;
                S.reset_state(h, lane);
                need_adv = true;
            }
        }
        if (need_adv) {
            S.advance(h, P, lane, row PH_PASS);
            need_adv = false;
            h.now = uni(h.now); h.flags = uni(h.flags); h.cur_group = uni(h.cur_group); h.n_groups = uni(h.n_groups);
            h.empty_passes = uni(h.empty_passes);
        }
        if (!(h.flags & DCM_FLAG_DONE) && left != 0) {
            WSYNC();
            f.reload(r, lane);
            for (;;) {
                const uint64_t k1 = mix64(gd);
                const int rlen = f.decide(r, h, P, lane, k1, ag, tk, mk);
                if (h.flags & DCM_FLAG_DONE) break;
                gd += GAMMA;
                left--;
                if (rlen == 0) {                                                  // worker.py:53 else same group, next leader
                    if (h.cur_group < h.n_groups) h.cur_group++;                  // worker.py:52 next group
                    else if (!f.next_event(r, h, P, lane)) { need_adv = true; break; }   // worker.py:85 -> :45
                }
                if (left == 0) break;
            }
            f.flush(r, lane);
            if (need_adv) continue;
        }
        if (left == 0) break;
        ep++;
    }
    PH_FLUSH(lane);
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
