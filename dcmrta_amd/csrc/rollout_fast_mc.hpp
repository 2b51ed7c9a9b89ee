// rollout_fast_mc.hpp -- the register-resident persistent rollout kernel (see rollout_fast.hpp) for the EXACT multi-chunk shapes with
// one agent chunk: A <= 64 agents, T > 64 tasks in NTC lane chunks (BASELINE configs[3]: 50A/200T = 4 chunks).
// Included by dcmrta_env.hip after rollout_fast.hpp.
//
// What differs from the one-chunk kernel:
//   * lane l owns the tasks l, 64 + l, 128 + l, ... (one register set per chunk, loops over chunks fully unrolled); the free lane
//     63 of the LAST chunk owns the depot pseudo-task (needs T % 64 != 0);
//   * task_update visits lane chunks incrementally exactly like Sim::task_update (INC): after an agent_step the chunk of the joined
//     task plus the chunks the previous call touched; at a new event the chunks with a due wake-up time plus those; the wake-up
//     times and the count of infeasible tasks live in registers;
//   * agents CACHE what agent_update / the agent observation read of their current task (feasible flag, time_start, time_finish,
//     duration): set when the agent joins, refreshed -- by a wave-uniform pass over the newly feasible tasks -- when its task
//     becomes feasible.  agent_update therefore reads no LDS at all, and the fast path needs LDS for ONE thing only: the member
//     arrival slots f64[M][T] that the joining agents' lanes write and the task's lane reads;
//   * those 8000 bytes are time-multiplexed with the general code's LDS image: Sim<CA,CT,false,MG> keeps the member arrivals in the
//     env's HBM record and everything else in a 11.3 KB image; while the fast path runs, the image bytes [64, 64 + 40 T) -- agent
//     arrays, time arrays, member ids, status words: all of it register-resident here -- hold the arrival slots instead.  Entering
//     the general code (once or twice per episode: terminal metrics, reset, first event, events nobody can decide at) parks the slots
//     in the record and rebuilds the image from the registers; leaving it does the reverse.  LDS per env stays 11.3 KB (the
//     round-3 kernel's figure), and the 7 KB of member-arrival re-reads per decision of that kernel are gone.
#pragma once
#include <utility>
#ifndef DCM_MC_WAVES
#define DCM_MC_WAVES 3
#endif

template <int CA, int CT, bool OBS>
struct FastM {
    static constexpr int NTC = (CT + 63) / 64;
    static constexpr int DC = NTC - 1, DL = 63;       // chunk / lane of the depot pseudo-task
    static_assert(CA >= 1 && CA <= 64 && CT > 64 && (CT % 64) != 0, "one agent chunk, several task chunks, a free lane in the last one");
    using SimT = Sim<CA, CT, false, true>;            // the general code: exact shape, member arrivals in the HBM record
    static constexpr Lay L{CA, CT};
    static constexpr uint32_t SLOTS_OFF = 64;         // LDS bytes [64, 64 + 40 T): the arrival slots while the fast path runs
    static_assert(SLOTS_OFF + 8u * M * CT <= L.tnab() - SimT::MSH, "the slots must only cover register-resident sections of the image");
    static constexpr uint64_t AM = CA >= 64 ? ~0ull : ((1ull << CA) - 1ull);
    __host__ __device__ static constexpr uint64_t tmask(int c) { return (c < NTC - 1) ? ~0ull : ((1ull << (CT - 64 * (NTC - 1))) - 1ull); }

    SimT S;
    unsigned char* rec;                                // the env's HBM record
    bool inA, isD;
    int la;
    // incremental task_update state (wave-uniform)
    mutable uint32_t touched = 0;                      // lane chunks the previous task_update call touched
    mutable int n_infeas = 0;                          // tasks that are not feasible

    struct R {
        double ax, ay, arr, nd, td;                    // agent: location, arrival_time[-1], next_decision, travel_dist
        int32_t cur; uint32_t ai;                      //        route[-1], ainfo word
        double cts, cend; bool cfeas;                  //        of route[-1]: time_start, time_start + duration (== time_finish once it
                                                       //        is feasible; 0.0 + duration before), feasible flag
        uint32_t ti[NTC];                              // task (per chunk): tinfo word
        uint64_t ids[NTC];                             //        ordered member ids
        double tx[NTC], ty[NTC]; float durf[NTC];      //        instance (depot lane of the last chunk: depot x, y, 0); the duration as
                                                       //        the observation holds it (the fp64 value stays in the LDS image)
    };
    // Not in registers (the kernel needs three waves per SIMD): time_start -- written once, when the task becomes feasible -- sits in
    // f64[T] of LDS behind the image; time_finish is not stored at all on the fast path: it is time_start + duration (:256-257, the
    // same fp64 addition) for a feasible task and the 0.0 of clear_decisions (:131) otherwise.  The wake-up times use the image's
    // own f32[T] array, which the arrival slots do not cover.
    __device__ __forceinline__ double* ts_x() const { return (double*)(S.base + SimT::lds_image_bytes(L)); }
    static constexpr uint32_t LDS_BYTES = SimT::lds_image_bytes(Lay{CA, CT}) + 8u * CT + 512u;   // + the removal path's dummy slots

    __device__ __forceinline__ double* slots() const { return (double*)(S.base + SLOTS_OFF); }
    __device__ __forceinline__ static bool in_task(int c, int lane) { return c * 64 + lane < CT; }
    __device__ __forceinline__ static int tidx(int c, int lane) { return in_task(c, lane) ? c * 64 + lane : 0; }   // clamped
    __device__ __forceinline__ void init(int lane) {
        inA = lane < CA; isD = lane == DL;
        la = inA ? lane : 0;
    }
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
    __device__ __forceinline__ static uint32_t rl(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

    // ------------------------------------------------------------------------------ registers <-> LDS image / HBM record
    __device__ __forceinline__ void load_consts(R& r, const typename SimT::XY& xy, int lane) const {
#pragma unroll
        for (int c = 0; c < NTC; c++) {
            const bool dep = (c == DC) && isD;
            r.tx[c] = dep ? ((const Hdr*)S.base)->depot_x : xy.x[c];
            r.ty[c] = dep ? ((const Hdr*)S.base)->depot_y : xy.y[c];
            r.durf[c] = dep ? 0.f : (float)S.tdur()[tidx(c, lane)];
        }
    }
    // general code -> fast path: registers from the image, then the arrival slots from the record into LDS (over the image)
    __device__ __forceinline__ void reload(R& r, int lane) const {
        r.ax = S.ax()[la]; r.ay = S.ay()[la]; r.arr = S.arr()[la]; r.nd = S.nd()[la]; r.td = S.tdist()[la];
        r.cur = S.cur()[la]; r.ai = S.ainfo()[la];
        const int K = r.cur < 0 ? 0 : r.cur;
        r.cts = S.ts()[K]; r.cend = r.cts + S.tdur()[K]; r.cfeas = S.tinfo()[K] & T_FEAS;
        int ninf = 0;
#pragma unroll
        for (int c = 0; c < NTC; c++) {
            const int t = tidx(c, lane);
            r.ti[c] = S.tinfo()[t]; r.ids[c] = S.mids()[t];
            if (in_task(c, lane)) { ts_x()[t] = S.ts()[t]; S.wake()[t] = -__builtin_inff(); }   // every chunk is due at the next new event
            ninf += __popcll(__ballot(!(r.ti[c] & T_FEAS)) & tmask(c));
        }
        n_infeas = ninf;
        touched = (1u << NTC) - 1u;
        WSYNC();
        const u32x4* src = (const u32x4*)(rec + L.marr());
        u32x4* dst = (u32x4*)slots();
        for (int i = lane; i < (int)(8 * M * CT / 16); i += WAVE) dst[i] = __builtin_nontemporal_load(src + i);
        WSYNC();
    }
    // fast path -> general code: park the arrival slots in the record, rebuild the image sections they covered from the registers
    __device__ __forceinline__ void flush(const R& r, int lane) const {
        WSYNC();
        {
            const uint4* src = (const uint4*)slots();
            uint4* dst = (uint4*)(rec + L.marr());
            for (int i = lane; i < (int)(8 * M * CT / 16); i += WAVE) dst[i] = src[i];
        }
        WSYNC();
        if (inA) {
            S.ax()[la] = r.ax; S.ay()[la] = r.ay; S.arr()[la] = r.arr; S.nd()[la] = r.nd; S.tdist()[la] = r.td;
            S.cur()[la] = r.cur; S.ainfo()[la] = r.ai;
        }
#pragma unroll
        for (int c = 0; c < NTC; c++) if (in_task(c, lane)) {
            const int t = c * 64 + lane;
            const double ts_ = ts_x()[t];
            S.tinfo()[t] = r.ti[c]; S.ts()[t] = ts_; S.tf()[t] = (r.ti[c] & T_FEAS) ? ts_ + S.tdur()[t] : 0.0; S.mids()[t] = r.ids[c];
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
        uint32_t info = r.ti[C];
        const bool feas0 = info & T_FEAS;
        const int req = info & 0xFF, n = (info >> 16) & 0xFF;                    // :250
        double av[M];
#pragma unroll
        for (int j = 0; j < M; j++) av[j] = slots()[j * CT + t];                 // :251 (unused slots hold NaN)
        const double dur = S.tdur()[t];                                          // (the image's duration section is never covered)
        const double tfin = feas0 ? ts_x()[t] + dur : 0.0;                       // time_finish (see R)
        const int status = req - n;                                              // :252
        double mx = av[0], mn = av[0];
#pragma unroll
        for (int j = 1; j < M; j++) { mx = nanmax2(mx, av[j]); mn = nanmin2(mn, av[j]); }
        const bool le0 = status <= 0;                                            // :254
        const bool ok = le0 && (mx - mn <= mwt);                                 // :255
        const double thr = mx - mwt;                                             // :262
        const bool any_drop = inT && !feas0 && (le0 ? (!ok && mn <= thr) : (now - mn >= mwt));
        const bool becomes = inT && !feas0 && ok;                                // :256-258
        const double ntf = mx + dur;                                             // time_finish if it becomes feasible now
        if (becomes) ts_x()[t] = mx;                                             // time_start :256
        int nn = n;
        const uint64_t dmask = __ballot(any_drop);
        if (dmask) {
            uint64_t gone = 0ull;
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
                double* const dump = ts_x() + CT + lane;                         // 64 doubles behind the time_start array
#pragma unroll
                for (int j = 0; j < M; j++) row0[j * CT] = __builtin_nan("");
#pragma unroll
                for (int j = 0; j < M; j++) {
                    const bool kp = (keep >> j) & 1u, lv = (drop >> j) & 1u;
                    const int kj = __popc(keep & ((1u << j) - 1u));
                    const uint32_t id = (j < 4 ? (idl >> (8 * j)) : idh) & 0xFFu;
                    nids |= kp ? (id << (8 * kj)) : 0u;
                    gone |= lv ? (1ull << id) : 0ull;
                    *(kp ? row0 + kj * CT : dump) = av[j];
                }
                r.ids[C] = (uint64_t)nids;
                S.tnab()[t] += (uint32_t)__popc(drop);                           // abandoned_agent.append :265/:271 (section not covered)
                nn = __popc(keep);
            }
            uint64_t todo = dmask;
            do {
                const int b = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1ull;
                const uint64_t g = rl(gone, b);
                const int tk_ = C * 64 + b;
                if ((g >> lane) & 1ull) {                                        // this lane's agent was dropped by task tk_
                    const uint32_t nth_ = r.ai >> 16;
                    r.ai += 1u << 16;
                    if (nth_ < (uint32_t)AB_CAP) S.ablog()[la * AB_CAP + nth_] = (uint16_t)tk_;
                    else { const uint32_t ci = (uint32_t)(la * CT + tk_); atomicAdd((uint32_t*)S.abcnt() + (ci >> 1), 1u << (16 * (ci & 1))); }
                    if (r.cur == tk_) r.ai &= ~A_MEMBER;
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
            if (r.cur == C * 64 + b) { r.cfeas = true; r.cts = ts_k; r.cend = tf_k; }
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
            if ((r.ai & A_INDEPOT) && now >= r.arr) r.ai |= A_RETURNED;
        }
    }

    // ------------------------------------------------------------------------------ agent_update, env/task_env.py:207-243
    __device__ __forceinline__ void agent_update(R& r, double now, double mwt) const {
        const int c = r.cur;
        const bool member = r.cfeas && (r.ai & A_MEMBER);                        // :229-230
        const double ndv = (c == -1) ? __builtin_nan("") : (member ? r.cend : r.arr + mwt);   // :226,:231,:235,:238
        const uint32_t as = member ? ((r.ai & A_ASSIGNED) | ((now >= r.cts) ? A_ASSIGNED : 0u)) : 0u;   // :232-240
        r.nd = (c != -2) ? ndv : r.nd;                                           // :209
        r.ai = (c >= 0) ? ((r.ai & ~A_ASSIGNED) | as) : r.ai;                    // depot leaves `assigned` untouched (Q6)
    }

    // ------------------------------------------------------------------------------ observation, worker.py:57-68
    struct BM { uint64_t w[NTC]; };
    // ag / tk / mk: the env's rows in the three output tensors (wave-uniform)
    __device__ __forceinline__ BM observe(const R& r, double now, int leader, int lane, float* __restrict__ ag, float* __restrict__ tk,
                                          uint8_t* __restrict__ mk) const {
        float* agrow = ag + 6 * la;
        const double lx = rl(r.ax, leader), ly = rl(r.ay, leader);
        if constexpr (OBS) {
            const bool on = r.cur >= 0;                                          // :168
            const double x = r.arr - now, w = now - r.arr, rem = r.cend - now;
            const double travel = (on && x > 0.) ? x : 0.;                       // :169
            const double waiting = (on && now <= r.cts && w > 0.) ? w : 0.;      // :170
            const double remaining = (on && now >= r.cts && rem > 0.) ? rem : 0.;   // :171
            const float f0 = (float)travel, f1 = (float)remaining, f2 = (float)waiting;
            const float f3 = (float)(lx - r.ax), f4 = (float)(ly - r.ay), f5 = (r.ai & A_ASSIGNED) ? 1.f : 0.f;
            if (inA) { agrow[0] = f0; agrow[1] = f1; agrow[2] = f2; agrow[3] = f3; agrow[4] = f4; agrow[5] = f5; }   // :176-177
        }
        BM bm;
        bool unf[NTC];
#pragma unroll
        for (int c = 0; c < NTC; c++) {
            const uint32_t info = (c == DC && isD) ? 0u : r.ti[c];
            unf[c] = !(info & T_FEAS) && (int)(int8_t)((info >> 8) & 0xFF) > 0;  // :199
            bm.w[c] = __ballot(unf[c]) & tmask(c);
        }
        if constexpr (OBS) {
            uint64_t any = 0ull;
#pragma unroll
            for (int c = 0; c < NTC; c++) any |= bm.w[c];
#pragma unroll
            for (int c = 0; c < NTC; c++) {
                const bool dep = (c == DC) && isD;
                const uint32_t info = dep ? 0u : r.ti[c];
                const bool zero = dep ? (any == 0ull) : unf[c];                  // :193; depot byte False iff every task is masked
                const uint8_t mv = zero ? 0 : 1;
                const float g0 = (float)(int)(int8_t)((info >> 8) & 0xFF), g1 = (float)(info & 0xFF), g2 = r.durf[c];
                const float g3 = (float)(r.tx[c] - lx), g4 = (float)(r.ty[c] - ly);   // :185-188
                if (in_task(c, lane) || dep) {
                    // (one per-lane base for every chunk, the chunk offset is an instruction immediate; the depot lane writes row 0)
                    uint8_t* mp = dep ? mk : (mk + lane) + (c * 64 + 1);
                    float* row = dep ? tk : (tk + 5 * lane) + 5 * (c * 64 + 1);
                    *mp = mv;
                    row[0] = g0; row[1] = g1; row[2] = g2; row[3] = g3; row[4] = g4;
                }
            }
        }
        return bm;
    }

    __device__ __forceinline__ int pick_leader(const R& r, const HdrRegs& h, uint64_t k1, uint64_t& gm) const {
        gm = __ballot((int)((r.ai >> 8) & 0xFFu) == h.cur_group) & AM;
        const int glen = __popcll(gm);
        if (glen == 0) return -1;
        return nth(gm, below((uint32_t)(k1 >> 32), glen));
    }

    // ------------------------------------------------------------------------------ one decision (see Fast::decide / apply)
    __device__ __forceinline__ int decide(R& r, HdrRegs& h, const KP& P, int lane, uint64_t k1, float* ag, float* tk, uint8_t* mk) const {
        uint64_t gm;
        const int leader = pick_leader(r, h, k1, gm);
        if (leader < 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; return 0; }    // unreachable: groups are never empty
        const double now = h.now;
        const BM bm = observe(r, now, leader, lane, ag, tk, mk);
        // uniform-random valid action (protocol slot 1): valid = ascending unmasked action ids
        int nv = 0;
#pragma unroll
        for (int c = 0; c < NTC; c++) nv += __popcll(bm.w[c]);
        int kc = -1, tl = DL;                                                    // chunk / lane of the target (depot: last chunk, lane 63)
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
        // the target's registers (one wave-uniform branch per chunk)
        double tx_ = 0., ty_ = 0., dur_k = 0.;
        uint32_t kinfo = 0; uint64_t ids = 0ull;
        const int rc = kc < 0 ? DC : kc;
#pragma unroll
        for (int c = 0; c < NTC; c++) if (c == rc) {
            tx_ = rl(r.tx[c], tl); ty_ = rl(r.ty[c], tl);
            if (kc >= 0) {
                kinfo = rl(r.ti[c], tl); ids = rl(r.ids[c], tl);
            }
        }
        // TaskEnv.step :326-342
        uint64_t rest = gm & ~(1ull << leader);                                  // :328
        int rlen = __popcll(gm) - 1;
        uint64_t mm = 1ull << leader, mlist = (uint64_t)(uint32_t)leader;
        int nm = 1;
        int mypos = 0;
        if (kc < 0) {                                                            // vacancy = len(group) :327 (Q9)
            mm |= rest; nm += rlen; rlen = 0;
        } else if (rlen != 0) {
            const int vacancy = (int)(int8_t)((kinfo >> 8) & 0xFF);              // :327 (may be stale)
            const int nf = (vacancy > 1) ? ((vacancy - 1 < rlen) ? vacancy - 1 : rlen) : 0;   // :330-331
            uint64_t kk = k1;
            for (int j = 0; j < nf; j++) {                                       // :331 choice without replacement
                if ((j & 1) == 0) kk = mix64(kk + GAMMA);
                const uint32_t rr = (j & 1) ? (uint32_t)kk : (uint32_t)(kk >> 32);
                const int f = nth(rest, below(rr, rlen));
                rest &= ~(1ull << f); rlen--;                                    // :332-333
                mm |= 1ull << f;
                mlist |= (uint64_t)(uint32_t)f << (8 * nm);
                mypos = lane == f ? nm : mypos;
                nm++;
            }
        }
        // agent_step :300-324 on ALL lanes (fp64 VALU work with fewer than 16 active lanes is 4x slower on gfx950)
        double d = dist2(r.ax, r.ay, tx_, ty_);
        double arrv = now + over_velocity(d);                                    // :315,:318
        asm volatile("" : "+v"(d), "+v"(arrv));
        const bool mem = (mm >> lane) & 1ull;
        int n = 0, slot = 0;
        if (kc >= 0) {
            // :321-322 members.append unless already listed (Q4: a re-joining agent keeps its slot, its arrival is overwritten)
            n = (kinfo >> 16) & 0xFF;
            slot = n + mypos;
            dur_k = S.tdur()[kc * 64 + tl];
            uint64_t listed = 0ull;                                              // the task's members as an agent bitmask (wave-uniform)
            for (int j = 0; j < n; j++) listed |= 1ull << ((ids >> (8 * j)) & 0xFF);
            if (listed & mm) {
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
            r.cur = k;                                                           // :314
            r.ai = (r.ai & ~(A_GRP | A_MEMBER)) | (kc < 0 ? A_INDEPOT : A_MEMBER);
            if (kc >= 0) {
                slots()[slot * CT + k] = arrv;
                // the new current task as agent_update / the observation read it: a task the device policy can pick is not feasible
                // yet, and time_start / time_finish of such a task are still the 0.0 of clear_decisions (:131, set at :256-257)
                r.cfeas = false; r.cts = 0.0; r.cend = 0.0 + dur_k;
            }
        }
        // A QUIET join: the task still lacks members after it (status = requirement - len(members) > 0) and the previous
        // task_update call -- at this same `now` -- left its lane chunk at a fixed point.  task_update (:245-281) then changes
        // nothing but this task's status: not enough members -> not feasible (:254); the joining members have not waited
        // (arrival >= now, :269); the members already listed were evaluated at this `now` by the previous call.  So the chunk
        // visit is skipped: the task's lane takes the new status itself and pulls its wake-up time forward to the joiners'
        // arrival + max_waiting_time (a group is co-located, so they all arrive when the leader does; a re-joining member may
        // make the true time later: the bound stays conservative).
        const int status_k = (int)(kinfo & 0xFFu) - n;
        const bool quiet = kc >= 0 && status_k > 0 && !((touched >> (kc < 0 ? 0 : kc)) & 1u);
        const float wj = __double2float_rd(rl(arrv, leader) + P.mwt);           // (wave-uniform)
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
        const double ndv = inA ? r.nd : __builtin_nan("");
        const double tmin = wave_nanmin(ndv);                                    // :287
        if (!(tmin == tmin)) return false;
        h.now = tmin;                                                            // worker.py:49
        const bool dec = (ndv == tmin);                                          // :288 exact ==
        const uint64_t dm = __ballot(dec);
        const int first = __ffsll((unsigned long long)dm) - 1;
        bool same = true;
        if (dm & (dm - 1ull)) {
            const double x0 = rl(r.ax, first), y0 = rl(r.ay, first);
            same = __ballot(dec && !(r.ax == x0 && r.ay == y0)) == 0ull;
        }
        if (same) {
            r.ai = (r.ai & ~A_GRP) | (dec ? (1u << 8) : 0u);
            h.n_groups = 1;
        } else {
            bool todo = dec;                                                     // groups in ascending (x, then y) order :293
            uint32_t gid = 0;
            int g = 0;
            for (;;) {
                const double mxv = wave_nanmin(todo ? r.ax : __builtin_nan(""));
                if (!(mxv == mxv)) break;
                const double myv = wave_nanmin((todo && r.ax == mxv) ? r.ay : __builtin_nan(""));
                g++;
                if (todo && r.ax == mxv && r.ay == myv) { gid = (uint32_t)g; todo = false; }
            }
            r.ai = (r.ai & ~A_GRP) | (gid << 8);
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
template <int CA, int CT, bool OBS>
__global__ __launch_bounds__(WAVE, DCM_MC_WAVES) void k_rollout_fast_mc(int A, int T, int PA, int PT, KP P, unsigned char* state, int episodes,
                                                         float* agents_out, float* tasks_out, uint8_t* mask_out,
                                                         int64_t* steps_out, double* summary, uint16_t* ablog,
                                                         const int32_t* sizes, int64_t budget_all, const int64_t* budget_in,
                                                         unsigned char* gscr, double* retlog, int retcap) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    using F = FastM<CA, CT, OBS>;
    using SimT = typename F::SimT;
    SimT S{CA, CT, PA, PT, smem, nullptr};
    constexpr Lay L{CA, CT};
    S.scr = gscr + (size_t)e * L.scratch_bytes();
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    S.gm = (double*)(rec + L.marr());
    typename SimT::XY xy;
    S.template load_record<true, false>(rec, lane, xy);
    S.set_ablog(ablog, e, CA, CT, lane);
    S.set_retlog(retlog, retcap, e, lane);
    if (lane == 0) S.inc_state()[1] = -1;
    WSYNC();
    HdrRegs h = load_hdr(smem);
    F f{S, rec};
    f.init(lane);
    float* ag = nullptr; float* tk = nullptr; uint8_t* mk = nullptr;
    if constexpr (OBS) {
        ag = agents_out + (size_t)e * 6 * CA;
        tk = tasks_out + (size_t)e * 5 * (CT + 1);
        mk = mask_out + (size_t)e * (CT + 1);
    }
    double* row = summary + (size_t)e * 8;
    constexpr int NO_BUDGET = 0x7FFFFFFF;
    int64_t bud = budget_in ? budget_in[e] : budget_all;
    const int left0 = uni((int)((bud < 0 || bud >= NO_BUDGET) ? NO_BUDGET : bud));
    int left = left0;
    uint64_t gd = h.seed + GAMMA * (h.d + 1);
    const uint64_t d0 = h.d;
    typename F::R r;
    f.load_consts(r, xy, lane);
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
