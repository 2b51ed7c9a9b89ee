// fast_rollout.hpp -- register-resident persistent rollout for shapes with A <= 64 and T <= 64
// (the headline BASELINE shape 20A/50T).  Included by dcmrta_env.hip inside its anonymous namespace.
//
// One wavefront per env as everywhere else, but here lane t OWNS task t and lane a OWNS agent a and keeps
// every field of both in VGPRs for the whole launch: task_update, the observation rows, the travel update and
// the next-event reduction are straight-line per-lane code with selects instead of exec-masked branches, and
// there is no LDS traffic and no fence between phases.  Cross-lane access:
//   * wave-uniform picks (leader, target task, member ids)    -> v_readlane
//   * an agent's view of its current task (feasible, ts, tf)  -> ds_bpermute gather
//   * next_decision / group ordering                          -> DPP min reduction + ballots
// Lanes beyond T / A hold inert data (a finished feasible task without members; a returned agent that never
// decides), so no phase needs a range guard; only the global observation stores are predicated.
// The LDS record image is used at the two ends only: loaded into registers at launch / episode start and
// written back for the terminal metrics (numpy pairwise sums, Sim::terminal_metrics) and the final copy-out.
//
// Semantics are those of Sim<> (same reference lines).  STATUS: opt-in (environment variable DCM_FAST_ROLLOUT=1 makes
// dcm_rollout_random dispatch here when the shape fits); parity-green against the oracle (tests/test_gpu_fast.py) but,
// as compiled by hipcc 7.2, not faster than the LDS-resident kernel: 10.6 LDS + 508 VALU + 382 SALU instructions per
// decision vs 62 + 460 + 344 (the select/move traffic of ~40 loop-carried state registers eats the LDS savings);
// 0.695 vs 0.675 ms per 4.9e5-decision launch at 4096 x 20A/50T.  Kept as the starting point for hand-scheduled code.
#pragma once

template <int CA, int CT>
struct Fast {
    using S_t = Sim<CA, CT>;

    struct St {
        // task owned by this lane
        uint32_t tinfo, tnab;
        uint64_t mids;
        double ts, tf, tx, ty, tdur;
        double av[M];  // ordered member arrivals, NaN beyond len(members)
        // agent owned by this lane
        double ax, ay, arr, nd, tdist;
        int32_t cur;
        uint32_t ainfo;
        double ts_c, dur_c;  // time_start / duration of the agent's current task (cache for the observation)
    };

    __device__ __forceinline__ static double rl64(double v, int l) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ static uint64_t rl64u(uint64_t v, int l) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ static uint32_t bperm32(uint32_t v, int src) {
        return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
    }
    __device__ __forceinline__ static double bperm64(double v, int src) {
        const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
        const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
        return __hiloint2double(hi, lo);
    }

    // ------------------------------------------------------------------------------ LDS image <-> registers
    __device__ __forceinline__ static St load(const S_t& S, int lane) {
        const int A_ = S.A(), T_ = S.T();
        const bool tv = lane < T_, avd = lane < A_;
        const int t = tv ? lane : 0, a = avd ? lane : 0;
        St s;
        s.tinfo = tv ? S.tinfo()[t] : (T_FEAS | T_FIN);   // inert: feasible + finished, no members, requirement 0
        s.tnab = tv ? S.tnab()[t] : 0u;
        s.mids = tv ? S.mids()[t] : 0ull;
        s.ts = tv ? S.ts()[t] : 0.0; s.tf = tv ? S.tf()[t] : 0.0;
        s.tx = S.tx()[t]; s.ty = S.ty()[t]; s.tdur = S.tdur()[t];
#pragma unroll
        for (int j = 0; j < M; j++) s.av[j] = tv ? S.marr()[j * T_ + t] : __builtin_nan("");
        s.ax = S.ax()[a]; s.ay = S.ay()[a]; s.arr = avd ? S.arr()[a] : 0.0;
        s.nd = avd ? S.nd()[a] : __builtin_nan("");        // inert: never decides
        s.tdist = avd ? S.tdist()[a] : 0.0;
        s.cur = avd ? S.cur()[a] : -1;
        s.ainfo = avd ? S.ainfo()[a] : A_RETURNED;
        const int K = s.cur < 0 ? 0 : s.cur;
        s.ts_c = bperm64(s.ts, K); s.dur_c = bperm64(s.tdur, K);
        return s;
    }
    __device__ __forceinline__ static void store(const St& s, const S_t& S, int lane) {
        const int A_ = S.A(), T_ = S.T();
        if (lane < T_) {
            S.tinfo()[lane] = s.tinfo; S.tnab()[lane] = s.tnab; S.mids()[lane] = s.mids;
            S.ts()[lane] = s.ts; S.tf()[lane] = s.tf;
#pragma unroll
            for (int j = 0; j < M; j++) S.marr()[j * T_ + lane] = s.av[j];
        }
        if (lane < A_) {
            S.ax()[lane] = s.ax; S.ay()[lane] = s.ay; S.arr()[lane] = s.arr; S.nd()[lane] = s.nd;
            S.tdist()[lane] = s.tdist; S.cur()[lane] = s.cur; S.ainfo()[lane] = s.ainfo;
        }
    }

    // ------------------------------------------------------------------------------ task_update (env/task_env.py:245-281)
    // rare path: some task removes members (:262-265 spread, :268-271 expiry).  Compacts the survivors in order on the
    // owning lane and books the abandonment on the agents' lanes.
    __device__ __forceinline__ static void drop_members(St& s, const S_t& S, uint32_t drop, int lane) {
        uint64_t dl = __ballot(drop != 0);
        while (dl) {                                       // agents' side, one dropping task at a time (wave-uniform)
            const int t = __ffsll((unsigned long long)dl) - 1;
            dl &= dl - 1;
            const uint32_t dm = (uint32_t)__builtin_amdgcn_readlane((int)drop, t);
            const uint64_t ids = rl64u(s.mids, t);
#pragma unroll
            for (int j = 0; j < M; j++) if (dm & (1u << j)) {
                const int id = (int)((ids >> (8 * j)) & 0xFF);
                if (lane == id) {                          // abandoned_agent.append(member) :265/:271
                    const uint32_t nth = s.ainfo >> 16;
                    if (nth < (uint32_t)AB_CAP) S.ablog()[id * AB_CAP + nth] = (uint16_t)t;
                    else { const uint32_t ci = (uint32_t)(id * S.T() + t); atomicAdd((uint32_t*)S.abcnt() + (ci >> 1), 1u << (16 * (ci & 1))); }
                    s.ainfo += 1u << 16;
                    if (s.cur == t) s.ainfo &= ~A_MEMBER;
                }
            }
        }
        if (drop != 0) {                                   // task's side: keep the survivors in order
            const int n = (s.tinfo >> 16) & 0xFF;
            double nav[M];
#pragma unroll
            for (int q = 0; q < M; q++) nav[q] = __builtin_nan("");
            uint64_t nids = 0;
            int k = 0;
#pragma unroll
            for (int j = 0; j < M; j++) {
                const bool kept = (j < n) && !(drop & (1u << j));
#pragma unroll
                for (int q = 0; q < M; q++) nav[q] = (kept && k == q) ? s.av[j] : nav[q];
                nids |= kept ? (((s.mids >> (8 * j)) & 0xFFull) << (8 * k)) : 0ull;
                k += kept ? 1 : 0;
            }
#pragma unroll
            for (int q = 0; q < M; q++) s.av[q] = nav[q];
            s.mids = nids;
            s.tnab += (uint32_t)(n - k);
            s.tinfo = (s.tinfo & ~0x00FF0000u) | ((uint32_t)k << 16);   // status stays as computed before the removal (Q3)
        }
    }

    __device__ __forceinline__ static void task_update(St& s, const S_t& S, double now, double mwt, int lane) {
        const uint32_t info0 = s.tinfo;
        const bool feas0 = info0 & T_FEAS;
        const int req = info0 & 0xFF, n = (info0 >> 16) & 0xFF;               // :250
        const int status = req - n;                                           // :252
        double mx = s.av[0], mn = s.av[0];
#pragma unroll
        for (int j = 1; j < M; j++) { mx = nanmax2(mx, s.av[j]); mn = nanmin2(mn, s.av[j]); }
        const bool le0 = status <= 0;                                         // :254
        const bool ok = !feas0 && le0 && (mx - mn <= mwt);                    // :255
        const double thr = mx - mwt;                                          // :262
        uint32_t spread = 0, q1 = 0;
        bool prev = false;
#pragma unroll
        for (int j = 0; j < M; j++) {
            spread |= (s.av[j] <= thr) ? (1u << j) : 0u;                      // :262-265
            const bool e = !prev && (now - s.av[j] >= mwt);                   // :268-271 with the iterator skip (Q1)
            q1 |= e ? (1u << j) : 0u;
            prev = e;
        }
        const uint32_t drop = feas0 ? 0u : (le0 ? (ok ? 0u : spread) : q1);
        const bool fin = feas0 && (now >= s.tf);                              // :273-274 (old time_finish)
        s.ts = ok ? mx : s.ts;                                                // :256
        s.tf = ok ? mx + s.tdur : s.tf;                                       // :257
        const uint32_t upd = (info0 & (T_FIN | 0x00FF00FFu)) | (ok ? T_FEAS : 0u) | ((uint32_t)(status & 0xFF) << 8);
        s.tinfo = feas0 ? (info0 | (fin ? T_FIN : 0u)) : upd;
        if (__any(drop != 0)) drop_members(s, S, drop, lane);
        const bool all_feasible = __all(s.tinfo & T_FEAS);
        const bool ret = (s.ainfo & A_INDEPOT) && all_feasible && (now >= s.arr);   // depot :277-280
        s.ainfo |= ret ? A_RETURNED : 0u;
    }

    // ------------------------------------------------------------------------------ agent_update (env/task_env.py:207-243)
    __device__ __forceinline__ static void agent_update(St& s, double now, double mwt) {
        const int c = s.cur;
        const int K = c < 0 ? 0 : c;
        const uint32_t info = bperm32(s.tinfo, K);                            // :228
        const double tfK = bperm64(s.tf, K), tsK = bperm64(s.ts, K);
        s.ts_c = tsK;
        const bool member = (info & T_FEAS) && (s.ainfo & A_MEMBER);          // :229-230 (cached membership)
        const double ndv = (c == -1) ? __builtin_nan("") : (member ? tfK : s.arr + mwt);   // :226,:231,:235,:238
        s.nd = (c != -2) ? ndv : s.nd;                                        // :209
        const uint32_t as = member ? ((s.ainfo & A_ASSIGNED) | ((now >= tsK) ? A_ASSIGNED : 0u)) : 0u;   // :232-240
        s.ainfo = (c >= 0) ? ((s.ainfo & ~A_ASSIGNED) | as) : s.ainfo;        // depot keeps `assigned` (Q6)
    }

    // ------------------------------------------------------------------------------ observation (worker.py:57-68)
    // returns the ballot of unfinished (unmasked) tasks
    __device__ __forceinline__ static uint64_t observe(const St& s, const S_t& S, double now, int lane, int leader,
                                                       float* __restrict__ ag, float* __restrict__ tk,
                                                       uint8_t* __restrict__ mask) {
        const int A_ = S.A(), T_ = S.T();
        const double lx = rl64(s.ax, leader), ly = rl64(s.ay, leader);
        const int status = (int)(int8_t)((s.tinfo >> 8) & 0xFF);
        const bool unfinished = !(s.tinfo & T_FEAS) && status > 0;            // env/task_env.py:199
        const uint64_t um = __ballot(unfinished);
        if (ag && lane < A_) {                                                // get_current_agent_status :165-180
            const bool on = s.cur >= 0;                                       // :168
            const double x = s.arr - now, w = now - s.arr, r = s.ts_c + s.dur_c - now;
            const double travel = (on && x > 0.) ? x : 0.;                    // :169
            const double waiting = (on && now <= s.ts_c && w > 0.) ? w : 0.;  // :170
            const double remaining = (on && now >= s.ts_c && r > 0.) ? r : 0.;   // :171
            float* row = ag + 6 * lane;
            row[0] = (float)travel; row[1] = (float)remaining; row[2] = (float)waiting;
            row[3] = (float)(lx - s.ax); row[4] = (float)(ly - s.ay);
            row[5] = (s.ainfo & A_ASSIGNED) ? 1.f : 0.f;
        }
        if (lane < T_) {                                                      // get_current_task_status :182-190, mask :192-200
            if (mask) mask[lane + 1] = unfinished ? 0 : 1;
            if (tk) {
                float* row = tk + 5 * (lane + 1);
                row[0] = (float)status; row[1] = (float)(s.tinfo & 0xFF); row[2] = (float)s.tdur;
                row[3] = (float)(s.tx - lx); row[4] = (float)(s.ty - ly);
            }
        }
        if (lane == 0) {
            if (mask) mask[0] = um ? 1 : 0;                                   // worker.py:58-61
            if (tk) {
                const Hdr* q = (const Hdr*)S.base;
                tk[0] = 0.f; tk[1] = 0.f; tk[2] = 0.f; tk[3] = (float)(q->depot_x - lx); tk[4] = (float)(q->depot_y - ly);
            }
        }
        return um;
    }

    // ------------------------------------------------------------------------------ TaskEnv.step (env/task_env.py:326-342)
    __device__ __forceinline__ static int apply(St& s, const S_t& S, Hdr& h, int lane, int leader, uint64_t gm, int action,
                                                uint64_t k1) {
        const int T_ = S.T();
        if (action < 0 || action > T_) { h.flags |= DCM_FLAG_BAD_ACTION | DCM_FLAG_DONE; return 0; }
        uint64_t rest = gm & ~(1ull << leader);                               // :328
        int rlen = __popcll(rest);
        uint64_t mm = 1ull << leader, mlist = (uint64_t)(uint32_t)leader;
        int nm = 1;
        double tx_, ty_, durk = 0.0;
        if (action == 0) {                                                    // vacancy = len(group): everybody leaves (Q9)
            mm |= rest; nm += rlen; rlen = 0;
            const Hdr* q = (const Hdr*)S.base;
            tx_ = q->depot_x; ty_ = q->depot_y;
        } else {
            const int k = action - 1;
            const uint32_t infok = (uint32_t)__builtin_amdgcn_readlane((int)s.tinfo, k);
            const int vacancy = (int)(int8_t)((infok >> 8) & 0xFF);           // :327 (may be stale)
            const int nf = (vacancy > 1) ? ((vacancy - 1 < rlen) ? vacancy - 1 : rlen) : 0;   // :330-331
            if (nf > M - 1) { h.flags |= DCM_FLAG_OVERFLOW | DCM_FLAG_DONE; return 0; }
            uint64_t kk = k1;
            for (int j = 0; j < nf; j++) {                                    // :331 choice without replacement
                if ((j & 1) == 0) kk = mix64(kk + GAMMA);
                const uint32_t r = (j & 1) ? (uint32_t)kk : (uint32_t)(kk >> 32);
                const int f = nth_set_bit(rest, below(r, rlen), lane);
                rest &= ~(1ull << f); rlen--;                                 // :332-333
                mm |= 1ull << f;
                mlist |= (uint64_t)(uint32_t)f << (8 * nm);
                nm++;
            }
            tx_ = rl64(s.tx, k); ty_ = rl64(s.ty, k); durk = rl64(s.tdur, k);
        }
        // agent_step (:300-324) for every member lane; the other lanes compute and discard
        const bool is_m = (mm >> lane) & 1ull;
        const double d = dist2(s.ax, s.ay, tx_, ty_);
        const double arr_new = h.now + over_velocity(d);                             // :315,:318
        s.tdist = is_m ? s.tdist + d : s.tdist;                               // :317
        s.arr = is_m ? arr_new : s.arr;
        s.ax = is_m ? tx_ : s.ax; s.ay = is_m ? ty_ : s.ay;                   // :320
        s.cur = is_m ? action - 1 : s.cur;                                    // :314
        const uint32_t ai = (s.ainfo & ~(A_GRP | A_MEMBER)) | ((action == 0) ? A_INDEPOT : A_MEMBER);   // :321-322
        s.ainfo = is_m ? ai : s.ainfo;
        s.dur_c = is_m ? durk : s.dur_c;
        if (action > 0) {
            // members.append unless already listed; a re-joining agent keeps its slot with the new arrival (Q4)
            const int k = action - 1;
            const uint32_t infok = (uint32_t)__builtin_amdgcn_readlane((int)s.tinfo, k);
            uint64_t ids = rl64u(s.mids, k);
            int n = (infok >> 16) & 0xFF;
            const bool me = lane == k;
            for (int j = 0; j < nm; j++) {
                const int m = (int)((mlist >> (8 * j)) & 0xFF);
                const uint64_t x = ids ^ (0x0101010101010101ull * (uint64_t)(uint32_t)m);
                uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
                z &= (n >= 8) ? ~0ull : ((1ull << (8 * n)) - 1ull);
                int pos;
                if (z) pos = (__ffsll((unsigned long long)z) - 1) >> 3;
                else {
                    if (n >= M) { h.flags |= DCM_FLAG_OVERFLOW | DCM_FLAG_DONE; return 0; }
                    pos = n++;
                    ids |= (uint64_t)(uint32_t)m << (8 * pos);
                }
                const double avm = rl64(arr_new, m);
                if (pos == 0) s.av[0] = me ? avm : s.av[0];
                else if (pos == 1) s.av[1] = me ? avm : s.av[1];
                else if (pos == 2) s.av[2] = me ? avm : s.av[2];
                else if (pos == 3) s.av[3] = me ? avm : s.av[3];
                else s.av[4] = me ? avm : s.av[4];
            }
            s.mids = me ? ids : s.mids;
            s.tinfo = me ? ((infok & ~0x00FF0000u) | ((uint32_t)n << 16)) : s.tinfo;
        }
        h.d += 1;
        return rlen;
    }

    // ------------------------------------------------------------------------------ next event (boxes D + first half of A)
    // check_finished (env/task_env.py:366-373), loop test (worker.py:45), next_decision (:283-289), get_unique_group
    // (:291-298).  Returns 1 when the episode is over, else 0 with `now` / the pending groups set and `any` telling
    // whether somebody decides; the caller then runs task_update + agent_update (worker.py:50-51) at its single site.
    __device__ __forceinline__ static int event(St& s, Hdr& h, const KP& P, bool& any) {
        const double tmin = wave_nanmin(s.nd);                                // np.nanmin :287
        any = (tmin == tmin);
        bool finished = false;
        if (!any) {                                                           // :368-370
            h.now = wave_nanmax((s.cur != -2) ? s.arr : 0.0);
            finished = __all(s.ainfo & A_RETURNED) && __all(s.tinfo & T_FIN);
        }
        if (finished) h.flags |= DCM_FLAG_FINISHED;
        if (finished || h.now >= P.max_time) return 1;                        // worker.py:45
        h.n_groups = 0;
        if (any) {
            h.now = tmin;                                                     // worker.py:49
            const bool dec = (s.nd == tmin);                                  // :288
            const uint64_t dm = __ballot(dec);
            const int first = __ffsll((unsigned long long)dm) - 1;
            const double x0 = rl64(s.ax, first), y0 = rl64(s.ay, first);
            const bool same = !dec || (s.ax == x0 && s.ay == y0);
            if (__all(same)) {                                                // one group
                s.ainfo = (s.ainfo & ~A_GRP) | (dec ? (1u << 8) : 0u);
                h.n_groups = 1;
            } else {                                                          // rows of np.unique(axis=0) :293
                bool todo = dec;
                uint32_t gid = 0;
                int g = 0;
                for (;;) {
                    const double mxv = wave_nanmin(todo ? s.ax : __builtin_nan(""));
                    if (!(mxv == mxv)) break;
                    const double myv = wave_nanmin((todo && s.ax == mxv) ? s.ay : __builtin_nan(""));
                    g++;
                    const bool hit = todo && s.ax == mxv && s.ay == myv;
                    gid = hit ? (uint32_t)g : gid;
                    todo = todo && !hit;
                }
                s.ainfo = (s.ainfo & ~A_GRP) | (gid << 8);
                h.n_groups = g;
            }
        }
        return 0;
    }
};

// The loop has ONE task_update/agent_update site (it serves worker.py:74-76 after a decision and worker.py:50-51 after a
// new event), which keeps the register footprint of the fully inlined kernel under 128 VGPRs (4 waves per SIMD).
template <int CA, int CT>
__global__ __launch_bounds__(WAVE, 4) void k_rollout_fast(int A, int T, KP P, unsigned char* state, int episodes,
                                                      float* agents_out, float* tasks_out, uint8_t* mask_out,
                                                      int64_t* steps_out, double* summary, uint16_t* ablog) {
    const int e = blockIdx.x, lane = threadIdx.x;
    Sim<CA, CT> S{A, T, smem};
    using F = Fast<CA, CT>;
    const Lay L = S.L();
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    copy16_in(smem, rec, L.rec_bytes(), lane);
    S.set_ablog(ablog, e, S.A(), S.T(), lane);
    WSYNC();
    Hdr h = load_hdr(smem);
    float* ag = agents_out ? agents_out + (size_t)e * 6 * L.A : nullptr;
    float* tk = tasks_out ? tasks_out + (size_t)e * 5 * (L.T + 1) : nullptr;
    uint8_t* mk = mask_out ? mask_out + (size_t)e * (L.T + 1) : nullptr;
    double* row = summary + (size_t)e * 8;
    int64_t steps = 0;
    int ep = 0;
    const uint32_t ERR = DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER;
    typename F::St s = F::load(S, lane);
    bool deciding = !(h.flags & DCM_FLAG_DONE);   // the record is either at a decision point or finished
    bool any = false;
    bool alive = episodes > 0 && !(h.flags & ERR);
    bool boundary = false;                        // go straight to the event boundary (fresh episode)
    if (alive && !deciding) {                     // finished record: restart from the loaded instance; d keeps running
        S.reset_state(h, lane);
        WSYNC();
        s = F::load(S, lane);
        boundary = true;
    }
    while (alive) {
        bool over = false;
        if (!boundary) {
            int rlen = 0;
            if (deciding) {                                                   // boxes B + C
                const uint64_t gm = __ballot((int)((s.ainfo >> 8) & 0xFFu) == h.cur_group);   // worker.py:53
                const int glen = __popcll(gm);
                if (glen == 0) { h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE; break; }
                const uint64_t k1 = key1(h.seed, h.d);
                const int leader = nth_set_bit(gm, below((uint32_t)(k1 >> 32), glen), lane);   // worker.py:54
                const uint64_t um = F::observe(s, S, h.now, lane, leader, ag, tk, mk);
                const int nv = __popcll(um);
                const int action = nv ? nth_set_bit(um, below((uint32_t)k1, nv), lane) + 1 : 0;   // uniform-random valid action
                rlen = F::apply(s, S, h, lane, leader, gm, action, k1);
                if (h.flags & DCM_FLAG_DONE) break;
                steps++;
            }
            F::task_update(s, S, h.now, P.mwt, lane);                         // worker.py:74 / :50
            F::agent_update(s, h.now, P.mwt);                                 // worker.py:76 / :51
            if (deciding) {
                if (rlen > 0) continue;                                       // worker.py:53 same group, next leader
                if (h.cur_group < h.n_groups) { h.cur_group++; continue; }    // worker.py:52 next group
            } else if (any) {                                                 // a new event with deciders is set up
                h.empty_passes = 0; h.cur_group = 1; deciding = true;
                continue;
            } else if (++h.empty_passes > 4) {                                // zero-decider guard
                h.flags |= DCM_FLAG_TRUNCATED;
                over = true;
            }
        }
        // ---- event boundary: worker.py:85 -> :45
        boundary = false;
        deciding = false;
        for (;;) {
            if (!over) over = F::event(s, h, P, any) != 0;
            if (!over) break;
            F::store(s, S, lane);                                             // terminal: worker.py:87,103-108
            S.terminal(h, P, lane, row);
            if (++ep >= episodes) { alive = false; break; }
            S.reset_state(h, lane);                                           // next episode of the same instance
            WSYNC();
            s = F::load(S, lane);
            over = false;
        }
    }
    if (!(h.flags & DCM_FLAG_DONE)) F::store(s, S, lane);   // mid-episode state (error exit) goes back to the record
    if (lane == 0 && steps_out) steps_out[e] = steps;
    WSYNC();
    store_hdr(h, lane);
    WSYNC();
    copy16(rec, smem, L.mut_bytes(), lane);
}
