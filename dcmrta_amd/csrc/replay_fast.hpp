// replay_fast.hpp -- register-resident route replay (included by dcmrta_replay.hip, inside its anonymous namespace).
//
// execute_by_route (env/task_env.py:562-593) with the design of the round-4 RL kernels (rollout_fast*.hpp) applied to the replay:
//
//   * lane a of agent chunk i owns agent i*64 + a: position, arrival_time[-1], next_decision, route cursor + next preset action,
//     route[-1], flags, a CACHE of its current task's time_finish (NaN while that task is not feasible) and the re-arm quantum
//     (next - 1) // batch * period of :221 -- so agent_update (:207-243) is lane-local arithmetic on all agents at once: the
//     literal reference semantics (every agent, every call) for the price the LDS version paid for one agent;
//   * lane t of task chunk c owns LIVE task c*64 + t: status word, time_finish and the member arrival slots (NaN-padded:
//     v_min/v_max_f64 ignore them) -- so task_update (:245-281) is lane-local select code on a whole chunk at once; only the
//     removal of members (rare) runs a wave-uniform loop;
//   * an agent_step reads the deciding agent's action and the joined task's status word with v_readlane and its target's
//     location + member ids in ONE LDS round trip (wave-uniform address); the distance chain runs on all lanes, each from its own
//     position, and lane l commits; results go back through lane-select moves;
//   * LIVE tasks: the tasks that can ever get a member -- all T without dynamic arrivals; with them only tasks 1..vis_cap,
//     because an agent is never sent to a task that is not visible yet (:578-584) and visible <= cap (:567).  At the
//     reference's constants (cap 100) that is 2 lane chunks of a 100A/500T instance; the other 400 tasks stay as they were
//     loaded (not feasible, no members) and only enter np.all(feasible) / np.all(finished) and the terminal metrics.
//
// Template: NAC agent chunks, NTL live task chunks, CMR member slots (bytes of one id word), REACTIVE = dynamic arrivals.
// Sizes are runtime values (A <= 64 NAC, live tasks <= 64 NTL, member_cap <= CMR), so every replay of a small shape runs this
// kernel too and the random parity sweeps exercise it.
//
// Registers hold what every call of task_update / agent_update reads on every lane (54 VGPRs of state for <2,2,5>); what a step
// reads or updates ONCE, by index, sits in LDS (FL below): the routes int32[A][route_cap], the live tasks' location / duration /
// ordered member ids / len(abandoned_agent), the agents' travel_dist (ds_add_f64 by the agent's own lane) and max(arrival_time).  About 9 KB per env at 100A/500T; after the loop the head of the same bytes holds the terminal metrics'
// serial-sum inputs (f64[T] + f64[A]).  time_start is written once per task to the handle's HBM scratch and read back at the end.

__device__ __forceinline__ int rli(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// LDS layout of k_replay_fast<NAC, NTL, ...>: [routes | txy | dur] are dead after the event loop and then hold tw f64[T], awl f64[A]
struct FL {
    int A, T, cap, NA, NT;                                                   // NA = 64 NAC agent lanes, NT = 64 NTL live-task lanes
    __host__ __device__ uint32_t routes() const { return 0; }                                    // i32[A][cap]
    __host__ __device__ uint32_t txy() const { return align16((uint32_t)(4 * A * cap)); }       // f64[NT][2]  task location
    __host__ __device__ uint32_t dur() const { return txy() + 16u * (uint32_t)NT; }             // f64[NT]
    __host__ __device__ uint32_t head_end() const { return dur() + 8u * (uint32_t)NT; }
    __host__ __device__ uint32_t term_end() const { return align16((uint32_t)(8 * T + 8 * A)); }   // tw f64[T], awl f64[A]
    __host__ __device__ uint32_t ids() const { return head_end() > term_end() ? head_end() : term_end(); }   // u64[NT]
    __host__ __device__ uint32_t nab() const { return ids() + 8u * (uint32_t)NT; }              // u32[NT]
    __host__ __device__ uint32_t td() const { return nab() + 4u * (uint32_t)NT; }               // f64[NA] travel_dist
    __host__ __device__ uint32_t amx() const { return td() + 8u * (uint32_t)NA; }               // f64[NA] max(arrival_time)
    __host__ __device__ uint32_t bytes() const { return align16(amx() + 8u * (uint32_t)NA); }
};
__host__ __device__ inline uint32_t replay_fast_lds_bytes(int A, int T, int route_cap, int NAC = 2, int NTL = 2) {
    return FL{A, T, route_cap, 64 * NAC, 64 * NTL}.bytes();
}

#ifndef DCM_REPLAY_WAVES
#define DCM_REPLAY_WAVES 1      // minimum waves per SIMD asked of the compiler for k_replay_fast (133 VGPRs = three on its own since the step has no
                                // exit in its middle -- round 6; four fit without spills, 128 VGPRs, and measure the same: 2.01e9 both ways)
#endif

template <int NAC, int NTL, int CMR, bool REACTIVE>
__global__ __launch_bounds__(WAVE, DCM_REPLAY_WAVES) void k_replay_fast(int A, int T, int TL, int PA, int PT, int MR, RP P, const unsigned char* state,
                                                     const int32_t* routes, const int32_t* route_len, int route_cap,
                                                     double* summary, int64_t* steps_out, uint32_t* flags_out,
                                                     uint8_t* finished, double* time_start, double* time_finish,
                                                     double* task_wait, int32_t* n_members, double* agent_wait,
                                                     double* travel_dist, uint8_t* returned, unsigned char* gscr) {
    static_assert(CMR <= 8, "member ids: one byte each of one 64-bit word");
    const int e = blockIdx.x, lane = threadIdx.x;
    const Lay EL{PA, PT};                                      // layout dims of the handle's records (>= the batch dims)
    const unsigned char* rec = state + (size_t)e * EL.rec_bytes();
    uint16_t* const gab = (uint16_t*)(gscr + (size_t)e * EL.scratch_bytes() + EL.s_absort());   // abandonment log u16[A][AB_CAP]
    double* const gts = (double*)(gscr + (size_t)e * EL.scratch_bytes() + EL.s_tw());            // time_start f64[T] (written once per task)
    // (reads of what the wave stored earlier go past the CU's vector L1)
    auto gload = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const Hdr* gh = (const Hdr*)rec;
    const double depot_x = uni(gh->depot_x), depot_y = uni(gh->depot_y);
    const FL F{A, T, route_cap, WAVE * NAC, WAVE * NTL};
    int32_t* const lroute = (int32_t*)(smem + F.routes());
    double* const ltxy = (double*)(smem + F.txy());
    double* const ldur = (double*)(smem + F.dur());
    uint64_t* const lids = (uint64_t*)(smem + F.ids());
    uint32_t* const lnab = (uint32_t*)(smem + F.nab());
    double* const ltd = (double*)(smem + F.td());
    double* const lamx = (double*)(smem + F.amx());
    const double NaN = __builtin_nan("");
    {
        const int32_t* my_routes = routes + (size_t)e * A * route_cap;
#pragma nounroll
        for (int i = lane; i < A * route_cap; i += WAVE) lroute[i] = my_routes[i];
    }
    // ---- clear_decisions (env/task_env.py:129-140) from the loaded instance
    uint32_t ti[NTL];
    double tf[NTL], sl[NTL][CMR];
    {
        const double *gtx = (const double*)(rec + EL.tx()), *gty = (const double*)(rec + EL.ty()), *gtd = (const double*)(rec + EL.tdur());
        const uint32_t* gti = (const uint32_t*)(rec + EL.tinfo());
#pragma unroll
        for (int c = 0; c < NTL; c++) {
            const int t = c * WAVE + lane, tt = t < TL ? t : 0;
            const uint32_t req = t < TL ? (gti[tt] & 0xFFu) : 1u;            // lanes beyond the live tasks: an inert task
            ti[c] = req | (req << 8);
            tf[c] = 0.0;
            ltxy[2 * t] = gtx[tt]; ltxy[2 * t + 1] = gty[tt]; ldur[t] = gtd[tt];
            lids[t] = 0; lnab[t] = 0;
#pragma unroll
            for (int j = 0; j < CMR; j++) sl[c][j] = NaN;
        }
    }
    double ax[NAC], ay[NAC], arr[NAC], nd[NAC], ctf[NAC], rq[NAC];           // rq: (next preset action - 1) // batch * period (:221)
    int nxt[NAC], cur[NAC], hl[NAC];                                         // hl = route cursor | len(pre_set_route) << 16 (-1 = None)
    uint32_t ai[NAC];
#pragma unroll
    for (int i = 0; i < NAC; i++) {
        const int a = i * WAVE + lane, aa = a < A ? a : 0;
        const int len = a < A ? route_len[(size_t)e * A + aa] : -1;          // pre_set_route :595-599
        hl[i] = len << 16;
        nxt[i] = len > 0 ? routes[((size_t)e * A + aa) * route_cap] : 0;
        rq[i] = REACTIVE ? Rep::rearm_quantum(nxt[i], P.vis_batch, P.vis_period) : 0.0;
        ax[i] = depot_x; ay[i] = depot_y; arr[i] = 0.0; ctf[i] = NaN;
        nd[i] = a < A ? 0.0 : NaN;
        cur[i] = -2; ai[i] = 0;
        ltd[a] = 0.0; lamx[a] = 0.0;
    }
    double now = 0.0;
    uint32_t flags = 0;
    bool finished_flag = false, redo = true;
    int visible = 0, guard = 0, n_infeas = T, steps = 0;
    const double mwt = P.mwt;                                                // :564
    const int step_cap = 64 * (A + T) + 4096;                                // guard shared with the oracle (not in the reference)
    WSYNC();

    // ---- task_update (:245-281) of the live chunks `cm` (bit c), lane-local; returns through redo / n_infeas
    bool tu_changed = false;          // the last task_update call made a task feasible or removed members (what other agents' next decisions read)
    auto task_update = [&](uint32_t cm) {
        bool touched = false;
        tu_changed = false;
#pragma unroll
        for (int c = 0; c < NTL; c++) if ((cm >> c) & 1u) {
            const uint32_t info = ti[c];
            const bool feas = info & T_FEAS, fin = info & T_FIN;
            const int req = info & 0xFF, n = (info >> 16) & 0xFF, status = req - n;     // :250-252
            double mx = sl[c][0], mn = sl[c][0];
#pragma unroll
            for (int j = 1; j < CMR; j++) { mx = nanmax2(mx, sl[c][j]); mn = nanmin2(mn, sl[c][j]); }
            const bool full = !feas && status <= 0;                          // :254
            const bool ok = full && (mx - mn <= mwt);                        // :255
            // members leave: the earliest arrival is <= max - mwt (:262) / has waited max_waiting_time (:269: now - v is the
            // largest for the earliest member, and the first such member of the scan is always removed)
            const bool rm = (full && !ok) || (!feas && status > 0 && (now - mn >= mwt));
            const uint32_t ninfo = feas ? (info | ((!fin && now >= tf[c]) ? T_FIN : 0u))                  // :273-274
                                        : ((info & (T_FIN | 0xFFu)) | ((uint32_t)(status & 0xFF) << 8) | ((uint32_t)n << 16) | (ok ? T_FEAS : 0u));
            ti[c] = rm ? info : ninfo;
            const uint64_t bm = __ballot(ok);
            if (bm) {                                                        // became feasible (once per task) :256-258
                tu_changed = true;
                const double tfin = mx + ldur[c * WAVE + lane];
                tf[c] = ok ? tfin : tf[c];
                if (ok) gts[c * WAVE + lane] = mx;
                // only a task that is already over changes again at this `now` (finished, :273)
                touched = touched || (ok && now >= tfin);
                n_infeas -= __popcll(bm);
                // the agents standing at such a task cache its finish time (see agent_update)
                for (uint64_t m = bm; m; m &= m - 1) {
                    const int L = __ffsll((unsigned long long)m) - 1, k = c * WAVE + L;
                    const double tfk = rl(tf[c], L);
#pragma unroll
                    for (int i = 0; i < NAC; i++) ctf[i] = cur[i] == k ? tfk : ctf[i];
                }
            }
            uint64_t rmm = __ballot(rm);
            if (rmm) {                                                       // :262-265 / :268-271, one task at a time, wave-uniform
                touched = true; tu_changed = true;
                for (; rmm; rmm &= rmm - 1) {
                    const int L = __ffsll((unsigned long long)rmm) - 1, t = c * WAVE + L;
                    const uint32_t inf = (uint32_t)rli((int)ti[c], L);
                    uint64_t idw = uni(lids[t]);
                    const int rq = inf & 0xFF, nn = (inf >> 16) & 0xFF, st = rq - nn;
                    double sv[CMR];
#pragma unroll
                    for (int j = 0; j < CMR; j++) sv[j] = rl(sl[c][j], L);
                    uint32_t drop = 0;
                    if (st <= 0) {
                        double m2 = sv[0];
#pragma unroll
                        for (int j = 1; j < CMR; j++) m2 = (j < nn && sv[j] > m2) ? sv[j] : m2;
                        const double thr = m2 - mwt;                         // :262
#pragma unroll
                        for (int j = 0; j < CMR; j++) if (j < nn && sv[j] <= thr) drop |= 1u << j;
                    } else {
                        bool skip = false;                                   // remove-while-iterating (quirk Q1)
#pragma unroll
                        for (int j = 0; j < CMR; j++) if (j < nn) {
                            if (skip) skip = false;
                            else if (now - sv[j] >= mwt) { drop |= 1u << j; skip = true; }   // :269
                        }
                    }
                    int left = nn;
#pragma unroll
                    for (int j = CMR - 1; j >= 0; j--) if ((drop >> j) & 1u) {
                        const int id = (int)((idw >> (8 * j)) & 0xFFu);
                        // abandoned_agent.append(id) :265/:271 -- the agent's own lane logs it
#pragma unroll
                        for (int i = 0; i < NAC; i++) if (i == (id >> 6)) {
                            const bool me = lane == (id & 63);
                            const uint32_t nth = ai[i] >> 16;
                            if (me && nth < (uint32_t)AB_CAP) gab[id * AB_CAP + (int)nth] = (uint16_t)t;
                            ai[i] = me ? ((ai[i] + (1u << 16)) & ((cur[i] == t) ? ~A_MEMBER : ~0u)) : ai[i];
                        }
                        // close the gap: slots j+1.. move down one
#pragma unroll
                        for (int q = j; q < CMR - 1; q++) sv[q] = sv[q + 1];
                        sv[CMR - 1] = NaN;
                        const uint64_t lowm = (j == 0) ? 0ull : (~0ull >> (64 - 8 * j));
                        idw = (idw & lowm) | ((idw >> 8) & ~lowm);
                        left--;
                    }
                    const bool me = lane == L;
#pragma unroll
                    for (int j = 0; j < CMR; j++) sl[c][j] = me ? sv[j] : sl[c][j];
                    if (lane == 0) { lids[t] = idw; lnab[t] += (uint32_t)(nn - left); }
                    // (the status byte is the one computed BEFORE the removal: quirk Q3)
                    ti[c] = me ? ((inf & (T_FEAS | T_FIN | 0xFFu)) | ((uint32_t)(st & 0xFF) << 8) | ((uint32_t)left << 16)) : ti[c];
                }
                WSYNC();
            }
        }
        redo = __any(touched);
        if (n_infeas == 0) {                                                 // depot :277-280 (uniform; false until the very end)
#pragma unroll
            for (int i = 0; i < NAC; i++) ai[i] |= ((ai[i] & A_INDEPOT) && now >= arr[i]) ? A_RETURNED : 0u;
        }
    };
    // ---- agent_update (:207-243, with the reactive depot branch :213-224) for every agent, lane-local
    // (am: the agent chunks to recompute -- all of them is the literal reference; after an agent_step that changed nothing another
    //  agent's branch reads, the stepping agent's chunk is enough: everybody else would store what it holds)
    constexpr uint32_t ALL_AGENTS = (1u << NAC) - 1u;
    // np.all(feasible[:visible_length]) (:214): changes when the visibility window moves on or a task becomes feasible
    bool allf_vis = false;
    auto recount_visible = [&]() {
        uint64_t inf = 0;
#pragma unroll
        for (int c = 0; c < NTL; c++) inf |= __ballot(c * WAVE + lane < visible && c * WAVE + lane < T && !(ti[c] & T_FEAS));
        allf_vis = inf == 0;
    };
    auto agent_update = [&](uint32_t am) {
        bool terr = false;
#pragma unroll
        for (int i = 0; i < NAC; i++) if ((am >> i) & 1u) {
            const int c = cur[i];
            double v = NaN;                                                                    // :226
            if constexpr (REACTIVE) {
                const int len = hl[i] >> 16, head = hl[i] & 0xFFFF;
                const bool waits = !allf_vis && !(len >= 0 && head >= len);                    // :215-218
                terr = terr || (c == -1 && waits && len < 0);                                  // :220 TypeError in the reference
                const bool rearm = c == -1 && waits && len >= 0;
                double vr = arr[i];                                                            // :221-222 np.max([arrival, quantum, now])
                vr = rq[i] > vr ? rq[i] : vr;
                vr = now > vr ? now : vr;
                ai[i] = rearm ? (ai[i] & ~A_INDEPOT) : ai[i];                                  // :223-224 depot['members'].remove
                v = rearm ? vr : NaN;
            }
            const bool member = (ctf[i] == ctf[i]) && (ai[i] & A_MEMBER);                      // :228-230
            const double vt = member ? ctf[i] : arr[i] + mwt;                                  // :231 / :235,:238
            v = c >= 0 ? vt : v;
            nd[i] = c == -2 ? nd[i] : v;                                                       // :209 (an agent that never moved)
        }
        if constexpr (REACTIVE) { if (__any(terr)) flags |= R_TYPE_ERROR; }
    };

    double tmin;
    auto read_next_decisions = [&]() {
        double lmin = nd[0];
#pragma unroll
        for (int i = 1; i < NAC; i++) lmin = nanmin2(lmin, nd[i]);
        tmin = wave_nanmin(lmin);
    };
    auto latest_arrival = [&]() {                                            // max(x) if x else 0 over the whole arrival lists
        double lmax = lamx[lane];
#pragma unroll
        for (int i = 1; i < NAC; i++) { const double v = lamx[i * WAVE + lane]; lmax = v > lmax ? v : lmax; }
        return wave_nanmax(lmax);
    };
    read_next_decisions();
    double vis_lo = __builtin_inf(), vis_hi = -__builtin_inf();
    constexpr uint32_t ALL_CHUNKS = (1u << NTL) - 1u;
    while (!finished_flag && now < P.cutoff) {                               // :565
        if (REACTIVE && !(now >= vis_lo && now < vis_hi)) {                  // :566-567 (once per visibility window)
            const double q = py_floordiv(now, (double)P.vis_period);
            vis_lo = q * (double)P.vis_period; vis_hi = (q + 1.0) * (double)P.vis_period;
            double v = q * (double)P.vis_batch + (double)P.vis_initial;
            v = v < (double)P.vis_initial ? (double)P.vis_initial : v; v = v > (double)P.vis_cap ? (double)P.vis_cap : v;
            visible = (int)v;
        }
        // next_decision :283-289
        const bool any = (tmin == tmin);
        now = any ? tmin : latest_arrival();                                 // :569
        uint64_t dm[NAC];                                                    // the deciding set (exact ==), fixed before the updates
#pragma unroll
        for (int i = 0; i < NAC; i++) dm[i] = __ballot(any && nd[i] == tmin);
        task_update(ALL_CHUNKS);                                             // :570
        if constexpr (REACTIVE) recount_visible();
        agent_update(ALL_AGENTS);                                            // :571
        if (flags & R_TYPE_ERROR) break;
        if (!any) { if (++guard > 8) { flags |= DCM_FLAG_TRUNCATED; break; } } else guard = 0;
#pragma unroll
        for (int i = 0; i < NAC; i++) {
            uint64_t m = dm[i];
            while (m) {                                                      // :572 for agent in decision_agents
                const int l = __ffsll((unsigned long long)m) - 1, a = i * WAVE + l;
                m &= m - 1;
                // the action: the next entry of the preset route, or a forced depot visit (:573-585) -- evaluated by every lane
                // for its own agent, lane l's result is the one that counts
                const int len_l = hl[i] >> 16, head_l = hl[i] & 0xFFFF;
                const bool pop_l = !(len_l < 0 || head_l >= len_l) && !(REACTIVE && nxt[i] > visible);
                // (the entry behind the cursor, for the pop below: requested now, a whole distance chain before it is used)
                const int h1_l = head_l + 1;
                const int up_l = lroute[(i * WAVE + lane < A ? i * WAVE + lane : 0) * route_cap + (h1_l < route_cap ? h1_l : 0)];
                int action = rli(pop_l ? nxt[i] : 0, l);
                const bool popped = (__ballot(pop_l) >> l) & 1ull;
                // (no exit from the middle of a step -- it would keep a second copy of every lane-owned field alive across the step,
                //  see Fast::decide: the env is flagged, the step runs as a depot visit and the loop ends at the test below the step)
                if (action < 0 || action > T) { flags |= DCM_FLAG_BAD_ACTION; action = 0; }
                const int k = action - 1, kk = k >= 0 ? k : 0, kl = kk & 63, kc = k >= 0 ? (k >> 6) : -1;
                // agent_step :300-324.  One LDS round trip: the task's location and member ids (wave-uniform address).  The distance
                // chain runs on all lanes (each from its own position: fp64 VALU work with fewer than 16 active lanes is 4x slower
                // on gfx950), lane l commits
                const double txk = ltxy[2 * kk], tyk = ltxy[2 * kk + 1];
                uint64_t kids = lids[kk];
                const double tx_ = action ? txk : depot_x, ty_ = action ? tyk : depot_y;
                const double d = dist2(ax[i], ay[i], tx_, ty_);
                const double arrival_l = now + over_velocity(d);             // :315,:318
                const double arrival = rl(arrival_l, l);
                const bool me = lane == l;
                // the task's side: is the agent already listed?  (:321-322)
                uint32_t kinfo = 0;
#pragma unroll
                for (int c = 0; c < NTL; c++) if (c == kc) kinfo = (uint32_t)rli((int)ti[c], kl);
                int n = (kinfo >> 16) & 0xFF, pos = -1;
                bool fresh = false;
                if (action) {
                    kids = uni(kids);
                    // lane j compares member byte j (the list holds an agent at most once)
                    const uint64_t hit = __ballot(lane < n && (int)((kids >> (8 * (lane & 7))) & 0xFFu) == a);
                    pos = hit ? __ffsll((unsigned long long)hit) - 1 : -1;
                    if (pos < 0) {
                        if (n >= MR) flags |= DCM_FLAG_OVERFLOW;
                        else { pos = n++; fresh = true; }
                    }
                    if (fresh) lids[kk] = kids | ((uint64_t)(uint32_t)a << (8 * pos));
                }
                const bool joined = pos >= 0;
                // the task's word as task_update would leave it if the join changes nothing else: len(members), and for a task that
                // is not feasible its status = requirements - len(members) (:250-252)
                const bool feas_k = kinfo & T_FEAS;
                const int status_k = (int)(kinfo & 0xFFu) - n;
                const uint32_t kinfo2 = feas_k ? ((kinfo & ~0x00FF0000u) | ((uint32_t)n << 16))
                                               : ((kinfo & (T_FIN | 0xFFu)) | ((uint32_t)(status_k & 0xFF) << 8) | ((uint32_t)n << 16));
#pragma unroll
                for (int c = 0; c < NTL; c++) if (c == kc) {
                    const bool mk = lane == kl;
                    ti[c] = mk ? kinfo2 : ti[c];
#pragma unroll
                    for (int j = 0; j < CMR; j++) sl[c][j] = (mk && j == pos) ? arrival : sl[c][j];
                }
                // the agent's side: travel_dist += d (:317) and max(arrival_time) by the agent's own lane in LDS (the others add 0 /
                // offer 0: a member released by its task finishing before it arrived re-decides early, so the arrival list is not
                // monotone in replays with surplus visitors; :286 takes the max over the whole list)
                __hip_atomic_fetch_add(&ltd[i * WAVE + lane], me ? d : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_max(&lamx[i * WAVE + lane], me ? arrival_l : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // (arrivals are >= 0)
                arr[i] = me ? arrival_l : arr[i];
                ax[i] = me ? tx_ : ax[i]; ay[i] = me ? ty_ : ay[i];          // :320
                cur[i] = me ? k : cur[i];                                    // :314
                ai[i] = me ? ((ai[i] & ~A_MEMBER) | (action == 0 ? A_INDEPOT : 0u) | (joined ? A_MEMBER : 0u)) : ai[i];
                if (popped) {                                                // :585 pop(0): the cursor moves on, the next entry is staged
                    const int up = h1_l < len_l ? up_l : 0;
                    hl[i] = me ? hl[i] + 1 : hl[i];
                    nxt[i] = me ? up : nxt[i];
                    if constexpr (REACTIVE) { const double q = Rep::rearm_quantum(up, P.vis_batch, P.vis_period); rq[i] = me ? q : rq[i]; }
                }
                if (++steps > step_cap) flags |= DCM_FLAG_TRUNCATED | DCM_FLAG_OVERFLOW;
                // :575/:582/:586.  After a call that changed nothing a second call at the same `now` could change again (redo false)
                // every task but the joined one is at a fixed point of task_update, and so is the joined one unless this join
                // completes its coalition: a member who has just arrived has not waited (:269), a task that still lacks members
                // only gets the status written above (:252-254), a feasible one was checked against this `now` already (:273).
                const bool was_redo = redo;
                const bool completes = action && !feas_k && status_k <= 0;
                task_update(redo ? ALL_CHUNKS : (completes ? (1u << kc) : 0u));
                // the agent now stands at task k: cache its finish time if it is feasible
                if (action) {
                    double tfk = NaN;
#pragma unroll
                    for (int c = 0; c < NTL; c++) if (c == kc) { const bool fk = (uint32_t)rli((int)ti[c], kl) & T_FEAS; const double t2 = rl(tf[c], kl); tfk = fk ? t2 : NaN; }
                    ctf[i] = me ? tfk : ctf[i];
                }
                if constexpr (REACTIVE) { if (tu_changed) recount_visible(); }
                agent_update((was_redo || tu_changed) ? ALL_AGENTS : (1u << i));      // :576/:583/:587
                if (flags & (R_TYPE_ERROR | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_ACTION)) break;
            }
            if (flags & (R_TYPE_ERROR | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_ACTION)) break;
        }
        if (flags & (R_TYPE_ERROR | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_ACTION)) break;
        // check_finished :366-373,:588
        read_next_decisions();
        if (!(tmin == tmin)) {
            now = latest_arrival();
            bool allret = true, allfin = TL >= T;                            // (a task that is never live is never finished)
#pragma unroll
            for (int i = 0; i < NAC; i++) allret = allret && (i * WAVE + lane >= A || (ai[i] & A_RETURNED));
#pragma unroll
            for (int c = 0; c < NTL; c++) allfin = allfin && (c * WAVE + lane >= T || (ti[c] & T_FIN));
            finished_flag = __all(allret) && __all(allfin);
        } else finished_flag = false;
    }
    WSYNC();
    // ---- outputs per task and get_episode_reward: calculate_waiting_time :344-364 (np.sum = pairwise block for n >= 8)
    double* const tw = (double*)smem;                                        // f64[T]
    double* const awl = tw + T;                                              // f64[A]
#pragma nounroll
    for (int t = NTL * WAVE + lane; t < T; t += WAVE) tw[t] = 0.0;           // tasks that were never live (exactly as loaded)
    double mxr[NTL], tsr[NTL];
    int nfin = 0;
#pragma unroll
    for (int c = 0; c < NTL; c++) {
        const int t = c * WAVE + lane;
        const uint32_t info = ti[c];
        const int n = (info >> 16) & 0xFF;
        const bool feas = info & T_FEAS;
        tsr[c] = (feas && t < T) ? gload(&gts[t]) : 0.0;
        double mx = sl[c][0];
#pragma unroll
        for (int j = 1; j < CMR; j++) mx = nanmax2(mx, sl[c][j]);
        mxr[c] = mx;
        double term[CMR];
#pragma unroll
        for (int j = 0; j < CMR; j++) term[j] = feas ? (mx - sl[c][j]) : (now - sl[c][j]);
        double s = 0.0;
        if (CMR == 8 && n == 8) s = ((term[0] + term[1 % CMR]) + (term[2 % CMR] + term[3 % CMR])) + ((term[4 % CMR] + term[5 % CMR]) + (term[6 % CMR] + term[7 % CMR]));
        else {
#pragma unroll
            for (int j = 0; j < CMR; j++) s = j < n ? s + term[j] : s;
        }
        const double w = s + (double)lnab[t] * mwt;
        nfin += __popcll(__ballot(t < T && (info & T_FIN)));
        if (t < T) {
            tw[t] = w;
            const size_t o = (size_t)e * T + t;
            if (finished) finished[o] = (info & T_FIN) ? 1 : 0;
            if (task_wait) task_wait[o] = w;
            if (n_members) n_members[o] = n;
            if (time_finish) time_finish[o] = tf[c];
        }
    }
#pragma nounroll
    for (int t = NTL * WAVE + lane; t < T; t += WAVE) {
        const size_t o = (size_t)e * T + t;
        if (finished) finished[o] = 0;
        if (task_wait) task_wait[o] = 0.0;
        if (n_members) n_members[o] = 0;
        if (time_finish) time_finish[o] = 0.0;
    }
    // :358-364 per agent in the reference's order: tasks ascending, member term first, then +max_waiting_time per entry of the
    // agent in that task's abandoned_agent list (entries from the abandonment log, sorted by task id).  Task-major: a
    // wave-uniform walk over the tasks that list members, each listed agent's lane adds its term.
    double aw[NAC];
    int abp[NAC], abn[NAC];
    bool over = false, any_ab = false;
#pragma unroll
    for (int i = 0; i < NAC; i++) {
        aw[i] = 0.0; abp[i] = 0;
        const uint32_t na = ai[i] >> 16;
        abn[i] = na < (uint32_t)AB_CAP ? (int)na : AB_CAP;
        over = over || na > (uint32_t)AB_CAP;
        any_ab = any_ab || na != 0;
        if (na > 1) {                                                        // sort the agent's log by task id (insertion sort)
            uint16_t* my = gab + (i * WAVE + lane) * AB_CAP;
            for (int q = 1; q < abn[i]; q++) {
                const uint16_t v = my[q];
                int j = q;
                while (j > 0 && my[j - 1] > v) { my[j] = my[j - 1]; j--; }
                my[j] = v;
            }
        }
    }
    const bool has_ab = __any(any_ab);
    if (__any(over)) flags |= DCM_FLAG_WAIT_ORDER;
#pragma unroll
    for (int c = 0; c < NTL; c++) {
        uint64_t tm = __ballot(((ti[c] >> 16) & 0xFF) != 0 || lnab[c * WAVE + lane] != 0);
        for (; tm; tm &= tm - 1) {
            const int L = __ffsll((unsigned long long)tm) - 1, t = c * WAVE + L;
            const uint32_t info = (uint32_t)rli((int)ti[c], L);
            const uint64_t idw = uni(lids[t]);
            const int n = (info >> 16) & 0xFF;
            const bool feas = info & T_FEAS;
            const double mxt = rl(mxr[c], L);
#pragma unroll
            for (int j = 0; j < CMR; j++) if (j < n) {
                const int id = (int)((idw >> (8 * j)) & 0xFFu);
                const double v = rl(sl[c][j], L);
                double term;
                if (feas) term = mxt - v;                                     // :360
                else { const double w = now - v; term = w > 0.0 ? w : 0.0; }  // :362
#pragma unroll
                for (int i = 0; i < NAC; i++) aw[i] = (i == (id >> 6) && lane == (id & 63)) ? aw[i] + term : aw[i];
            }
            if (has_ab) {                                                    // :363-364
#pragma unroll
                for (int i = 0; i < NAC; i++) {
                    const uint16_t* my = gab + (i * WAVE + lane) * AB_CAP;
                    while (abp[i] < abn[i] && my[abp[i]] == (uint16_t)t) { aw[i] += mwt; abp[i]++; }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NAC; i++) {
        const int a = i * WAVE + lane;
        const uint32_t na = ai[i] >> 16;
        aw[i] += (double)(na - (uint32_t)abn[i]) * mwt;
        if (a < A) {
            awl[a] = aw[i];
            const size_t o = (size_t)e * A + a;
            if (agent_wait) agent_wait[o] = aw[i];
            if (travel_dist) travel_dist[o] = ltd[a];
            if (returned) returned[o] = (ai[i] & A_RETURNED) ? 1 : 0;
        }
    }
    WSYNC();
    const double Td = (double)T, Ad = (double)A;
    const double m3 = psum<4>(awl, A) / Ad, m4 = psum<4>(ltd, A), m5 = psum<4>(tw, T) / Td;
    WSYNC();
    // np.nanmean(time_start) (worker.py:105): one serial pairwise sum over T values, through the bytes the waiting sums have left
#pragma unroll
    for (int c = 0; c < NTL; c++) {
        const int t = c * WAVE + lane;
        if (t < T) { tw[t] = tsr[c]; if (time_start) time_start[(size_t)e * T + t] = tsr[c]; }
    }
#pragma nounroll
    for (int t = NTL * WAVE + lane; t < T; t += WAVE) { tw[t] = 0.0; if (time_start) time_start[(size_t)e * T + t] = 0.0; }
    WSYNC();
    const double m2 = psum<4>(tw, T) / Td;
    if (lane == 0) {
        double* row = summary + (size_t)e * 8;
        row[0] = -now; row[1] = (double)nfin; row[2] = (double)nfin / Td; row[3] = now;
        row[4] = m2; row[5] = m3; row[6] = m4; row[7] = m5;
        if (steps_out) steps_out[e] = steps;
        if (flags_out) flags_out[e] = flags | DCM_FLAG_DONE | (finished_flag ? DCM_FLAG_FINISHED : 0u);
    }
}
