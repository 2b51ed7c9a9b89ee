// step_fast.hpp -- the lockstep kernel of the one-chunk layouts for the PLAIN call shape of dcm_step: no injected leader /
// followers, no route log, all five outputs, grouping on (what a policy in the loop calls: worker.py:54-76 once per env).
// Included by dcmrta_env.hip after rollout_fast.hpp.
//
// k_step runs the general Sim<> code on the LDS image: ~500 VALU + ~650 scalar + ~100 LDS instructions per step in a dozen dependent
// LDS round trips, and at a machine-filling batch its rate is (resident workgroups) / (workgroup lifetime), not HBM bandwidth
// (profiles/r03_lockstep: 53 % of the wave time parked on s_waitcnt).  Here the step itself runs on the register-resident
// simulator of rollout_fast.hpp -- record -> LDS (coalesced), lane-owned fields -> registers, one decision with the HOST's
// action, registers -> LDS, dirty sections -> HBM -- whenever that action is one the device policy could have taken (the depot,
// or an unmasked task: every action of a mask-respecting policy).  Anything else (a masked or out-of-range action, an event at
// which nobody can decide, the end of an episode, auto-reset) takes the general code on the same LDS image, exactly as k_step.
// The observation of the next decision is built from the registers when they hold the env, from the LDS image otherwise.
#pragma once
#ifndef DCM_STEP_WAVES
#define DCM_STEP_WAVES 4       // minimum waves per SIMD asked of the compiler for k_step_fast
#endif

// The terminal metrics' scratch (calculate_waiting_time: 3.6 KB at 20A/50T) sits in LDS behind the dummy slots when 16 workgroups
// per CU -- all that the kernel's VGPRs allow -- still fit: the env whose episode ends in a launch is that launch's slowest wave,
// and with the scratch in HBM every write -> WSYNC -> read phase of the metrics is a global-memory round trip.
template <int CA, int CT>
constexpr bool step_scratch_in_lds() { return Lay{CA, CT}.lds_bytes() + 512u <= 10240u; }

template <int CA, int CT, bool RS>
__global__ __launch_bounds__(WAVE, DCM_STEP_WAVES) void k_step_fast(int A, int T, int PA, int PT, KP P, unsigned char* state, const int32_t* actions,
                                                   float* agents_out, float* tasks_out, uint8_t* mask_out, int32_t* leader_out,
                                                   uint8_t* active_out, double* summary, uint16_t* ablog, uint32_t mode,
                                                   const int32_t* sizes, unsigned char* gscr, uint32_t max_episodes, double* retlog,
                                                   int retcap, unsigned char* side, uint32_t side_pitch, uint32_t* pendq, const unsigned char* init) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    using F = Fast<CA, CT, RS, true, true>;
    using SimT = typename F::SimT;
    using AMask = typename SimT::AMask;
    SimT S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = step_scratch_in_lds<CA, CT>() ? smem + SimT::lds_image_bytes(L) + 512u : gscr + (size_t)e * L.scratch_bytes();
    const int BA = S.BA(A), BT = S.BT(T);
    unsigned char* rec = state + (size_t)e * L.rec_bytes();
    double* row = summary + (size_t)e * 8;
    const int act_in = actions[e];                 // requested first: its latency hides behind the record copy
    typename SimT::XY xy;
    S.template load_record<false>(rec, lane, xy);
    S.set_ablog(ablog, e, BA, BT, lane);
    S.set_retlog(retlog, retcap, e, lane);
    if (lane == 0) { *S.dirty() = 0; *S.dirty2() = 0; }
    WSYNC();
    HdrRegs h = load_hdr(smem);
    F f{S, (double*)(smem + SimT::lds_image_bytes(L))};      // (512 bytes of LDS behind the image: the removal path's dummy slots)
    f.init(lane);
    typename F::R r;
    bool regs = false;       // the registers hold the env: agent arrays / member ids / abandonment counts of the LDS image are stale
    constexpr uint32_t ERR = DCM_FLAG_BAD_ACTION | DCM_FLAG_OVERFLOW | DCM_FLAG_BAD_LEADER | DCM_FLAG_BAD_INSTANCE;
    const bool was_active = !(h.flags & DCM_FLAG_DONE);
    PH_DECL;
    // the rest of the launch: registers -> image, header, write-back, the next decision's observation
    auto finish = [&](bool regs) __attribute__((always_inline)) {
    if (was_active) {
        if (regs) f.flush(r);
        WSYNC();
        store_hdr(h, lane);
        WSYNC();
        // write back what the step can have changed (see k_step)
        const uint32_t sd = uni(*S.dirty());
        const uint32_t dm = sd | f.dirty;
        const bool big = gridDim.x >= 8192u;           // (see copy16_nt)
        auto put = [&](uint32_t lo, uint32_t hi) {
            lo &= ~15u; hi = (hi + 15u) & ~15u;
            if (big) copy16_nt(rec + lo, smem + lo, hi - lo, lane); else copy16(rec + lo, smem + lo, hi - lo, lane);
        };
        const uint32_t Tn = (uint32_t)S.PT();
        // When only the register-resident step has touched the record (no general code: sd == 0) and it removed nobody, what it
        // changed of time_start / time_finish / the arrival rows / the member ids belongs to the tasks it names (a join: one
        // task; tasks that became feasible): their 64-byte pieces of those sections go back instead of the 400-byte sections.
        const bool fine = sd == 0 && (f.dirty & SimT::DIRTY_ROWS) != SimT::DIRTY_ROWS && !(f.dirty & SimT::DIRTY_NAB);
        const uint32_t An = (uint32_t)L.A;                 // (the layout's agent count: the pitch of the agent arrays)
        if (fine && regs && (An & 3u) == 0u) {
            // header, then the 16-byte pieces of the (contiguous) agent arrays that hold a changed agent: five f64 arrays -- two agents
            // per piece -- and two 32-bit ones -- four per piece
            put(0, 64);
            const uint32_t n8 = An / 2u, n4 = An / 4u, total = 5u * n8 + 2u * n4;
            for (uint32_t i = lane; i < total; i += WAVE) {
                const bool wide = i < 5u * n8;
                const uint32_t c = wide ? i % n8 : (i - 5u * n8) % n4;
                const uint64_t bits = wide ? (f.achg >> (2u * c)) & 3ull : (f.achg >> (4u * c)) & 15ull;
                if (bits) ((uint4*)(rec + 64))[i] = ((const uint4*)(smem + 64))[i];
            }
        } else put(0, L.tb());                                                        // header + agent arrays
        if (fine) {
            auto piece = [&](uint32_t sec, int t) {                                   // the aligned 64 bytes of a f64 / u64 [T] section that hold task t
                const uint32_t lo = (sec + 8u * (uint32_t)t) & ~63u, end = sec + 8u * Tn;
                put(lo < sec ? sec : lo, lo + 64u < end ? lo + 64u : end);
            };
            for (uint64_t m = f.dt_times; m; m &= m - 1) {
                const int t = __ffsll((unsigned long long)m) - 1;
                piece(L.ts(), t); piece(L.tf(), t);
            }
            for (uint64_t m = f.dt_join; m; m &= m - 1) {
                const int t = __ffsll((unsigned long long)m) - 1;
#pragma unroll
                for (int j = 0; j < M; j++) if (dm & (2u << j)) piece(L.marr() + 8u * Tn * j, t);
                piece(L.mids(), t);
            }
            put(L.tinfo(), L.tnab());                                                 // status words
        } else {
            if (dm & SimT::DIRTY_TIMES) put(L.ts(), L.marr());                        // time_start, time_finish
            if ((dm & SimT::DIRTY_ROWS) == SimT::DIRTY_ROWS) put(L.marr(), L.mids());
            else {
#pragma unroll
                for (int j = 0; j < M; j++) if (dm & (2u << j)) put(L.marr() + 8u * Tn * j, L.marr() + 8u * Tn * (j + 1));
            }
            if (dm & SimT::DIRTY_IDS) put(L.mids(), L.tinfo());
            put(L.tinfo(), (dm & SimT::DIRTY_NAB) ? L.mut_bytes() : L.tnab());        // status words (+ abandonment counts)
        }
    
    }
    // mask + observation of the next decision (worker.py:57-68), fused
    WSYNC();
    float* ag = agents_out + (size_t)e * 6 * BA;
    float* tk = tasks_out + (size_t)e * 5 * (BT + 1);
    uint8_t* mk = mask_out + (size_t)e * (BT + 1);
    int leader = -1;
    if (!(h.flags & DCM_FLAG_DONE)) {
        const uint64_t k1n = key1(h.seed, h.d);
        if (regs) { uint64_t gm; leader = f.pick_leader(r, h, k1n, gm); }
        else { AMask gm; leader = S.pick_leader(h, lane, -1, k1n, gm, false); }
    }
    if (leader >= 0) {
        // rows built in LDS and stored as contiguous runs for grids that fill the machine several times over (see k_step)
        const uint32_t need = 24u * (uint32_t)S.A() + 21u * ((uint32_t)S.T() + 1u) + 16u;
        const bool staged = gridDim.x >= 8192u && L.tinfo() - L.marr() >= need;
        float* sag = (float*)(smem + L.marr());
        float* stk = sag + 6 * S.A();
        uint8_t* smk = (uint8_t*)(stk + 5 * (S.T() + 1));
        float* oag = staged ? sag : ag;
        float* otk = staged ? stk : tk;
        uint8_t* omk = staged ? smk : mk;
        if (regs) f.observe(r, h.now, leader, oag + 6 * f.la, otk + (f.inT ? 5 * (lane + 1) : 0), omk + (f.inT ? lane + 1 : 0));
        else S.observe(h, lane, leader, oag, otk, omk, xy);
        if (staged) {
            WSYNC();
            for (int i = lane; i < 6 * S.A(); i += WAVE) __builtin_nontemporal_store(sag[i], ag + i);
            for (int i = lane; i < 5 * (S.T() + 1); i += WAVE) __builtin_nontemporal_store(stk[i], tk + i);
            for (int i = lane; i <= S.T(); i += WAVE) __builtin_nontemporal_store(smk[i], mk + i);
        }
    } else {
        S.write_inactive_obs(lane, ag, tk, mk);
    }
    if constexpr (RS) S.write_pad_obs(lane, BA, BT, ag, tk, mk);
    if (lane == 0) {
        leader_out[e] = leader;
        active_out[e] = leader >= 0 ? 1 : 0;
    }
    };
    if (was_active) {
        f.load_consts(r);
        f.reload(r);
        const uint64_t k1 = key1(h.seed, h.d);
        // an action the device policy could have taken?  (env/task_env.py:192-200 + worker.py:58-61: an unmasked task; the depot
        // is simulated the same way whether or not it is masked: the whole co-located group returns)
        bool plain = act_in == 0;
        if (act_in >= 1 && act_in <= S.T()) {
            const uint32_t ik = (uint32_t)__builtin_amdgcn_readlane((int)r.ti, act_in - 1);
            plain = !(ik & T_FEAS) && (int)(int8_t)((ik >> 8) & 0xFF) > 0;
        }
        bool general = !plain;                     // the step, or the rest of it, needs the general code
        if (plain) {
            uint64_t gm;
            const int leader = f.pick_leader(r, h, k1, gm);
            if (leader < 0) h.flags |= DCM_FLAG_BAD_LEADER | DCM_FLAG_DONE;     // unreachable: groups are never empty
            else {
                const int rlen = f.apply(r, h, P, lane, k1, gm, leader, act_in);
                h.d += 1;
                regs = true;
                if (rlen == 0) {                                                  // worker.py:53 else same group, next leader
                    if (h.cur_group < h.n_groups) h.cur_group++;                  // worker.py:52 next group
                    else general = !f.next_event(r, h, P, lane);                  // worker.py:85 -> :45
                }
            }
        }
        if (uni((uint32_t)general) != 0u) {
            // The general code -- the end of an episode (+ auto-reset), MAX_TIME, a masked action -- runs to the end of the launch
            // in a region of its own that the common path never rejoins.  Its out-of-line terminal metrics clobber 96 scalar and 68
            // vector registers; with one shared tail behind the call the common path's values lived across it and the kernel sat at
            // its 128-VGPR limit with 40 B of spills, some of them on the common path.  Two tails: 105 VGPRs, no spills of its own.
            // The restart's image (dcm_env::init: the record dcm_reset left, i.e. reset_state + the first event of this instance) is
            // requested now and lands in LDS behind the snapshot: its round trip hides behind the end-of-episode code, and the wave
            // skips reset_state + the first advance() (3700 of its 21 400 clocks at 4096 envs)
            constexpr uint32_t IN16 = Lay{CA, CT}.mut_bytes() / 16, ICH = (IN16 + WAVE - 1) / WAVE;
            u32x4 iv[ICH];
            if (plain) {
                if (init) {
                    const u32x4* q = (const u32x4*)(init + (size_t)e * L.rec_bytes());
#pragma unroll
                    for (uint32_t c = 0; c < ICH; c++) { const uint32_t i = c * WAVE + lane; iv[c] = __builtin_nontemporal_load(q + (i < IN16 ? i : IN16 - 1)); }
                }
                f.flush(r);
                // Deferred terminal metrics (pendq != nullptr, see dcm_env::side): if this event ends the episode and the env restarts
                // right away, the wave only parks the final record; calculate_waiting_time -- 6-7 us of this wave's 13-15, and this
                // wave is what a 4096-env launch waits for: 23.8 -> 15.9 us per step without it -- runs in k_terminal_flush later.
                // Not when an abandonment log overflowed into the count table (the restart clears it).  An env whose previous snapshot
                // is still waiting overwrites it: the summary row holds the LAST finished episode, the return log has the older one.
                bool defer = false;
                if (pendq && (mode & DCM_PARAM_AUTO_RESET)) {
                    const uint32_t ep = uni(((const Hdr*)smem)->episodes);
                    if (max_episodes == 0 || ep + 1 < max_episodes) {
                        bool spilled = false;
                        S.for_agents(lane, [&](int a) { spilled = spilled || (S.ainfo()[a] >> 16) > (uint32_t)AB_CAP; });
                        defer = !__any(spilled);
                    }
                }
                S.advance(h, P, lane, row PH_PASS, false, true, defer);
            } else {
                // masked / out-of-range action: simulated (or refused, DCM_PARAM_STRICT_MASK) by the general code, see apply_and_advance
                AMask gm;
                const int leader = S.pick_leader(h, lane, -1, k1, gm, false);
                if (leader >= 0)
                    S.apply_and_advance(h, P, lane, leader, gm, act_in, k1, -1, nullptr, row PH_PASS, RouteLog{nullptr, nullptr, nullptr, 0}, 0,
                                        false, (mode & DCM_PARAM_STRICT_MASK) ? 2 : 1, false, true, &xy);
            }
            h.now = uni(h.now); h.flags = uni(h.flags); h.cur_group = uni(h.cur_group); h.n_groups = uni(h.n_groups);
            h.empty_passes = uni(h.empty_passes); h.d = uni(h.d);
            // park the record (the LDS image is current: an episode only ever ends in the general code), the final time and the env's
            // abandonment rows (its own earlier stores: agent-scope loads, past the CU's vector L1), then announce it.  The rows are
            // requested here and stored behind the restart, which hides their round trip (1 us of this wave, the launch's slowest)
            const bool deferred = (h.flags & SimT::FLAG_DEFERRED) != 0u;
            unsigned char* const sp = side + (size_t)e * side_pitch;
            constexpr int ABN = 64 * (AB_CAP / 4) / WAVE;            // rows of at most 64 agents, as 64-bit words per lane
            unsigned long long ab[ABN] = {};
            const int nab = S.A() * (AB_CAP / 4);
            if (deferred) {
                h.flags &= ~SimT::FLAG_DEFERRED;
                const unsigned long long* src = (const unsigned long long*)S.ablog();
#pragma unroll
                for (int k = 0; k < ABN; k++)
                    if (lane + k * WAVE < nab) ab[k] = __hip_atomic_load(src + lane + k * WAVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WSYNC();
                copy16(sp, smem, L.rec_bytes(), lane);
                if (lane == 0) {
                    ((Hdr*)sp)->now = h.now;          // (after lane 0's own copy of the header piece: same lane, same address, in order)
                    pendq[e] = 1u;
                }
            } else if (pendq && (h.flags & DCM_FLAG_DONE) && lane == 0) {
                // the episode ended with its metrics computed here (masked action, overflowed log, last episode of the handle's budget):
                // a snapshot of an earlier episode must not overwrite the row later
                pendq[e] = 0u;
            }
            // DCM_PARAM_AUTO_RESET: the episode has just ended -> start the next one from the loaded instance (see k_step); an episode
            // only ever ends in the general code, so the LDS image is current here
            if ((mode & DCM_PARAM_AUTO_RESET) && (h.flags & DCM_FLAG_DONE) && !(h.flags & ERR) &&
                (max_episodes == 0 || uni(((const Hdr*)smem)->episodes) < max_episodes)) {
                if (deferred && init) {
                    // (a deferred end is a plain one: the image was requested.  LDS operations of a wave execute in order, so the
                    //  snapshot's reads of the old image are done)
                    const uint32_t ep = uni(((const Hdr*)smem)->episodes);
                    u32x4* d = (u32x4*)smem;
#pragma unroll
                    for (uint32_t c = 0; c < ICH; c++) { const uint32_t i = c * WAVE + lane; d[i < IN16 ? i : IN16 - 1] = iv[c]; }
                    WSYNC();
                    const HdrRegs hi = load_hdr(smem);                  // time 0, first group of the first event; seed unchanged since dcm_reset
                    h.now = hi.now; h.flags = hi.flags; h.cur_group = hi.cur_group; h.n_groups = hi.n_groups; h.empty_passes = hi.empty_passes;
                    if (lane == 0) { ((Hdr*)smem)->episodes = ep; *S.dirty() = SimT::DIRTY_ALL; }
                } else {
                    S.reset_state(h, lane);
                    if (lane == 0) *S.dirty() = SimT::DIRTY_ALL;
                    S.advance(h, P, lane, row PH_PASS, false);
                    h.now = uni(h.now); h.flags = uni(h.flags); h.cur_group = uni(h.cur_group); h.n_groups = uni(h.n_groups);
                    h.empty_passes = uni(h.empty_passes);
                }
            }
            if (deferred) {
                unsigned long long* dst = (unsigned long long*)(sp + L.rec_bytes());
#pragma unroll
                for (int k = 0; k < ABN; k++)
                    if (lane + k * WAVE < nab) dst[lane + k * WAVE] = ab[k];
            }
            finish(false);
            return;
        }
        // wave-uniform by construction; tell the compiler so
        h.now = uni(h.now); h.flags = uni(h.flags); h.cur_group = uni(h.cur_group); h.n_groups = uni(h.n_groups);
        h.empty_passes = uni(h.empty_passes); h.d = uni(h.d);
    }
    finish(regs);
}


// Reward + perf metrics (env/task_env.py:344-364,420-425, worker.py:103-108) of the episodes whose final records k_step_fast parked
// (dcm_env::side): one workgroup per env, those without a waiting snapshot leave at once.
template <int CA, int CT, bool RS>
__global__ __launch_bounds__(WAVE, DCM_STEP_WAVES) void k_terminal_flush(int A, int T, int PA, int PT, KP P, const unsigned char* side, uint32_t side_pitch,
                                                        uint32_t* pendq, double* summary, const int32_t* sizes, unsigned char* gscr) {
    const int e = env_of_workgroup(), lane = threadIdx.x;
    if (uni(pendq[e]) == 0u) return;
    int eA, eT;
    env_dims<CA, CT, RS>(sizes, e, A, T, eA, eT);
    using SimT = Sim<CA, CT, RS, false>;
    SimT S{eA, eT, PA, PT, smem, nullptr};
    const Lay L = S.L();
    S.scr = step_scratch_in_lds<CA, CT>() ? smem + SimT::lds_image_bytes(L) + 512u : gscr + (size_t)e * L.scratch_bytes();
    const unsigned char* sp = side + (size_t)e * side_pitch;
    typename SimT::XY xy;
    S.template load_record<false>(sp, lane, xy);                 // (every load in flight before the first LDS write)
    if (lane == 0) {                                             // the image's pointers: this snapshot's abandonment rows, nothing else
        *(const uint16_t**)(smem + S.aux_off()) = (const uint16_t*)(sp + L.rec_bytes());
        *(uint8_t**)(smem + S.aux_off() + 16) = nullptr;         // (a log that overflowed into the count table is never deferred)
        *(double**)(smem + S.aux_off() + 32) = nullptr;
    }
    WSYNC();
    const double now = uni(((const Hdr*)smem)->now);
    (void)SimT::terminal_metrics(S, now, P.mwt, lane, summary + (size_t)e * 8);
    if (lane == 0) pendq[e] = 0u;
}
