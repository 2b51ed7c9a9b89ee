"""Roofline arithmetic of the hot path.

Algorithmic bytes per env step: W = 2*S + O + 4 -- what the lockstep kernel k_step really moves per
decision: that kernel is priced against the HBM peak.  The persistent kernels (k_rollout_random, k_replay) keep the record
in LDS for whole episodes and are bound by instruction issue instead; their utilisation is computed from SQ instruction
counters (profiles/counters.json, written by tools/collect_profile.py from the rocprofv3 CSVs under profiles/) priced PER
INSTRUCTION CLASS with the clocks measured by tools/calib/pmc_calib.hip (profiles/r03_calib):

  * `SQ_ACTIVE_INST_VALU` reads 1.0 per wave64 VALU instruction whatever the instruction costs, so it is an instruction count,
    not a busy time (round 2 priced every instruction at 4 clocks with it);
  * measured SIMD time per wave64 instruction at 8 waves per SIMD (profiles/r06_calib): 2.35 clocks for 32-bit VOP1 / VOP2
    instructions in their e32 encoding (integer add / logic / shift, v_mov_b32, fp32); 4.2-4.3 clocks for everything else -- fp64
    arithmetic / compare / convert, v_mov_b64, DPP (also folded into a VOP2), v_readlane, v_mbcnt, VOP3 with a scalar mask or
    destination (v_cndmask_e64, v_cmp_*_e64), v_bfe_u32, v_lshlrev_b64;
  * which share of a kernel's EXECUTED VALU instructions is in the 2.35 class has no counter: it comes from the ISA
    (tools/loop_insts.py --phases: the decision loop's instructions classified and weighted with measured trip counts,
    profiles/r06_budget.md) and is kept in profiles/isa_shares.json;
  * one scalar unit per CU issues 1 SALU instruction per 1.07 clocks (4.27 clocks per SIMD with all four SIMDs issuing).
"""
import json
import os


def state_bytes(A, T):
    """Canonical compact state of one env: 64 B globals + 48 B/agent + 96 B/task."""
    return 64 + 48 * A + 96 * T


def observation_bytes(A, T):
    """f32[A,6] + f32[T+1,5] + u8[T+1] (worker.py:57-68)."""
    return 24 * A + 21 * (T + 1)


def algorithmic_bytes_per_step(A, T):
    """read state + write state + write observation + read action."""
    return 2 * state_bytes(A, T) + observation_bytes(A, T) + 4


HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E peak (MI355X_MICROARCH.md)
N_CU = 256
N_SIMD = N_CU * 4              # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9          # peak engine clock (MI355X_MICROARCH.md per-instruction table: "256 CU x 2.4 GHz")

# profiles/r03_calib/table.txt: SIMD clocks per wave64 instruction, throughput at 8 waves per SIMD
CLOCKS_VALU_64 = 4.25          # v_fma/add/min/cmp_f64, v_cvt_f32_f64, v_mov_b32_dpp, v_readlane, v_mbcnt, v_cndmask_e64 (SGPR mask)
CLOCKS_VALU_32 = 2.35          # v_add_u32, v_fma_f32, v_mov_b32
CLOCKS_SALU_PER_CU = 1.07      # s_add_u32 / s_mul_i32: 4.27 clocks per SIMD-instruction with 4 SIMDs sharing the scalar unit
CALIB_SOURCE = "profiles/r06_calib/table.txt"
# SQ_INSTS_VALU_* classes that the calibration prices at CLOCKS_VALU_64
F64_CLASS_COUNTERS = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                      "SQ_INSTS_VALU_CVT")

_COUNTERS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "counters.json")


def load_counters(key):
    """Per-decision / per-launch PMC averages of one profiled workload, or None when none is committed for it."""
    try:
        with open(_COUNTERS) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


_ISA_SHARES = os.path.join(os.path.dirname(_COUNTERS), "isa_shares.json")


def isa_e32_share(kernel):
    """Share of the kernel's executed VALU instructions that are in the 2.35-clock class (profiles/isa_shares.json), or None."""
    try:
        with open(_ISA_SHARES) as f:
            e = json.load(f).get(kernel)
        return float(e["valu_e32_share"]) if e else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def issue_roofline(c, units_per_step, step_s, unit="decision", e32_share=None):
    """Issue-bound roofline of a persistent (LDS-resident) kernel from its per-`unit` instruction counters `c`.

    VALU pipe time per unit = sum over instruction classes of count x measured clocks.  The class counters cover the fp64
    arithmetic and conversions; the remaining instructions (moves, selects, compares, integer and cross-lane work) are priced
    at the 32-bit rate for `frac` (an estimate that can only be low: part of them are 4-clock instructions) and at the 64-bit
    rate for `frac_hi` (every VALU instruction at 4.25 clocks -- the upper bound; round 2's figure).  With `e32_share` -- the
    ISA-derived share of executed VALU instructions in the 2.35-clock class (isa_e32_share) -- `frac` prices exactly that share at
    2.35 and the rest at 4.25: one number instead of a bracket (the class-counter estimate moves to `frac_class_counters`).  Both are fractions of
    1024 SIMDs x 2.4 GHz.  `salu_issue_frac`: scalar-unit issue slots used, 256 CUs x 2.4 GHz / 1.07 clocks per instruction; when
    it exceeds `frac_hi` the scalar unit is the binding resource: `bound` = "salu_issue" and `achieved` / `peak` / `frac` describe
    it (the VALU figures move to `valu_issue_frac` / `valu_issue_frac_hi`)."""
    n_valu = c[f"SQ_INSTS_VALU_per_{unit}"]
    have_classes = all(f"{k}_per_{unit}" in c for k in F64_CLASS_COUNTERS)
    n64 = sum(c[f"{k}_per_{unit}"] for k in F64_CLASS_COUNTERS) if have_classes else 0.0
    lo_counters = CLOCKS_VALU_64 * n64 + CLOCKS_VALU_32 * (n_valu - n64)
    hi = CLOCKS_VALU_64 * n_valu
    lo = n_valu * (e32_share * CLOCKS_VALU_32 + (1.0 - e32_share) * CLOCKS_VALU_64) if e32_share is not None else lo_counters
    peak = N_SIMD * PEAK_CLOCK_HZ
    per_s = units_per_step / step_s
    out = {"bound": "valu_issue", "achieved": lo * per_s / 1e9, "peak": peak / 1e9, "unit": "G SIMD-clocks/s (VALU pipe busy)",
           "frac": lo * per_s / peak, "frac_hi": hi * per_s / peak,
           "pricing": {"clocks_fp64_class": CLOCKS_VALU_64, "clocks_other": CLOCKS_VALU_32, "fp64_class_insts_per_" + unit: n64,
                       "valu_insts_per_" + unit: n_valu, "class_counters": have_classes, "calibration": CALIB_SOURCE,
                       "valu_e32_share_isa": e32_share,
                       "note": ("frac prices the ISA-derived share of 32-bit e32 instructions at 2.35 clocks and the rest at 4.25 "
                                "(profiles/isa_shares.json, profiles/r06_budget.md); frac_hi prices every VALU instruction at 4.25; "
                                "frac_class_counters is the old low estimate (only the fp64 class counters at 4.25)")
                               if e32_share is not None else
                               "frac prices the fp64-class instructions at 4.25 clocks and all others at 2.35 (low estimate); "
                               "frac_hi prices every VALU instruction at 4.25"}}
    if e32_share is not None:
        out["frac_class_counters"] = lo_counters * per_s / peak
    n_salu = c.get(f"SQ_INSTS_SALU_per_{unit}")
    if n_salu is not None:
        salu = n_salu * CLOCKS_SALU_PER_CU * per_s / (N_CU * PEAK_CLOCK_HZ)
        out["salu_issue_frac"] = salu
        out["salu_insts_per_" + unit] = n_salu
        if salu > out["frac_hi"]:
            # the CU's single scalar unit is busier than the four VALU pipes can be even at the upper price (route replay since
            # its state left LDS for 14 resident waves per CU): that is the binding resource, and `frac` describes it; the VALU
            # figures stay on record next to it
            out.update({"bound": "salu_issue", "valu_issue_frac": out["frac"], "valu_issue_frac_hi": out["frac_hi"],
                        "achieved": n_salu * CLOCKS_SALU_PER_CU * per_s / 1e9, "peak": N_CU * PEAK_CLOCK_HZ / 1e9,
                        "unit": "G scalar-unit clocks/s (one scalar unit per CU, 1.07 clocks per instruction)", "frac": salu})
            del out["frac_hi"]
    if f"SQ_THREAD_CYCLES_VALU_per_{unit}" in c and f"SQ_ACTIVE_INST_VALU_per_{unit}" in c:
        out["lane_util"] = c[f"SQ_THREAD_CYCLES_VALU_per_{unit}"] / (64.0 * c[f"SQ_ACTIVE_INST_VALU_per_{unit}"])
    wave = c.get(f"SQ_WAVE_CYCLES_per_{unit}")
    if wave:
        out["wave_time_split"] = {k: c[f"{name}_per_{unit}"] / wave for k, name in
                                  (("executing", "SQ_ACTIVE_INST_ANY"), ("parked_on_waitcnt", "SQ_WAIT_ANY"),
                                   ("issue_stalled", "SQ_WAIT_INST_ANY")) if f"{name}_per_{unit}" in c}
    return out


def rollout_kernel_name(A, T):
    """Which persistent rollout kernel dcm_rollout_random launches for a uniform batch of this shape (all three observation
    buffers given): the register-resident kernels for the one-chunk layouts, for 50A/200T and for the mid-size class (A <= 128,
    T <= 256), the general one otherwise."""
    if A <= 64 and T <= 63:
        return "k_rollout_fast"
    if (A, T) == (50, 200):
        return "k_rollout_fast_mc"
    if A <= 128 and T <= 256 and not (A <= 64 and T <= 64):
        return "k_rollout_fast_g"
    return "k_rollout_random"


def replay_kernel_name(A, T, member_cap, reactive, vis_cap):
    """Which kernel dcm_execute_routes launches (default placement): the register-resident one when the agents, the LIVE tasks (all
    of them without dynamic arrivals, tasks 1..cap with them) and the member slots fit it, the general one otherwise."""
    live = min(T, vis_cap) if reactive else T
    return "k_replay_fast" if (A <= 128 and live <= 128 and member_cap <= 8) else "k_replay"


def step_kernel_name(A, T):
    """The lockstep kernel dcm_step launches for the plain call shape (no injected choices, no route log, all outputs)."""
    return "k_step_fast" if (A <= 64 and T <= 63) else "k_step"


def staleness(c, build_id):
    """True when the counter set was measured on another build of the kernels than the loaded library (dcm_build_id)."""
    return c.get("build_id") != build_id
