"""bench.py's line layout (CPU): the summary object is last, compact, and auxiliary failures are found."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _line():
    big = {"workload": "x" * 3000}
    return {"metric": "env_steps_per_sec", "value": 1.4e9, "unit": "steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.05,
            "ms_per_step_min": 1.01, "ms_per_step_max": 1.2, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "roofline": {"kernel": "k_rollout_fast", "frac": 0.5234567, "frac_hi": 0.6, "salu_issue_frac": 0.4, "stale": False, "bulk": "y" * 500},
            "parity": {"envs_checked": 4096, "mismatches": 0, "fields": ["a"] * 20},
            "config": dict(big),
            "lockstep_kernel": {"envs": 65536, "frac": 0.75, "traffic_frac": 0.57, "rocprof_avg_launch_us": 143.7, "steady_state_us": 16.5},
            "other_configs": {"config4_shard": {"value": 8.1e8, "roofline": {"frac": 0.45}, "parity": {"envs_checked": 8192, "mismatches": 0},
                                                "cpu_baseline": {"value": 1e7}, "workload": "w" * 2000},
                              "config3": {"value": 5e6}},
            "cpu_baseline": {"value": 1.7e7, "cores": 16, "sample": "s" * 400}}


def test_summary_is_last_and_fits_the_tail():
    import bench
    out = bench.order_line(_line())
    keys = list(out)
    assert keys[-4:] == ["roofline", "parity", "cpu_baseline", "summary"]
    assert keys.index("config") < keys.index("roofline") and keys.index("other_configs") < keys.index("roofline")
    assert keys[:3] == ["metric", "value", "unit"]
    s = json.dumps(out)
    tail = s[-2000:]
    sm = json.dumps(out["summary"])
    assert len(sm) < 1500 and tail.endswith(sm + "}")
    assert out["summary"]["roofline"]["frac"] == 0.5235 and out["summary"]["parity"] == {"envs_checked": 4096, "mismatches": 0}
    assert out["summary"]["config4_shard"] == {"value": 8.1e8, "frac": 0.45, "parity_mismatches": 0, "parity_envs": 8192, "cpu": 1e7}
    assert out["summary"]["lockstep_kernel"]["steady_state_us"] == 16.5


def test_auxiliary_failures_are_reported():
    import bench
    j = _line()
    assert bench.aux_failures(j) == []
    j["other_configs"]["config4_shard"]["parity"]["mismatches"] = 3
    j["other_configs"]["config3"] = {"error": "RuntimeError: boom"}
    j["lockstep_kernel"] = {"error": "x"}
    bad = bench.aux_failures(j)
    assert len(bad) == 3 and any("3 of 8192" in b for b in bad) and any("boom" in b for b in bad)
