"""CPU: host logic and the C-ABI library (load + exported symbols; no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dcmrta_amd import _lib
    hdr = open(_lib.HEADER_PATH).read()
    declared = set(re.findall(r"\b(dcm_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.dcm_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to run instead of silently computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    with pytest.raises(_lib.DcmError):
        BatchedTaskEnv(4, 20, 50, device="cuda:0")
    with pytest.raises(_lib.DcmError):
        BatchedTaskEnv(4, 20, 50, device="cpu")
    lib = _lib.load()
    p = _lib.DcmParams(4, 20, 50, 0, 10.0, 100.0, 0, 0)
    h = C.c_void_p()
    assert lib.dcm_create(C.byref(p), C.byref(h)) == -3
    assert b"no HIP device" in lib.dcm_last_error()


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dcmrta_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src and "dcmrta_oracle" not in src, f


def test_instance_generator_matches_reference_draw_order(golden_dir):
    """generate_instance == TaskEnv(seed=s) of the reference (arrays stored in the golden traces)."""
    from dcmrta_amd.instances import generate_batch, generate_instance
    for p in H.full_traces():
        tr = H.load_trace(p)
        if int(tr["inst_seed"]) < 0:
            continue  # hand-built symmetric instances (micro_*.npz)
        g = generate_instance(int(tr["A"]), int(tr["T"]), int(tr["inst_seed"]))
        for k in ("depot", "task_xy", "req", "dur"):
            assert np.array_equal(g[k], tr[k]), (p, k)
    b = generate_batch(3, 20, 50, base_seed=5, first=2)
    g = generate_instance(20, 50, 8)
    assert np.array_equal(b["task_xy"][1], g["task_xy"]) and np.array_equal(b["req"][1], g["req"])


def test_record_size_formula():
    """S(A,T) = 64 + 48A + 96T and W = 2S + O + 4 as used by bench.py."""
    from dcmrta_amd.roofline import algorithmic_bytes_per_step, state_bytes
    assert state_bytes(20, 50) == 5824 and algorithmic_bytes_per_step(20, 50) == 13203
    assert algorithmic_bytes_per_step(50, 200) == 48753 and algorithmic_bytes_per_step(100, 500) == 118653


def test_ctasd_yaml_export_matches_shipped_files(golden_dir, tmp_path):
    """dcmrta_amd.ctasd_io against the planner inputs the reference ships (written by TestSetGenerator.py:40-116)."""
    import hashlib
    import json

    import yaml
    from dcmrta_amd.ctasd_io import export_ctasd_yaml
    from dcmrta_amd.instances import load_instances_npz
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    dig = json.load(open(os.path.join(golden_dir, "ctasd_yaml_digest.json")))
    for i, ref in dig.items():
        i = int(i)
        folder = ref["planner_param"]["graphFile"].split("/")[1]
        export_ctasd_yaml(str(tmp_path / f"env_{i}"), inst["depot"][i], inst["task_xy"][i], inst["req"][i], inst["dur"][i], A,
                          folder=folder, index=i, solver_time=ref["planner_param"]["solverMaxTime"])
        for name, want in ref["sha256"].items():
            doc = yaml.safe_load(open(tmp_path / f"env_{i}" / f"{name}.yaml"))
            got = hashlib.sha256(json.dumps(doc, sort_keys=True).encode()).hexdigest()
            assert got == want, (i, name)
        # graph.yaml: same keys in the same order, same integer columns; weights to 1e-14 (the shipped files were written
        # with an older math.hypot that differs in the last ulp)
        g = yaml.safe_load(open(tmp_path / f"env_{i}" / "graph.yaml"))["vehicle0"]
        assert hashlib.sha256(json.dumps(list(g.keys()), sort_keys=True).encode()).hexdigest() == ref["graph_keys_sha256"]
        z = np.load(os.path.join(golden_dir, "ctasd_graph.npz"))
        edges = [k for k in g if k.startswith("edge")]
        assert np.array_equal(np.array([[g[k][0], g[k][1], g[k][2], g[k][4]] for k in edges]), z[f"ends_{i}"])
        np.testing.assert_allclose([g[k][3] for k in edges], z[f"dist_{i}"], rtol=1e-14)
        np.testing.assert_allclose([g[k][5] for k in edges], z[f"time_{i}"], rtol=1e-14)
        assert np.array_equal(np.array([g[k] for k in g if k.startswith("node")]), z[f"node_{i}"])


def test_read_ctasd_routes(tmp_path):
    import yaml
    from dcmrta_amd.ctasd_io import read_ctasd_routes
    (tmp_path / "p.yaml").write_text(yaml.dump({"flagSolver": "TEAMPLANNER_CONDET", "vehNum": 1, "vehNumPerType": [3]}))
    (tmp_path / "r.yaml").write_text(yaml.dump({"vehicle": {"vv1": {"node": [0, 3, 1, 0]}, "vv2": {"node": [0]}, "vv3": {"node": [0, 2, 0]}}}))
    assert read_ctasd_routes(tmp_path / "r.yaml", tmp_path / "p.yaml") == [[3, 1, 0], None, [2, 0]]
    (tmp_path / "r2.yaml").write_text(yaml.dump({"result": {"flagSuccess": 0}}))
    assert read_ctasd_routes(tmp_path / "r2.yaml", tmp_path / "p.yaml") is None


def test_instance_generator_with_ranges(golden_dir):
    """TaskEnv((10,20),(20,50),...,seed=s) of the reference: sizes and arrays (tests/golden/instances_ranges.json)."""
    import json
    from dcmrta_amd.instances import generate_instance_ranges
    ref = json.load(open(os.path.join(golden_dir, "instances_ranges.json")))
    for sd, r in ref.items():
        A, inst = generate_instance_ranges((10, 20), (20, 50), int(sd))
        assert (A, len(inst["req"])) == (r["A"], r["T"])
        assert inst["depot"].tolist() == r["depot"] and inst["task_xy"][0].tolist() == r["task_xy0"]
        assert inst["task_xy"][-1].tolist() == r["task_xy_last"] and inst["req"].tolist() == r["req"]


def test_ragged_batch_generator_matches_reference_sizes(golden_dir):
    """generate_batch_ranges pads the per-seed instances of the reference (instances_ranges.json) to the range maxima."""
    import json
    from dcmrta_amd.instances import generate_batch_ranges
    ref = json.load(open(os.path.join(golden_dir, "instances_ranges.json")))
    seeds = [int(s) for s in ref]
    inst = generate_batch_ranges(seeds, (10, 20), (20, 50))
    assert inst["task_xy"].shape == (len(seeds), 50, 2) and inst["req"].shape == (len(seeds), 50)
    for b, s in enumerate(seeds):
        r = ref[str(s)]
        a, t = int(inst["n_agents"][b]), int(inst["n_tasks"][b])
        assert (a, t) == (r["A"], r["T"]) and inst["depot"][b].tolist() == r["depot"]
        assert inst["req"][b, :t].tolist() == r["req"] and inst["task_xy"][b, t - 1].tolist() == r["task_xy_last"]
        assert (inst["req"][b, t:] == 1).all() and not inst["task_xy"][b, t:].any() and (inst["dur"][b, :t] == 5.0).all()


def _traj_fixture(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    A = z["route"].shape[0]
    routes = []
    for a in range(A):
        n = int((z["route"][a] != -2).sum())
        routes.append(([int(x) for x in z["route"][a, :n]], [float(x) for x in z["arrival"][a, :n]]))
    members = [[int(x) for x in row if x >= 0] for row in z["members"]]
    ends = np.cumsum(z["traj_len"])
    ref = [z["traj"][e - n:e] for e, n in zip(ends, z["traj_len"])]
    return z, routes, members, ref


@pytest.mark.parametrize("name", ["traj_5A8T_random_s3.npz", "traj_10A20T_nearest_s4.npz", "traj_6A9T_random_s5.npz"])
def test_generate_traj_matches_reference(golden_dir, name):
    """trajectory.trajectories restates generate_traj (env/task_env.py:375-418): sampled (x, y, heading) of every agent,
    bit for bit, from the reference's own routes / member lists / task times (tests/golden/make_golden_extra.py)."""
    from dcmrta_amd.trajectory import trajectories
    z, routes, members, ref = _traj_fixture(golden_dir, name)
    got = trajectories(routes, z["depot"], z["task_xy"], members, z["feasible"].astype(bool), z["time_start"], z["time_finish"],
                       float(z["current_time"]))
    assert len(got) == len(ref)
    for a, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape and np.array_equal(g, r), (name, a)


def test_instances_from_reference_dicts(golden_dir):
    """instance_from_dicts / batch_from_dicts read the reference's task_dic / agent_dic / depot layout (env/task_env.py:76-113)."""
    from dcmrta_amd.instances import batch_from_dicts, instance_from_dicts, load_instances_npz
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))

    def dicts(i):
        return ({t: {"ID": t, "requirements": np.array([inst["req"][i][t]]), "location": inst["task_xy"][i][t], "time": np.array([inst["dur"][i][t]])}
                 for t in range(50)}, {a: {"ID": a} for a in range(A)}, {"location": inst["depot"][i], "ID": -1})
    a, one = instance_from_dicts(*dicts(2))
    assert a == A and all(np.array_equal(one[k], inst[k][2]) for k in ("depot", "task_xy", "req", "dur"))
    a, many = batch_from_dicts([dicts(i) for i in (0, 4, 9)])
    assert a == A and all(np.array_equal(many[k], inst[k][[0, 4, 9]]) for k in ("depot", "task_xy", "req", "dur"))


def test_make_test_set_example(tmp_path, golden_dir):
    """examples/make_test_set.py (TestSetGenerator.py:1-116): instance i = TaskEnv((A,A), (T,T), seed=i) in the reference's draw
    order.  Depot and task coordinates equal the reference's shipped test set (same seeds, same first draws -- the shipped
    pickles come from an older generator with random durations, so requirements / durations are compared with the current
    draw order instead); the planner input files are written."""
    import subprocess
    import sys
    from dcmrta_amd.instances import generate_instance_ranges, load_instances_npz
    out = tmp_path / "testSet_20A_50T_CONDET"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "make_test_set.py"), "--out", str(out), "--num", "6"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got, A = load_instances_npz(str(out / "instances_20A50T.npz"))
    want, A0 = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    assert A == A0 == 20
    for k in ("depot", "task_xy"):
        assert np.array_equal(got[k], want[k][:6]), k
    for i in range(6):
        _, ref = generate_instance_ranges((20, 20), (50, 50), i)            # draw order pinned against the reference in test_host
        assert np.array_equal(got["req"][i], ref["req"]) and np.array_equal(got["dur"][i], ref["dur"]) and (got["dur"][i] == 5.0).all()
        for name in ("vehicle_param", "task_param", "planner_param", "graph"):
            assert (out / f"env_{i}" / f"{name}.yaml").exists()


def test_persistent_kernel_keeps_four_waves_per_simd(tmp_path):
    """BASELINE configs[1] is 4096 envs = exactly four single-wave workgroups per SIMD (1024 SIMDs): the one-chunk instantiations of
    the persistent kernel must stay within 128 VGPRs, or a 4096-env batch needs two rounds of workgroups (a 30 % drop that round 3
    hit when a transient 32-VGPR copy pushed one instantiation to 132).  Checked on the compiler's own resource report."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "dcmrta_amd", "csrc", "dcmrta_env.hip")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S",
                          "--cuda-device-only", src, "-o", str(tmp_path / "env.s"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    usage = {}
    spills = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", out.stderr, re.S):
        usage[m.group(1)] = (int(m.group(2)), int(m.group(3)))
        spills[m.group(1)] = int(m.group(4))
    seen = 0
    for name, (vgprs, occ) in usage.items():
        # the persistent kernels of the one-chunk layouts (general and register-resident form, with the observation stores): all
        # 4096 envs of the BASELINE batch resident at once = four waves per SIMD
        if ("k_rollout_random" in name and any(t in name for t in ("ILi20ELi50ELb0ELi5E", "ILi20ELi50ELb1ELi5E", "ILi64ELi64ELb1ELi5E"))) or \
                ("k_rollout_fastI" in name and any(t in name for t in ("ILi20ELi50ELb0ELb1E", "ILi20ELi50ELb1ELb1E", "ILi64ELi64ELb1ELb1E"))):
            assert vgprs <= 128 and occ >= 4, (name, vgprs, occ)
            seen += 1
        if "k_rollout_fast_mcILi50ELi200ELb1E" in name:                  # config 4: three waves per SIMD
            assert vgprs <= 168 and occ >= 3, (name, vgprs, occ)
            seen += 1
        if "k_rollout_fast_gI" in name:                                   # the mid-size class: two waves per SIMD ...
            assert occ >= 2, (name, vgprs, occ)
            if "ILi2ELi3ELb1E" in name or "ILi1ELi2ELb1E" in name:       # ... without spills for the 70A/130T- and 30A/100T-class batches
                assert spills[name] == 0, (name, spills[name])
            seen += 1
    # 3 general + 6 register-resident (three layouts, with and without the wave-priority bookkeeping) + config 4 + 12 mid-size
    assert seen == 10 + 12, sorted(usage)


def test_synthetic_route_arrays_match_the_list_form():
    """The batch / array form of the benchmark's preset routes (bench.py --config 5) == the per-env list form."""
    from dcmrta_amd.instances import generate_batch, synthetic_route_arrays, synthetic_routes
    inst = generate_batch(7, 13, 37, base_seed=3)
    for max_task in (None, 20, 100):
        routes, length = synthetic_route_arrays(inst["req"], 13, max_task)
        assert routes.dtype == np.int32 and length.dtype == np.int32 and routes.shape[:2] == (7, 13)
        for b in range(7):
            ref = synthetic_routes(inst["req"][b], 13, max_task)
            for a in range(13):
                assert list(routes[b, a, :length[b, a]]) == ref[a] and (routes[b, a, length[b, a]:] == 0).all()
                assert ref[a][-1] == 0                                   # every route ends at the depot


def test_issue_roofline_pricing():
    """roofline.issue_roofline: fp64-class instructions at 4.25 clocks, the rest at 2.35 (frac), everything at 4.25 (frac_hi), the
    scalar unit at 1.07 clocks per instruction per CU; a counter set of another kernel build is reported stale."""
    from dcmrta_amd import roofline as R
    c = {"SQ_INSTS_VALU_per_decision": 100.0, "SQ_INSTS_SALU_per_decision": 50.0, "SQ_INSTS_VALU_ADD_F64_per_decision": 10.0,
         "SQ_INSTS_VALU_MUL_F64_per_decision": 2.0, "SQ_INSTS_VALU_FMA_F64_per_decision": 3.0, "SQ_INSTS_VALU_TRANS_F64_per_decision": 1.0,
         "SQ_INSTS_VALU_CVT_per_decision": 4.0, "SQ_THREAD_CYCLES_VALU_per_decision": 3200.0, "SQ_ACTIVE_INST_VALU_per_decision": 100.0,
         "SQ_WAVE_CYCLES_per_decision": 1000.0, "SQ_ACTIVE_INST_ANY_per_decision": 400.0, "SQ_WAIT_ANY_per_decision": 500.0,
         "SQ_WAIT_INST_ANY_per_decision": 100.0, "build_id": "abc"}
    r = R.issue_roofline(c, units_per_step=1e6, step_s=1e-3)
    busy = 4.25 * 20 + 2.35 * 80
    assert abs(r["frac"] - busy * 1e9 / (1024 * 2.4e9)) < 1e-12 and abs(r["frac_hi"] - 425 * 1e9 / (1024 * 2.4e9)) < 1e-12
    assert abs(r["salu_issue_frac"] - 50 * 1.07 * 1e9 / (256 * 2.4e9)) < 1e-12 and r["lane_util"] == 0.5
    assert r["wave_time_split"] == {"executing": 0.4, "parked_on_waitcnt": 0.5, "issue_stalled": 0.1}
    assert r["frac"] <= r["frac_hi"] and r["pricing"]["class_counters"] is True and r["bound"] == "valu_issue"
    assert R.staleness(c, "abc") is False and R.staleness(c, "abd") is True and R.staleness({}, "abc") is True
    # with the ISA-derived share of 2.35-clock instructions `frac` is that exact mix; the class-counter estimate stays on record
    r1 = R.issue_roofline(c, 1e6, 1e-3, e32_share=0.25)
    assert abs(r1["frac"] - 100 * (0.25 * 2.35 + 0.75 * 4.25) * 1e9 / (1024 * 2.4e9)) < 1e-12 and r1["frac_hi"] == r["frac_hi"]
    assert abs(r1["frac_class_counters"] - r["frac"]) < 1e-15 and r1["pricing"]["valu_e32_share_isa"] == 0.25
    assert abs(r1["frac"] - r1["achieved"] / r1["peak"]) < 1e-12 and r["frac"] < r1["frac"] < r1["frac_hi"]
    sh = R.isa_e32_share("k_rollout_fast")
    assert sh is not None and 0.1 < sh < 0.5 and R.isa_e32_share("no_such_kernel") is None
    del c["SQ_INSTS_VALU_CVT_per_decision"]                            # no class counters: every instruction at the 32-bit rate
    r2 = R.issue_roofline(c, 1e6, 1e-3)
    assert r2["pricing"]["class_counters"] is False and abs(r2["frac"] - 235 * 1e9 / (1024 * 2.4e9)) < 1e-12
    # the committed counter sets all describe ONE kernel build
    import json
    allc = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    assert len({v.get("build_id") for v in allc.values()}) == 1 and all(v.get("build_id") for v in allc.values())
    for key in ("k_rollout_fast:20A50T", "k_rollout_fast_mc:50A200T", "k_replay_fast:100A500T", "k_step:4096x20A50T", "k_step:65536x20A50T"):
        assert key in allc, key                                         # every BASELINE config's dominant kernel has a profile
    # a kernel whose scalar-unit time exceeds even the upper VALU price is bound by the CU's one scalar unit
    c3 = dict(c, **{"SQ_INSTS_SALU_per_decision": 400.0})
    r3 = R.issue_roofline(c3, 1e6, 1e-3)
    assert r3["bound"] == "salu_issue" and abs(r3["frac"] - 400 * 1.07 * 1e9 / (256 * 2.4e9)) < 1e-12 and r3["frac"] == r3["salu_issue_frac"]
    assert abs(r3["frac"] - r3["achieved"] / r3["peak"]) < 1e-12 and r3["valu_issue_frac"] <= r3["valu_issue_frac_hi"] < r3["frac"]


def test_kernel_name_helpers_follow_the_dispatch():
    """roofline.rollout_kernel_name / step_kernel_name (which rocprofv3 kernel a bench line is priced with) restate the shape
    dispatch of dcm_rollout_random / dcm_step (csrc/dcmrta_env.hip): one-chunk layouts, 50A/200T, the mid-size class, the rest."""
    from dcmrta_amd.roofline import replay_kernel_name, rollout_kernel_name, step_kernel_name
    # dcm_execute_routes (csrc/dcmrta_replay.hip): the register-resident kernel up to 128 agents / 128 LIVE tasks / 8 member slots
    assert replay_kernel_name(100, 500, 5, True, 100) == replay_kernel_name(20, 50, 8, False, 100) == "k_replay_fast"
    assert replay_kernel_name(100, 500, 5, False, 100) == replay_kernel_name(100, 500, 5, True, 500) == "k_replay"
    assert replay_kernel_name(20, 50, 16, False, 100) == "k_replay" and replay_kernel_name(100, 128, 5, False, 100) == "k_replay_fast"
    assert rollout_kernel_name(20, 50) == rollout_kernel_name(15, 35) == rollout_kernel_name(64, 63) == "k_rollout_fast"
    assert rollout_kernel_name(50, 200) == "k_rollout_fast_mc"
    for shape in ((70, 130), (65, 65), (128, 256), (30, 100), (100, 64), (64, 65), (65, 10)):
        assert rollout_kernel_name(*shape) == "k_rollout_fast_g", shape
    for shape in ((64, 64), (100, 500), (50, 257), (128, 300)):     # T = 64 of the <64,64> layout has no free depot lane; larger shapes
        assert rollout_kernel_name(*shape) == "k_rollout_random", shape
    assert step_kernel_name(20, 50) == "k_step_fast" and step_kernel_name(70, 130) == "k_step" and step_kernel_name(64, 64) == "k_step"
