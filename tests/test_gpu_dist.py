"""-m gpu: the N > 1 launch path of bench.py end to end -- `python -m torch.distributed.run --nproc-per-node 2 bench.py
--gpus 2`, here as two ranks sharing the one GPU of the test box (DCM_FORCE_DEVICE=0) with the gloo backend (RCCL needs
one device per rank; on a multi-GPU node the same command runs over RCCL/xGMI).  Checks the contract line: aggregate
value over both ranks, weak scaling, the sharded env blocks and the return all-gather."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line(gpu_device):
    env = dict(os.environ, DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs", "512"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]          # rank 0 prints exactly one JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3 and j["warmup"] == 1
    assert j["metric"] == "env_steps_per_sec" and j["value"] > 0 and j["vs_baseline"] is None
    assert j["config"]["envs_per_gpu"] == 512 and "x2" in j["config"]["sharding"]
    # both ranks' decisions are counted: about 2 x 512 envs x 3 episodes x ~120 decisions per pass
    per_pass_all_ranks = j["value"] * j["ms_per_step"] / 1e3
    assert 2 * 512 * 3 * 80 < per_pass_all_ranks < 2 * 512 * 3 * 160
    assert "cpu_baseline" not in j                                           # rank 0 at N = 1 only


def test_single_gpu_bench_contract_line(gpu_device):
    """python bench.py (N = 1): one JSON line with the contract keys, the roofline object and the bounded CPU baseline."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-lockstep-probe"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"] == "f64" and j["data"] == "synthetic" and j["higher_is_better"] is True
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["envs_per_gpu"] == 4096
    r = j["roofline"]
    # the persistent kernel is issue-bound: frac = VALU-busy SIMD-cycles / available SIMD-cycles, a utilisation (<= 1)
    assert r["bound"] == "valu_issue" and r["kernel"] == "k_rollout_fast" and r["peak"] == 1024 * 2.4
    assert r["frac"] is not None and 0.05 < r["frac"] <= r["frac_hi"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["salu_issue_frac"] <= 1.0 and r["pricing"]["calibration"].startswith("profiles/")
    assert 0.0 < r["lane_util"] <= 1.0 and r["counters_source"].startswith("profiles/")
    # the counters are tied to the kernel build they were measured on: a profile of another build is reported, not hidden
    assert r["stale"] == (r["counters_build_id"] != r["build_id"]) and len(r["build_id"]) == 16
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["hbm"]["frac"] is None or 0.0 < r["hbm"]["frac"] <= 1.0          # measured HBM bytes: far below the peak
    assert r["w_scored"]["algorithmic_bytes_per_step"] == 13203
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "steps/s" and "sample" in c
    assert c["at_8_threads"]["cores"] <= 8 and c["at_8_threads"]["value"] > 0   # mirror of NUM_META_AGENT = 8 (runner.py:74)
    assert j["value"] > 1e6          # the north-star floor (1M env-steps/s on one MI355X)


def test_rccl_branch_runs_on_one_rank(gpu_device):
    """The RCCL ("nccl") branch of dcmrta_amd/dist.py on real hardware: a one-rank process group (DCM_DIST_FORCE_INIT=1)
    makes bench.py take exactly the N > 1 code path -- init_process_group("nccl", device_id=...), the asynchronous
    all_gather_into_tensor of the episode returns racing the next pass, work.wait() ordering, the gathered-vector
    verification, the barrier and the max / sum reductions over ranks."""
    env = dict(os.environ, DCM_DIST_FORCE_INIT="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29573")
    env.pop("DCM_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--envs", "1024", "--no-cpu-baseline", "--no-lockstep-probe"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["config"]["dist_backend"] == "nccl" and "all-gather" in j["config"]["sharding"]
    assert j["n_gpus"] == 1 and j["value"] > 1e6


def test_config4_strong_scaling_line(gpu_device):
    """bench.py --config 4 (BASELINE configs[3]: 65 536 envs x 50A/200T sharded over the ranks, strong scaling), here two
    ranks on the one GPU over gloo with a reduced env count: contiguous shards, gathered returns verified on every rank."""
    env = dict(os.environ, DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29575", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "2", "--warmup", "1",
           "--envs", "1001"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["agents"] == 50 and j["config"]["tasks"] == 200
    assert j["config"]["envs_total"] == 1001 and j["config"]["envs_per_gpu"] == 501      # uneven shards: 501 + 500
    per_pass = j["value"] * j["ms_per_step"] / 1e3
    assert 1001 * 200 < per_pass < 1001 * 500                                             # ~320 decisions per episode


def test_config5_replay_line(gpu_device):
    """bench.py --config 5 (BASELINE configs[4]: 100A/500T route replay with dynamic task arrivals, sharded over the ranks), two
    ranks on the one GPU over gloo with a reduced env count; and the N = 1 line with the oracle's replay as cpu_baseline."""
    env = dict(os.environ, DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "2", "--warmup", "1",
           "--envs", "301"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["agents"] == 100 and j["config"]["tasks"] == 500
    assert j["config"]["envs_total"] == 301 and j["config"]["envs_per_gpu"] == 151 and j["config"]["visibility"] == [20, 20, 10, 100]
    assert j["roofline"]["kernel"] == "k_replay_fast" and "agent_step" in j["config"]["step_definition"]
    per_pass = j["value"] * j["ms_per_step"] / 1e3
    assert 301 * 100 < per_pass < 301 * 2000          # every agent takes a few steps (routes over the 100 visible tasks + depot)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--envs", "256", "--steps", "2",
                          "--warmup", "1", "--visibility", "100,100,10,500"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert "GENERALISED" in j["config"]["workload"] and j["config"]["visibility"] == [100, 100, 10, 500]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "execute_by_route" in c["sample"]


@pytest.mark.parametrize("extra,check", [
    (["--envs", "512"], lambda j: j["scaling"] == "weak" and j["config"]["envs_per_gpu"] == 512 and j["config"]["envs_total"] == 1024),
    (["--config", "4", "--envs", "600"], lambda j: j["scaling"] == "strong" and j["config"]["envs_per_gpu"] == 300 and j["config"]["tasks"] == 200),
    (["--config", "5", "--envs", "200"], lambda j: j["scaling"] == "strong" and j["config"]["envs_per_gpu"] == 100 and j["roofline"]["kernel"] == "k_replay_fast"),
], ids=["config2", "config4", "config5"])
def test_plain_command_launches_its_own_ranks(gpu_device, extra, check):
    """`python bench.py --gpus 2 ...` exactly as the driver types it for N = 1, with NO torch.distributed.run around it: the
    parent starts the two ranks itself before touching the GPU (dcmrta_amd/launch.py), rank 0's single line comes back through
    it.  Two ranks share the test box's one GPU over gloo; on a multi-GPU node the same command runs one rank per GPU over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0
    assert j["config"]["world"] == 2 and j["config"]["process_group_ranks"] == 2 and j["config"]["dist_backend"] == "gloo"
    assert j["config"]["self_launched"] is True and check(j)
    assert "cpu_baseline" not in j
    # every rank compared its own block with the oracle after the timed region; the counts are summed over the group
    assert j["parity"]["mismatches"] == 0 and j["parity"]["envs_checked"] == j["config"]["envs_total"]
    assert len(j["config"]["rank_devices"]) == 2


def test_plain_command_propagates_rank_failure(gpu_device):
    """A rank that dies takes the job down with a non-zero exit code and no JSON line (here: an impossible shape)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--agents", "4000"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_default_line_times_the_other_baseline_configs(gpu_device):
    """The full default run (what the driver clocks) also times the per-GPU shards of BASELINE configs[3] and configs[4] and the
    HBM-bound lockstep kernel, after the timed region."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[0])
    oc = j["other_configs"]
    assert oc["config4_shard"]["kernel"] == "k_rollout_fast_mc" and oc["config4_shard"]["value"] > 1e8
    assert 8192 * 200 < oc["config4_shard"]["steps_per_pass"] < 8192 * 500           # ~320 decisions per 50A/200T episode
    assert oc["config5_shard"]["kernel"] == "k_replay_fast" and oc["config5_shard"]["value"] > 1e8
    assert oc["midsize_70A130T"]["kernel"] == "k_rollout_fast_g" and oc["midsize_70A130T"]["value"] > 1e8
    assert 4096 * 3 * 200 < oc["midsize_70A130T"]["steps_per_pass"] < 4096 * 3 * 500   # ~296 decisions per 70A/130T episode
    lk = j["lockstep_kernel"]
    assert lk["kernel"] == "k_step_fast" and 0.2 < lk["frac"] < 1.2 and (lk["traffic_frac"] is None or lk["traffic_frac"] < lk["frac"])
    assert 0.0 < lk["empty_event_pair_us"] < 0.5 * 1e3 * lk["median_launch_ms"]      # the timing brackets' own cost, reported beside it
    assert j["cpu_baseline"]["value"] > 0 and j["value"] > 1e6
    # bench-size oracle parity of the headline workload and of every shard; each shard with its own roofline and CPU baseline
    assert j["parity"]["envs_checked"] == 4096 and j["parity"]["mismatches"] == 0
    for name, n_envs in (("config4_shard", 8192), ("config5_shard", 4096), ("midsize_70A130T", 4096)):
        e = oc[name]
        assert e["parity"]["envs_checked"] == n_envs and e["parity"]["mismatches"] == 0, name
        assert e["cpu_baseline"]["value"] > 0 and e["cpu_baseline"]["cores"] >= 1 and e["cpu_baseline"]["kind"] == "port", name
        assert "frac" in e["roofline"] and "stale" in e["roofline"] or e["roofline"].get("note"), name
    c3 = oc["config3"]
    assert c3["steps_per_s_end_to_end"] > 1e4 and 0.5 < c3["policy_share"] < 1.0 and c3["env_ms_per_batched_step"] > 0
    assert "policy-bound" in c3["note"] and c3["policy_dtype"] == "fp32"
    lim = j["config"]["limits"]
    assert lim["members_per_task"] == 5 and lim["members_per_task_wide_handle"] == 16 and lim["A"] == 128 and lim["T"] == 1023
    assert j["config"]["rank_devices"] and j["config"]["rank_devices"][0].startswith("cuda:")
    # the compact summary is the LAST key and fits the 2000-character tail the driver keeps; it repeats the numbers a reader needs
    assert out.returncode == 0
    assert list(j)[-1] == "summary" and list(j)[-4:] == ["roofline", "parity", "cpu_baseline", "summary"]
    sm = j["summary"]
    assert len(json.dumps(sm)) < 1500 and lines[0].rstrip().endswith(json.dumps(sm) + "}")
    assert sm["parity"] == {"envs_checked": 4096, "mismatches": 0} and sm["value"] > 1e6 and "frac" in sm["roofline"]
    assert sm["lockstep_kernel"]["frac"] > 0 and sm["lockstep_kernel"].get("steady_state_us", 1) > 0
    for name in ("config4_shard", "config5_shard", "midsize_70A130T"):
        assert sm[name]["value"] > 1e8 and sm[name]["parity_mismatches"] == 0 and sm[name]["cpu"] > 0, name
    assert sm["config3"]["value"] > 1e4
    assert 0 < j["ms_per_step_min"] <= j["ms_per_step_max"]


def test_unprofiled_shape_borrows_the_counters_of_its_kernel(gpu_device):
    """A shape without a committed PMC profile is priced with the instruction counts of the profiled shape that runs the SAME kernel,
    and the line says so (`roofline.counters_shape`): the mid-size class borrows 70A/130T's, a one-chunk training shape 20A/50T's."""
    for extra, kernel, shape in ((["--agents", "65", "--tasks", "65"], "k_rollout_fast_g", "70A130T"),
                                 (["--agents", "12", "--tasks", "30"], "k_rollout_fast", "20A50T")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "512", "--steps", "2", "--warmup", "1", "--streams", "1",
                              "--no-cpu-baseline", "--no-lockstep-probe"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and len(lines) == 1, out.stdout[-1500:] + out.stderr[-2500:]
        r = json.loads(lines[0])["roofline"]
        assert r["kernel"] == kernel and r["counters_shape"] == shape and r["frac"] is not None and r["frac"] > 0
