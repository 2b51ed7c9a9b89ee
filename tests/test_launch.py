"""CPU: the self-launch path of `python bench.py --gpus N` (dcmrta_amd/launch.py) -- the plain command starts its own N ranks
as children before anything touches HIP, relays rank 0's one JSON line and propagates a failing rank's exit code.  The
reference's driver starts its actors itself (driver.py:99, runner.py:74-77); nobody wraps it in a launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "launch_child.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DCM_SELF_LAUNCHED")}


def test_requested_ranks_and_rank_environment():
    from dcmrta_amd import launch
    assert launch.requested_ranks(["--steps", "3", "--gpus", "8"]) == 8
    assert launch.requested_ranks(["--gpus=4", "--config", "5"]) == 4
    assert launch.requested_ranks(["--steps", "3"]) == 1 and launch.requested_ranks(["--gpus"]) == 1
    assert launch.requested_ranks(["--gpus", "x"]) == 1
    assert launch.in_rank_environment({"WORLD_SIZE": "2"}) and launch.in_rank_environment({"RANK": "0"})
    assert not launch.in_rank_environment({"PATH": "/bin"})
    cmd = launch.launcher_command("/x/bench.py", ["--gpus", "2", "--steps", "1"], 2, port=1234)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "2", "--steps", "1"] and "127.0.0.1" in cmd and "1234" in cmd
    # default: the launcher picks and holds its own rendezvous port (no probe-then-bind race between concurrent jobs)
    cmd = launch.launcher_command("/x/bench.py", ["--gpus", "2"], 2)
    assert "--standalone" in cmd and "--master-port" not in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"


def test_launch_module_imports_no_torch():
    code = "import sys; import dcmrta_amd.launch; assert 'torch' not in sys.modules and 'ctypes' not in sys.modules"
    subprocess.run([sys.executable, "-c", code], cwd=ROOT, check=True, timeout=60)


@pytest.mark.parametrize("n", [2, 3])
def test_plain_command_starts_its_own_ranks(n):
    out = subprocess.run([sys.executable, CHILD, "--gpus", str(n), "--envs", "13"], env=_clean_env(), capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["process_group_ranks"] == n and j["envs"] == 13 and j["self_launched"] == str(n)
    assert j["gathered"] == [float(i) for i in range(13)] and j["rank_devices"] == ["cpu"] * n


def test_failing_rank_propagates_exit_code():
    out = subprocess.run([sys.executable, CHILD, "--gpus", "2", "--fail-rank", "1"], env=_clean_env(), capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_inside_a_launcher_no_second_launch():
    """Under torch.distributed.run (RANK / WORLD_SIZE present) the script is a rank, not a launcher."""
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               DCM_DIST_FORCE_INIT="1")
    out = subprocess.run([sys.executable, CHILD, "--gpus", "1"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["process_group_ranks"] == 1 and j["self_launched"] is None


def test_bench_self_launch_happens_before_heavy_imports():
    """bench.py's launcher call sits above its torch / HIP-library imports."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("maybe_self_launch(__file__)") < src.index("import torch") < src.index("from dcmrta_amd import _lib")


def test_hung_rank_ends_the_job_within_the_limit():
    """A rank that never joins the process group: the others fail in bring-up after --dist-timeout with a reason, and if even
    that does not end the job the parent's --launch-timeout ends the child tree it started: non-zero exit, no JSON line, nothing
    left running."""
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, CHILD, "--gpus", "2", "--hang-rank", "1", "--dist-timeout", "600", "--launch-timeout", "25"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    took = time.monotonic() - t0
    assert out.returncode == 124 and took < 90, (out.returncode, took, out.stderr[-2000:])
    assert "did not finish within --launch-timeout" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # the ranks are gone (their command line carried the marker flag)
    time.sleep(1.0)
    left = []
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                cmd = open(f"/proc/{d}/cmdline").read()
            except OSError:
                continue
            if "--hang-rank" in cmd and "launch_child.py" in cmd and "pytest" not in cmd:
                left.append(d)
    assert not left, left


def test_hung_rank_fails_the_others_after_the_dist_timeout():
    """--dist-timeout bounds process-group bring-up: the healthy rank gives up with an error, torch.distributed.run tears the job
    down, the parent exits non-zero well before --launch-timeout."""
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, CHILD, "--gpus", "2", "--hang-rank", "1", "--dist-timeout", "8", "--launch-timeout", "240"],
                         env=_clean_env(), capture_output=True, text=True, timeout=400, cwd=ROOT)
    took = time.monotonic() - t0
    assert out.returncode not in (0, 124) and took < 200, (out.returncode, took, out.stderr[-2000:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_more_ranks_than_devices_is_a_clean_error(monkeypatch):
    """WORLD_SIZE larger than the device count: the preflight (before any other GPU call) raises with a clear message."""
    import torch
    from dcmrta_amd.dist import DistContext
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    for k, v in dict(RANK="3", WORLD_SIZE="4", LOCAL_RANK="3").items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("DCM_FORCE_DEVICE", raising=False)
    with pytest.raises(RuntimeError, match="larger than the device count"):
        DistContext.from_env(expected_world=4)
    # sharing one device on purpose is still possible
    monkeypatch.setenv("DCM_FORCE_DEVICE", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    ctx = DistContext.from_env(expected_world=1)
    assert ctx.world == 1
