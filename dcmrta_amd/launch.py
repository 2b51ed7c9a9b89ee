"""Self-launch of the one-process-per-GPU job: `python bench.py --gpus N` with no torch.distributed.run around it.

The reference's data parallelism is NUM_META_AGENT isolated actor processes started by the driver itself
(driver.py:99 `RLRunner.remote(i)` for every meta agent, runner.py:74-77); the caller never wraps the driver in a launcher.
Same here: when a script asks for N > 1 ranks and no rank environment (RANK / WORLD_SIZE) is present, the parent process
becomes the launcher -- BEFORE anything initialises HIP (this module imports neither torch nor the HIP library, and the
parent never does afterwards): it starts `python -m torch.distributed.run --nproc-per-node N <script> <argv>` as a CHILD
process (never exec: a process that has touched the GPU must not be replaced, and the parent stays free of GPU state
anyway), relays the children's stdout / stderr unchanged (rank 0 prints the one JSON line) and exits with their code.
"""
import os
import socket
import subprocess
import sys


def requested_ranks(argv, flag="--gpus", default=1):
    """Value of `--gpus N` / `--gpus=N` in argv without building the script's full parser (which imports torch)."""
    n = default
    for i, a in enumerate(argv):
        if a == flag and i + 1 < len(argv):
            n = argv[i + 1]
        elif a.startswith(flag + "="):
            n = a.split("=", 1)[1]
    try:
        return int(n)
    except (TypeError, ValueError):
        return default


def in_rank_environment(environ=None):
    """True when a launcher (torch.distributed.run, the driver, a test) has already set this process up as one rank."""
    environ = os.environ if environ is None else environ
    return "RANK" in environ or "WORLD_SIZE" in environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(script, argv, n, port=None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script] + list(argv)


def launch_ranks(script, argv, n, timeout=None):
    """Start the N ranks as a child process tree and wait.  Returns the exit code (non-zero if any rank failed:
    torch.distributed.run tears the others down and reports it)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL fails without it on this pool
    env.setdefault("GPU_MAX_HW_QUEUES", "8")               # bench.py's side streams (see there)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "1")                 # what torch.distributed.run would set (with a warning) anyway
    env["DCM_SELF_LAUNCHED"] = str(n)                      # reported in the bench line
    proc = subprocess.Popen(launcher_command(script, argv, n), env=env)
    try:
        return proc.wait(timeout=timeout)
    except BaseException:
        proc.terminate()                                    # the exact child started above, nothing else
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise


def maybe_self_launch(script, argv=None, flag="--gpus"):
    """Call first thing in a script's __main__ path (before importing torch).  Returns when this process is a rank (or N = 1);
    otherwise runs the N ranks as children and exits with their code."""
    argv = sys.argv[1:] if argv is None else argv
    n = requested_ranks(argv, flag)
    if n <= 1 or in_rank_environment():
        return
    sys.exit(launch_ranks(os.path.abspath(script), argv, n))
