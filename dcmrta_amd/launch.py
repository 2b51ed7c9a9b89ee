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
import signal
import socket
import subprocess
import sys
import time


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


DEFAULT_LAUNCH_TIMEOUT_S = 1500.0        # wall-clock limit of the whole N-rank job (--launch-timeout), below the driver's own 1800 s


def launcher_command(script, argv, n, port=None):
    """port=None (the default): torch.distributed.run picks and HOLDS its own rendezvous port (--standalone: c10d store on
    127.0.0.1:0), so two jobs started at the same moment cannot race for a port number probed here; an explicit port gives the
    static --master-addr / --master-port form."""
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n)]
    if port is None:
        rdzv = ["--standalone", "--local-addr", "127.0.0.1"]
    else:
        rdzv = ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    return head + rdzv + [script] + list(argv)


def _float_flag(argv, flag, default):
    v = default
    for i, a in enumerate(argv):
        if a == flag and i + 1 < len(argv):
            v = argv[i + 1]
        elif a.startswith(flag + "="):
            v = a.split("=", 1)[1]
    try:
        return float(v)
    except (TypeError, ValueError):
        return default


def _descendants(root):
    """PIDs of every live descendant of `root` (children first read from /proc: ppid chains), root excluded."""
    ppid = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                with open(f"/proc/{d}/stat") as f:
                    ppid[int(d)] = int(f.read().rsplit(")", 1)[1].split()[1])
            except (OSError, ValueError, IndexError):
                pass
    out, frontier = [], [root]
    while frontier:
        nxt = [p for p, pp in ppid.items() if pp in frontier and p not in out]
        out += nxt
        frontier = nxt
    return out


def _alive(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except (OSError, IndexError):
        return False


def _stop_tree(proc, grace=10.0):
    """End exactly the process tree this module started: the launcher child and the rank processes below it (found by their
    parent-pid chain BEFORE the first signal; they stay in the caller's process group, so a supervisor that ends the group still
    ends them too) -- SIGTERM, a grace period, then SIGKILL for what is left of those pids."""
    pids = [proc.pid] + _descendants(proc.pid)
    for sig in (signal.SIGTERM, signal.SIGKILL):
        # the launcher child through its Popen handle (a no-op once it has been reaped: its pid may belong to someone else by
        # then); the ranks below it by pid, only while /proc still shows them alive
        if proc.poll() is None:
            try:
                proc.send_signal(sig)
            except (ProcessLookupError, PermissionError):
                pass
        for pid in pids[1:]:
            if _alive(pid):
                try:
                    os.kill(pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass
        try:
            proc.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            continue
        t_end = time.monotonic() + grace
        while time.monotonic() < t_end and any(_alive(p) for p in pids[1:]):
            time.sleep(0.1)
        if not any(_alive(p) for p in pids[1:]):
            return


def launch_ranks(script, argv, n, timeout=None):
    """Start the N ranks as a child process tree and wait at most `timeout` seconds (None, the library default = forever; the
    scripts' `--launch-timeout` applies DEFAULT_LAUNCH_TIMEOUT_S through maybe_self_launch).  Returns the exit code
    (non-zero if any rank failed: torch.distributed.run tears the others down and reports it); on expiry the child tree is
    ended and the exit code is 124 with a one-line reason on stderr.  The ranks are always FRESH children of a parent that
    never touched the GPU -- nothing is re-exec'ed."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL fails without it on this pool
    env.setdefault("GPU_MAX_HW_QUEUES", "8")               # bench.py's side streams (see there)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "1")                 # what torch.distributed.run would set (with a warning) anyway
    env["DCM_SELF_LAUNCHED"] = str(n)                      # reported in the bench line
    proc = subprocess.Popen(launcher_command(script, argv, n), env=env)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        _stop_tree(proc)
        print(f"{os.path.basename(script)}: the {n}-rank job did not finish within --launch-timeout {timeout:g} s "
              f"(a rank hung, e.g. in process-group bring-up): its process tree was ended", file=sys.stderr, flush=True)
        return 124
    except BaseException:
        _stop_tree(proc)
        raise


def maybe_self_launch(script, argv=None, flag="--gpus"):
    """Call first thing in a script's __main__ path (before importing torch).  Returns when this process is a rank (or N = 1);
    otherwise runs the N ranks as children and exits with their code.  `--launch-timeout SECONDS` in argv bounds the job."""
    argv = sys.argv[1:] if argv is None else argv
    n = requested_ranks(argv, flag)
    if n <= 1 or in_rank_environment():
        return
    limit = _float_flag(argv, "--launch-timeout", DEFAULT_LAUNCH_TIMEOUT_S)
    sys.exit(launch_ranks(os.path.abspath(script), argv, n, timeout=limit if limit > 0 else None))
