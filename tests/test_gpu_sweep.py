"""-m gpu: randomised parity sweep (a short run of tools/sweep.py): random shapes, waiting limits and durations; the persistent
kernel in one launch and in randomly budgeted launches, and the lockstep API against the oracle, every terminal quantity bit-exact."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shape_sweep(gpu_device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "60", "16"], capture_output=True, text=True,
                         timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("sweep:")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    assert " 0 mismatches" in line[0] and "MISMATCH" not in out.stdout, out.stdout[-3000:]


def test_random_replay_sweep(gpu_device):
    """tools/sweep_replay.py: random routes (None routes, too few / surplus visitors, shuffled order, reactive or not)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_replay.py"), "60"], capture_output=True, text=True,
                         timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("replay sweep:")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    assert " 0 mismatches" in line[0] and "MISMATCH" not in out.stdout, out.stdout[-3000:]


def test_incremental_task_update_selfcheck(gpu_device):
    """tools/inc_selfcheck.py: a -DDCM_INC_DEBUG build of the kernels dry-runs every task an incremental task_update call skips
    (the chunks not touched by the previous call / without a due wake-up time) and prints a line for each one a full pass would
    have changed.  Five multi-chunk shapes incl. short waiting limits; expected: no such line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "inc_selfcheck.py")], capture_output=True, text=True, timeout=900)
    checked = [l for l in out.stdout.splitlines() if l.startswith("checked ")]
    assert len(checked) == 5, out.stdout[-2000:] + out.stderr[-2000:]
    assert "INC-DEBUG" not in out.stdout, out.stdout[:3000]
