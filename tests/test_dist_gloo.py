"""CPU, world_size=2, gloo: the N>1 path of bench.py / the runner -- contiguous env shards, no data-path
collective, one all-gather of per-env episode returns.  Returns are produced by the oracle here
(the HIP env cannot run without a GPU); what is under test is the sharding + gather logic in dcmrta_amd/dist.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.dist import DistContext, shard_range
    from dcmrta_amd.instances import generate_batch
    ctx = DistContext.from_env(expected_world=world, backend="gloo", device=torch.device("cpu"))
    lo, hi = shard_range(n_total, rank, world)
    inst = generate_batch(hi - lo, 5, 8, base_seed=11, first=lo)
    seeds = env_seeds(3, lo, hi - lo)
    _, reward, steps, _ = oracle.batch_rollout(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], seeds, 5)
    gathered = ctx.all_gather_returns(torch.from_numpy(reward), n_total=n_total)    # uneven blocks are padded and trimmed
    ctx.verify_gather(gathered, torch.from_numpy(reward), lo)                       # own block in place + all ranks agree
    bad = gathered.clone()
    bad[(hi + 1) % n_total] += 1.0 if rank == 0 else 0.0     # rank 0 holds a different vector (outside its own block)
    mismatch_detected = 0
    for wrong in (bad, gathered.roll(1) if rank == 1 else gathered):                 # ... and rank 1 a vector without its own block
        try:
            ctx.verify_gather(wrong, torch.from_numpy(reward), lo)
        except RuntimeError:
            mismatch_detected += 1        # raised on BOTH ranks each time: nobody is left waiting in a collective
    total = ctx.sum_over_ranks(int(steps.sum()))
    tmax = ctx.max_over_ranks(float(rank + 1))
    ctx.barrier()
    q.put((rank, gathered.numpy().copy(), total, tmax, (lo, hi), mismatch_detected))
    ctx.shutdown()


def test_shard_range_partitions():
    from dcmrta_amd.dist import shard_range
    for n in (1, 7, 8, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_total", [12, 13])
def test_all_gather_returns_world2(oracle_lib, n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over the unsharded batch
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(n_total, 5, 8, base_seed=11)
    _, reward, steps, _ = oracle_lib.batch_rollout(inst["depot"], inst["task_xy"], inst["req"], inst["dur"],
                                                   env_seeds(3, 0, n_total), 5)
    for rank, gathered, total, tmax, (lo, hi), mismatch_detected in res:
        assert mismatch_detected == 2, rank                                          # a corrupted vector is refused on every rank
        assert np.array_equal(gathered, reward), rank   # every rank holds the full return vector, rank-major
        assert total == int(steps.sum())
        assert tmax == float(world)
    assert sorted(r[4] for r in res) == ([(0, 6), (6, 12)] if n_total == 12 else [(0, 7), (7, 13)])


def test_gather_checksum_reads_raw_bits():
    """verify_gather's cross-rank checksum is taken over the values' bit patterns: float32 returns in (-1, 1) (which a
    value conversion to int64 would all truncate to 0) and float64 values that differ in the last mantissa bit stay distinct."""
    from dcmrta_amd.dist import _raw_bits
    a = torch.tensor([0.25, -0.5, 0.75], dtype=torch.float32)
    b = torch.tensor([0.25, -0.5, 0.7500001], dtype=torch.float32)
    assert _raw_bits(a).dtype == torch.int64 and not torch.equal(_raw_bits(a), _raw_bits(b)) and int(_raw_bits(a).abs().min()) > 0
    x = torch.tensor([-100.68334319265496], dtype=torch.float64)
    y = torch.from_numpy(np.nextafter(x.numpy(), 0.0))
    assert not torch.equal(_raw_bits(x), _raw_bits(y))
    with pytest.raises(TypeError):
        _raw_bits(torch.zeros(2, dtype=torch.uint8))
