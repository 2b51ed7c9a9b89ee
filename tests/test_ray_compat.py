"""CPU: the Ray call shapes of dcmrta_amd.ray_compat (driver.py:99,117,129-130,233-272) on a plain Python class -- deferred
execution, wait / get semantics, error delivery, kill; the GPU-backed RLRunner itself is exercised by
tests/test_gpu_runner.py::test_actor_results_are_consumable_by_a_learner."""
import pytest


def _counter_cls(ray):
    @ray.remote(num_cpus=1, num_gpus=0.125)
    class Counter:
        log = []

        def __init__(self, metaAgentID):
            self.id, self.closed = metaAgentID, False

        def job(self, w, bw, episode, agents_num, tasks_num):
            type(self).log.append((self.id, episode))
            return [episode], {"makespan": float(episode)}, {"id": self.id, "episode_number": episode}

        def testing(self, seed=None):
            return -float(seed)

        def boom(self):
            raise ValueError("task failed")

        def close(self):
            self.closed = True
    return Counter


@pytest.mark.parametrize("concurrent", [False, True])
def test_remote_wait_get_kill(concurrent):
    from dcmrta_amd import ray_compat as ray
    ray.init(concurrent=concurrent)
    Counter = _counter_cls(ray)
    Counter.__wrapped__.log.clear()
    actors = [Counter.remote(i) for i in range(4)]                       # driver.py:99
    jobs = [a.job.remote({}, {}, ep, 12, 23) for ep, a in enumerate(actors)]   # :116-118
    if not concurrent:
        assert Counter.__wrapped__.log == []                             # deferred: nothing has run yet
    done, rest = ray.wait(jobs, num_returns=2)                           # :129 (partial wait)
    assert len(done) == 2 and len(rest) == 2 and all(d.done() for d in done)
    done2, rest2 = ray.wait(rest, num_returns=2)
    assert rest2 == [] and len(done2) == 2
    results = ray.get(done + done2)                                      # :130
    assert sorted(r[2]["episode_number"] for r in results) == [0, 1, 2, 3]
    assert sorted(Counter.__wrapped__.log) == [(i, i) for i in range(4)]
    assert ray.get(actors[1].testing.remote(seed=7)) == -7.0             # :245-248 single ref
    with pytest.raises(ValueError, match="task failed"):                 # errors surface from get, like Ray task errors
        ray.get(actors[0].boom.remote())
    with pytest.raises(ValueError):
        ray.wait(jobs, num_returns=5)
    obj = actors[2]._obj
    ray.kill(actors[2])                                                  # :237
    assert obj.closed
    with pytest.raises(RuntimeError, match="killed"):
        actors[2].job.remote({}, {}, 9, 1, 1)
    with pytest.raises(AttributeError):
        actors[0].id                                                     # only methods are remote-callable
    ray.shutdown()
    assert all(a._dead for a in actors)


def test_rlrunner_needs_a_gpu():
    """No CPU path: constructing the runner-level actor without a HIP device fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dcmrta_amd import ray_compat as ray
    ray.init()
    with pytest.raises(RuntimeError, match="HIP device"):
        ray.RLRunner.remote(0)
