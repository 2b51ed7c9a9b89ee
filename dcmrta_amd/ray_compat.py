"""The handful of Ray calls driver.py makes, served by batched HIP runners instead of Ray actors.

The reference's learner never touches the env: it creates `NUM_META_AGENT` actors with `RLRunner.remote(i)`
(driver.py:99), starts work with `meta_agent.job.remote(weights, baseline_weights, episode, agents_num, tasks_num)`
(:117,198), collects it with `ray.wait(jobList, num_returns=N)` + `ray.get(done_id)` (:129-130), evaluates with
`testing.remote(seed=...)` / `set_baseline_weights.remote(w)` (:241-264) and disposes of actors with `ray.kill(a)`
(:237,253,270).  This module offers exactly that call shape, so the loop of driver.py:99-305 runs with two import lines
changed and nothing else:

    from dcmrta_amd import ray_compat as ray          # instead of: import ray
    from dcmrta_amd.ray_compat import RLRunner        # instead of: from runner import RLRunner

An "actor" here is a BatchedRunner bound to one GPU: `RLRunner.remote(i)` places actor i on device `i % n_gpus` and a
`job` plays `n_envs` episodes at once (one reference job = one episode; `episodeNumber * n_envs` keeps the instance
blocks of successive jobs disjoint, dist.shard_range splits a total env budget over the actors).  Calls are deferred:
`.remote()` returns an ObjectRef immediately; the work runs when `wait` / `get` asks for it -- in the caller's thread, one
actor after the other (default: deterministic, and the GPU is kept busy by the batch, not by actor concurrency), or on one
worker thread per actor (`init(concurrent=True)`, for a node with one GPU per actor).  Exceptions raised by the work
surface from `get`, as with Ray.  No Ray features beyond these are provided.
"""
import concurrent.futures as _cf
import threading

import torch

from .dist import shard_range

_CFG = dict(n_envs=256, devices=None, net_factory=None, base_seed=0, concurrent=False, runner_kwargs={}, total_envs=None,
            num_actors=8)
_ACTORS = []


def init(n_envs=None, devices=None, net_factory=None, base_seed=0, concurrent=False, total_envs=None, num_actors=8,
         **runner_kwargs):
    """ray.init() stand-in + configuration of the actors created afterwards.

    n_envs: episodes per `job` of every actor -- or total_envs: env budget per round split over `num_actors` actors by
    dist.shard_range (actor i gets the i-th contiguous share).  devices: list of torch devices (default: all visible
    GPUs); net_factory: () -> policy module (default: the stand-in AttentionNet; pass the reference's own class to keep
    using attention.py).  runner_kwargs go to BatchedRunner (rollout_precision, check_every, ...)."""
    _CFG.update(n_envs=n_envs if n_envs is not None else 256, devices=devices, net_factory=net_factory, base_seed=base_seed,
                concurrent=bool(concurrent), runner_kwargs=dict(runner_kwargs), total_envs=total_envs, num_actors=int(num_actors))


def shutdown():
    for a in list(_ACTORS):
        kill(a)


class ObjectRef:
    """Deferred result of an actor call."""

    def __init__(self, fn, actor):
        self._fn, self._actor = fn, actor
        self._done, self._value, self._exc = False, None, None
        self._future = None
        if actor._executor is not None:
            self._future = actor._executor.submit(fn)

    def _run(self):
        if self._future is not None:
            return
        if not self._done:
            try:
                self._value = self._fn()
            except BaseException as ex:   # delivered by get(), like a Ray task error
                self._exc = ex
            self._done = True

    def done(self):
        return self._future.done() if self._future is not None else self._done

    def result(self):
        if self._future is not None:
            return self._future.result()
        self._run()
        if self._exc is not None:
            raise self._exc
        return self._value


def wait(refs, num_returns=1, timeout=None):
    """ray.wait: (ready, remaining) with len(ready) == num_returns (refs keep their order inside each list)."""
    refs = list(refs)
    if num_returns > len(refs):
        raise ValueError("num_returns cannot be greater than the number of refs")
    futs = [r._future for r in refs if r._future is not None]
    if futs:
        while sum(r.done() for r in refs) < num_returns:
            _cf.wait([f for f in futs if not f.done()], timeout=timeout, return_when=_cf.FIRST_COMPLETED)
            if timeout is not None:
                break
    for r in refs:                     # deferred refs: run in submission order until enough are done
        if sum(x.done() for x in refs) >= num_returns:
            break
        r._run()
    ready = [r for r in refs if r.done()][:num_returns]
    ids = set(map(id, ready))
    return ready, [r for r in refs if id(r) not in ids]


def get(refs):
    """ray.get of one ref or a list of refs; re-raises the exception of a failed call."""
    if isinstance(refs, ObjectRef):
        return refs.result()
    return [r.result() for r in refs]


def kill(actor):
    """ray.kill: release the actor's device memory."""
    actor._kill()


class _Method:
    def __init__(self, actor, name):
        self._actor, self._name = actor, name

    def remote(self, *args, **kwargs):
        actor, name = self._actor, self._name
        if actor._dead:
            raise RuntimeError("actor was killed")
        return ObjectRef(lambda: getattr(actor._obj, name)(*args, **kwargs), actor)


class ActorHandle:
    def __init__(self, obj, concurrent):
        self._obj, self._dead = obj, False
        self._executor = _cf.ThreadPoolExecutor(max_workers=1) if concurrent else None
        _ACTORS.append(self)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        if not callable(getattr(self._obj, name)):
            raise AttributeError(f"{name} is not a method of the actor")
        return _Method(self, name)

    def _kill(self):
        if self._dead:
            return
        self._dead = True
        if self._executor is not None:
            self._executor.shutdown(wait=True)
        close = getattr(self._obj, "close", None)
        if close:
            close()
        if self in _ACTORS:
            _ACTORS.remove(self)


def remote(*dargs, **dkwargs):
    """@ray.remote / @ray.remote(num_cpus=..., num_gpus=...) on a class: adds `.remote(...)` construction (the resource
    arguments are accepted and ignored: placement is one GPU per actor index)."""
    def wrap(cls):
        class _Remote:
            __wrapped__ = cls

            @staticmethod
            def remote(*args, **kwargs):
                return ActorHandle(cls(*args, **kwargs), _CFG["concurrent"])
        _Remote.__name__ = cls.__name__
        return _Remote
    if len(dargs) == 1 and isinstance(dargs[0], type) and not dkwargs:
        return wrap(dargs[0])
    return wrap


_lock = threading.Lock()


class Runner:
    """runner.py:11-71 on top of BatchedRunner: same constructor argument and method names, `job` returning the 9 lists of
    per-decision tensors driver.py:135-164 concatenates and stacks."""

    def __init__(self, metaAgentID):
        from .runner import BatchedRunner
        self.metaAgentID = metaAgentID
        devices = _CFG["devices"]
        if devices is None:
            n = torch.cuda.device_count()
            if n < 1:
                raise RuntimeError("RLRunner needs a HIP device (there is no CPU path)")
            devices = [torch.device("cuda", i) for i in range(n)]
        dev = devices[int(metaAgentID) % len(devices)]
        n_envs = _CFG["n_envs"]
        stride = None
        if _CFG["total_envs"] is not None:
            lo, hi = shard_range(int(_CFG["total_envs"]), int(metaAgentID) % _CFG["num_actors"], _CFG["num_actors"])
            n_envs = hi - lo
            # unequal shards: every actor strides its jobs by the LARGEST shard, so that job e of one actor and job e + 1 of a
            # smaller-sharded one never overlap (driver.py:116-118 hands out one episode number per job)
            stride = -(-int(_CFG["total_envs"]) // _CFG["num_actors"])
        with _lock:
            self._r = BatchedRunner(metaAgentID=metaAgentID, n_envs=n_envs, device=str(dev), net_factory=_CFG["net_factory"],
                                    base_seed=_CFG["base_seed"], episode_stride=stride, **_CFG["runner_kwargs"])
        self.device = self._r.device
        self.localNetwork, self.localBaseline = self._r.localNetwork, self._r.localBaseline

    def get_weights(self):
        return self._r.get_weights()

    def set_weights(self, weights):
        self._r.set_weights(weights)

    def set_baseline_weights(self, weights):
        self._r.set_baseline_weights(weights)

    def job(self, global_weights, baseline_weights, episodeNumber, agents_num, tasks_num):
        return self._r.job(global_weights, baseline_weights, episodeNumber, agents_num, tasks_num, as_lists=True)

    def testing(self, agents_range=(10, 20), tasks_range=(20, 50), seed=None):
        return self._r.testing(agents_range, tasks_range, seed=seed)

    def close(self):
        self._r.close()


RLRunner = remote(num_cpus=1, num_gpus=1)(Runner)
