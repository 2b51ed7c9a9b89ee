"""HIP-graph capture of the policy-in-the-loop step.

One decision of the batched loop -- policy forward on the observation tensors, action selection, dcm_step with the
fused observation of the next decision (worker.py:62-76) -- is a fixed sequence of kernels on fixed buffers, so it is
captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm) and replayed: no per-kernel launch cost, no host work
between the policy and the env.  The host only looks at `active` every `check_every` replays to detect the end of the
batch of episodes (steps on finished envs are no-ops on the device).  The policy itself stays stock PyTorch.
"""
import torch


class GraphedRollout:
    def __init__(self, env, policy, check_every=8, warmup=3):
        """policy(obs) -> int32/int64 actions[B] computed with torch ops on env.device (no host syncs, no new
        persistent allocations); it is captured together with env.step."""
        self.env, self.policy, self.check_every = env, policy, int(check_every)
        self.action = torch.zeros((env.B,), dtype=torch.int32, device=env.device)
        self.graph = None
        self._warmup = warmup

    def _one_step(self, obs):
        self.action.copy_(self.policy(obs).to(torch.int32))
        return self.env.step(self.action)

    def capture(self, seeds):
        env = self.env
        obs = env.reset(seeds)
        s = torch.cuda.Stream(device=env.device)
        s.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(s):                 # warm-up on a side stream, as torch's capture recipe requires
            for _ in range(self._warmup):
                obs = self._one_step(obs)
        torch.cuda.current_stream(env.device).wait_stream(s)
        torch.cuda.synchronize(env.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._one_step(obs)
        return self

    @torch.no_grad()
    def run(self, seeds, max_steps=100000):
        """Play one episode per env; returns (summary[B,8], batched_steps)."""
        env = self.env
        if self.graph is None:
            self.capture(seeds)
        obs = env.reset(seeds)
        n = 0
        while n < max_steps:
            for _ in range(self.check_every):
                self.graph.replay()
            n += self.check_every
            if not bool(obs.active.any()):        # obs tensors are the env's static output buffers
                break
        return env.summary(), n
