#!/usr/bin/env python3
"""Golden traces of a policy that does NOT respect the mask (run in the build container only).

    python tests/golden/make_golden_masked.py  ->  tests/golden/trace_<A>A<T>T_anymask_s<seed>.npz, overflow_*.npz

TaskEnv.step never looks at the mask (env/task_env.py:326-342) and worker.py:140's argmax can return a masked index, so the
reference SIMULATES an action on a masked task: vacancy = status <= 0 sends the leader alone, the task lists one more member
(possibly more than it requires, possibly after it is over -- then the agent is released at time_finish, i.e. in the past,
and the event time steps backwards).  Same harness as make_golden.py (the reference env driven through the loop of
worker.py:45-87 with the keyed choice protocol), policy "anymask": of the 32-bit draw r of protocol slot 1,
r % 16 == 1 -> the depot whatever the mask says, r % 4 == 0 -> task (r >> 4) % T, else a uniformly random VALID action --
mirrored by ORC_POLICY_ANY in oracle/dcmrta_oracle.c.

trace_*_anymask_*  episodes in which no task ever lists more than 5 members (DCM_MAX_MEMBERS: the HIP env simulates those)
overflow_*         episodes in which one does; `overflow_step` = index of the decision whose agent_step makes a 6th member
Only numbers are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import TaskEnv, below, draw  # noqa: E402

CAP = 3000          # decisions; longer episodes are skipped (time can step backwards: an episode may take long)


class _TooLong(Exception):
    pass


def rollout_anymask(env, seed_e):
    T = env.tasks_num
    st = dict(n=0, max_members=0, overflow_step=-1)
    inner = env.agent_step

    def agent_step(agent_id, task_id):                  # observes len(members) right after every append
        r = inner(agent_id, task_id)
        if task_id - 1 >= 0:
            n = len(env.task_dic[task_id - 1]["members"])
            st["max_members"] = max(st["max_members"], n)
            if n > 5 and st["overflow_step"] < 0:
                st["overflow_step"] = st["n"] - 1
        return r
    env.agent_step = agent_step

    def policy(env_, mask, leader, seed, d):
        st["n"] += 1
        if st["n"] > CAP:
            raise _TooLong()
        r = draw(seed, d, 1)
        if r % 16 == 1:
            return 0
        if r % 4 == 0:
            return 1 + (r >> 4) % T
        valid = np.flatnonzero(~mask)
        return int(valid[below(r, len(valid))])
    tr = mg.rollout(env, seed_e, policy)
    return tr, st


def main():
    wanted = {(5, 8): 3, (10, 20): 3, (20, 50): 4, (13, 37): 2}
    got = {k: 0 for k in wanted}
    overflow = 0
    for (A, T), n_want in wanted.items():
        for s in range(400):
            if got[(A, T)] >= n_want and (overflow >= 3 or (A, T) != (20, 50)):
                break
            env = TaskEnv((A, A), (T, T), 1, 5, seed=3000 + s)
            ia = mg.instance_arrays(env)
            seed_e = mg.env_seed(77, s)
            try:
                tr, st = rollout_anymask(env, seed_e)
            except _TooLong:
                continue
            if int(tr["truncated"]):
                continue
            masked_picks = int(sum(int(tr["mask"][i][a]) for i, a in enumerate(tr["action"])))
            backwards = int((np.diff(tr["now"]) < 0).sum())
            tr.update(ia)
            tr["seed_e"] = np.uint64(seed_e)
            tr["inst_seed"] = np.int64(3000 + s)
            tr["masked_picks"] = np.int64(masked_picks)
            tr["time_steps_backwards"] = np.int64(backwards)
            if st["max_members"] <= 5:
                if got[(A, T)] < n_want and masked_picks >= 3:
                    np.savez_compressed(os.path.join(HERE, f"trace_{A}A{T}T_anymask_s{s}.npz"), **tr)
                    got[(A, T)] += 1
                    print("trace", A, T, s, "steps", int(tr["n_steps"]), "masked picks", masked_picks, "backwards", backwards,
                          "max members", st["max_members"], flush=True)
            elif overflow < 3 and (A, T) == (20, 50):
                tr["overflow_step"] = np.int64(st["overflow_step"])
                np.savez_compressed(os.path.join(HERE, f"overflow_{A}A{T}T_anymask_s{s}.npz"), **tr)
                overflow += 1
                print("overflow", A, T, s, "at decision", st["overflow_step"], "of", int(tr["n_steps"]), flush=True)
    print(got, "overflow cases", overflow)


if __name__ == "__main__":
    main()
