"""Route / result export: the data the reference's visualisation and CSV tooling starts from.

routes_to_yaml      -- Worker.generate_route (worker.py:244-251): {agent: [task id + 1, ...]} (0 = depot), yaml
route_history       -- per-agent (route, arrival_time) lists of one env, as env/task_env.py:95-96 stores them (the input of
                       generate_traj, env/task_env.py:375-418)
write_results_csv   -- RL_test.py:31,45-51 / baselines/CTAS-D.py:56,96: one row of the six perf metrics per instance
trajectories / generate_traj -- env/task_env.py:375-418: positions (x, y, heading) of every agent sampled every dt, the
                       input of the reference's plot_animation
"""
import csv

import numpy as np
import yaml

METRIC_COLUMNS = ("success_rate", "makespan", "time_cost", "waiting_time", "travel_dist", "efficiency")


def route_history(env, b=0):
    task, arrival, length = (x[b].cpu().numpy() for x in env.routes())
    out = []
    for a in range(env.A):
        n = int(length[a])
        if n > task.shape[1]:
            raise ValueError(f"route of agent {a} has {n} entries but the log holds {task.shape[1]}; raise enable_route_log(cap)")
        out.append(([int(t) for t in task[a, :n]], [float(x) for x in arrival[a, :n]]))
    return out


def routes_to_yaml(env, path, b=0):
    routes = {a: [t + 1 for t in r] for a, (r, _) in enumerate(route_history(env, b))}   # worker.py:246-248
    with open(path, "w") as f:
        yaml.dump(routes, f, sort_keys=False)
    return routes


def write_results_csv(path, summary):
    """summary: [N,8] rows of dcm_summary (reward, n_finished, 6 metrics)."""
    rows = summary.cpu().numpy() if hasattr(summary, "cpu") else summary
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(("",) + METRIC_COLUMNS)
        for i, r in enumerate(rows):
            w.writerow([i] + [repr(float(x)) for x in r[2:8]])


def trajectories(routes, depot, task_xy, members, feasible, time_start, time_finish, current_time, max_waiting_time=10.0,
                 dt=0.1, velocity=0.2):
    """generate_traj (env/task_env.py:375-418) as a pure function of an episode's final state.

    routes: per agent (route, arrival_time) with task ids, -1 = depot; members: per task the final member id list.
    Returns one float64 array [n_samples, 3] = (x, y, heading) per agent, sample k at time (k+1)*dt accumulated like
    the reference (time_step += dt)."""
    depot = np.asarray(depot, np.float64)
    loc = lambda t: depot if t == -1 else np.asarray(task_xy[t], np.float64)
    out = []
    for aid, (route, arrival) in enumerate(routes):
        traj = []
        time_step = 0
        angle = 0.0
        for i in range(len(route)):
            # (route[i-1] with i == 0 is route[-1] in the reference, :380; its value is only used through prev_decision,
            # which is 0 for the depot -- restated literally)
            prev_t = route[i - 1] if i > 0 and route[i - 1] != -1 else -1
            cur_t = route[i]
            p, c = loc(prev_t), loc(cur_t)
            angle = np.arctan2(c[1] - p[1], c[0] - p[0])                                   # :382-383
            distance = np.linalg.norm(p - c)                                               # :384
            total_time = distance / velocity                                               # :385
            arr_cur = arrival[i]
            arr_prev = arrival[i - 1] if prev_t != -1 else 0                               # :387
            if cur_t != -1 and aid in members[cur_t] and feasible[cur_t]:                  # :388-393
                next_decision = time_finish[cur_t] if time_start[cur_t] - arr_cur <= max_waiting_time \
                    else arr_cur + max_waiting_time
            else:
                next_decision = arr_cur + max_waiting_time                                 # :394-395
            if prev_t == -1:
                prev_decision = 0                                                          # :396-397
            elif aid in members[prev_t] and time_start[prev_t] - arr_prev <= max_waiting_time and feasible[prev_t]:
                prev_decision = time_finish[prev_t]                                        # :399-402
            else:
                prev_decision = arr_prev + max_waiting_time                                # :403-404
            while time_step < next_decision:                                               # :405-414
                time_step += dt
                if time_step < arr_cur:
                    f = (time_step - prev_decision) / total_time
                    traj.append(np.hstack([p[0] + f * (c[0] - p[0]), p[1] + f * (c[1] - p[1]), angle]))
                else:
                    traj.append(np.array([c[0], c[1], angle]))
        while time_step < current_time:                                                    # :415-417
            time_step += dt
            traj.append(np.array([depot[0], depot[1], angle]))
        out.append(np.array(traj, np.float64).reshape(-1, 3))
    return out


def generate_traj(env, b=0, dt=0.1):
    """trajectories() of env b of a BatchedTaskEnv whose route log was enabled before the episode (enable_route_log)."""
    ts = {k: v[b].cpu().numpy() for k, v in env.tasks_state().items()}
    mem = env.task_members()[b].cpu().numpy()
    members = [[int(x) for x in row if x >= 0] for row in mem]
    d, xy, _, _ = env._instances
    st = env.status()
    return trajectories(route_history(env, b), d[b].cpu().numpy(), xy[b].cpu().numpy(), members, ts["feasible"].astype(bool),
                        ts["time_start"], ts["time_finish"], float(st["now"][b]), env.max_waiting_time, dt)
