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


def _leg_table(aid, route, arrival, depot, task_xy, members, feasible, time_start, time_finish, mwt, velocity):
    """One row per route entry of an agent: where the leg starts / ends, its heading, when the agent left the previous stop
    (`leave`), when it reaches this one (`reach`) and until when it stays (`stay`) -- the quantities env/task_env.py:380-404 derive
    from the final task state."""
    n = len(route)
    stop = np.asarray(route, np.int64)
    xy = np.where((stop >= 0)[:, None], np.asarray(task_xy, np.float64)[np.maximum(stop, 0)], np.asarray(depot, np.float64)[None, :])
    # the stop before entry i; the first leg, and any leg after a depot visit, starts at the depot with "previous decision" 0 (:387,:396)
    before = np.concatenate([[-1], stop[:-1]])
    src = np.where((before >= 0)[:, None], np.asarray(task_xy, np.float64)[np.maximum(before, 0)], np.asarray(depot, np.float64)[None, :])
    reach = np.asarray(arrival, np.float64)
    reach_before = np.concatenate([[0.0], reach[:-1]])
    listed = np.array([s >= 0 and aid in members[s] for s in stop], bool)
    listed_before = np.array([s >= 0 and aid in members[s] for s in before], bool)
    feas = np.asarray(feasible, bool)
    ts, tf = np.asarray(time_start, np.float64), np.asarray(time_finish, np.float64)
    k, kb = np.maximum(stop, 0), np.maximum(before, 0)
    works = listed & feas[k] & (stop >= 0)                                                  # :388
    stay = np.where(works & (ts[k] - reach <= mwt), tf[k], reach + mwt)                     # :389-395
    worked_before = listed_before & feas[kb] & (ts[kb] - reach_before <= mwt)               # :399
    leave = np.where(before < 0, 0.0, np.where(worked_before, tf[kb], reach_before + mwt))  # :396-404
    heading = np.arctan2(xy[:, 1] - src[:, 1], xy[:, 0] - src[:, 0])                        # :382-383
    duration = np.array([np.linalg.norm(src[i] - xy[i]) for i in range(n)], np.float64) / velocity   # :384-385
    return src, xy, heading, duration, leave, reach, stay


def trajectories(routes, depot, task_xy, members, feasible, time_start, time_finish, current_time, max_waiting_time=10.0,
                 dt=0.1, velocity=0.2):
    """generate_traj (env/task_env.py:375-418) as a pure function of an episode's final state, bit-equal with the reference.

    routes: per agent (route, arrival_time) with task ids, -1 = depot; members: per task the final member id list.
    Returns one float64 array [n_samples, 3] = (x, y, heading) per agent.

    The reference advances one clock per agent (`time_step += dt`) through all legs of the route; sample k therefore sits at the
    k-fold running sum of dt -- np.cumsum reproduces that accumulation exactly -- and leg i owns the samples from where leg i-1
    stopped up to the first clock value >= its `stay` time.  Inside a leg the samples before `reach` interpolate between the
    two stops (same expression, evaluated on the whole slice at once), the others sit on the stop."""
    depot = np.asarray(depot, np.float64)
    horizon = float(current_time)
    for route, arrival in routes:
        if len(arrival):
            horizon = max(horizon, float(np.max(arrival)) + float(max_waiting_time), float(np.max(time_finish)) if len(time_finish) else 0.0)
    n_clock = int(horizon / dt) + 16
    clock = np.concatenate([[0.0], np.cumsum(np.full(n_clock, dt, np.float64))])            # clock[m]: the value after m increments
    out = []
    for aid, (route, arrival) in enumerate(routes):
        rows = []
        done = 0                                                                           # increments taken so far
        heading_last = 0.0
        if len(route):
            src, dst, heading, duration, leave, reach, stay = _leg_table(aid, route, arrival, depot, task_xy, members, feasible,
                                                                          time_start, time_finish, max_waiting_time, velocity)
            for i in range(len(route)):
                upto = max(done, int(np.searchsorted(clock, stay[i], side="left")))        # first m with clock[m] >= stay (:405)
                t = clock[done + 1:upto + 1]
                frac = (t - leave[i]) / duration[i] if duration[i] != 0 else np.full(t.shape, np.nan)
                moving = t < reach[i]                                                      # :407
                leg = np.empty((len(t), 3), np.float64)
                leg[:, 0] = np.where(moving, src[i, 0] + frac * (dst[i, 0] - src[i, 0]), dst[i, 0])   # :409-411 / :413
                leg[:, 1] = np.where(moving, src[i, 1] + frac * (dst[i, 1] - src[i, 1]), dst[i, 1])
                leg[:, 2] = heading[i]
                rows.append(leg)
                done = upto
                heading_last = heading[i]
        upto = max(done, int(np.searchsorted(clock, current_time, side="left")))           # :415-417: parked at the depot afterwards
        tail = np.empty((upto - done, 3), np.float64)
        tail[:, 0], tail[:, 1], tail[:, 2] = depot[0], depot[1], heading_last
        rows.append(tail)
        out.append(np.concatenate(rows, axis=0) if rows else np.zeros((0, 3)))
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
