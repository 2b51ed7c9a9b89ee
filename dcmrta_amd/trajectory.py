"""Route / result export: the data the reference's visualisation and CSV tooling starts from.

routes_to_yaml      -- Worker.generate_route (worker.py:244-251): {agent: [task id + 1, ...]} (0 = depot), yaml
route_history       -- per-agent (route, arrival_time) lists of one env, as env/task_env.py:95-96 stores them (the input of
                       generate_traj, env/task_env.py:375-418)
write_results_csv   -- RL_test.py:31,45-51 / baselines/CTAS-D.py:56,96: one row of the six perf metrics per instance
"""
import csv

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
