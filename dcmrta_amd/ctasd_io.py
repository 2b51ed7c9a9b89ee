"""Instance / route I/O in the formats of the reference's external-planner tooling.

export_ctasd_yaml  -- what TestSetGenerator.py:40-116 writes for one instance (vehicle_param / task_param /
                      planner_param / graph yaml of the CTAS-D planner), so external planners can be run on
                      instances of this framework and compared through the replay mode.
read_ctasd_routes  -- baselines/CTAS-D.py:10-33,36-46: results.yaml -> per-agent preset routes
                      (vehicle.vvK.node minus the leading depot), in the form BatchedTaskEnv.load_routes takes.
Host-side Python only (pyyaml); nothing here touches the GPU.
"""
import math
import os
from itertools import permutations

import yaml


def ctasd_documents(depot, task_xy, req, dur, n_agents, planner="TEAMPLANNER_CONDET", solver_time=300.0,
                    folder="testSet", index=0):
    """The four yaml documents as python dicts (TestSetGenerator.py:51-112)."""
    T = len(req)
    coords = [tuple(float(c) for c in xy) for xy in task_xy]
    dist = {i: {j: (0 if i == j else math.hypot(coords[i][0] - coords[j][0], coords[i][1] - coords[j][1]))
                for j in range(T)} for i in range(T)}                                      # :26-38
    depot_d = [math.hypot(float(depot[0]) - coords[j][0], float(depot[1]) - coords[j][1]) for j in range(T)]   # :52
    p = list(permutations(range(T), 2))                                                    # :53
    agent_yaml, graph_yaml = {}, {}

    def vehicle_graph(src, dst):
        g = {f"edge{i}": [t[0], t[1], 0, dist[t[0]][t[1]], 0, float(dist[t[0]][t[1]] / 0.2)] for i, t in enumerate(p)}
        for j in range(T):
            g[f"edge{2 * j + len(p)}"] = [src, j, 0, depot_d[j], 0, depot_d[j] / 0.2]
            g[f"edge{2 * j + len(p) + 1}"] = [j, dst, 0, depot_d[j], 0, depot_d[j] / 0.2]
        for j in range(T):
            g[f"node{j}"] = float(dur[j])
        return g

    if planner == "TEAMPLANNER_CONDET":                                                    # :54-63
        agent_yaml["vehicle0"] = {"engCap": 1e6, "engCost": 0., "capVector": [1.0], "capVar": [0.]}
        graph_yaml["vehicle0"] = vehicle_graph(T, T + 1)
    elif planner == "TEAMPLANNER_DET":                                                     # :64-75
        for a in range(n_agents):
            agent_yaml[f"vehicle{a}"] = {"engCap": 1e6, "engCost": 1., "capVector": [1.0], "capVar": [0.]}
            graph_yaml[f"vehicle{a}"] = vehicle_graph(T + a, T + n_agents + a)
    else:
        raise ValueError(planner)
    task_yaml = {f"task{j}": {"and0": {"or0": {"geq": True, "capId": 0, "capReq": float(req[j]), "capVar": 0.}}}
                 for j in range(T)}                                                        # :77-78
    base = f"./{folder}/env_{index}"
    planner_param = {                                                                      # :83-112
        "flagOptimizeCost": True, "flagTaskComplete": True, "flagSprAddCutToSameType": True,
        "taskCompleteReward": 10000, "timePenalty": 100, "recoursePenalty": 1.0, "taskRiskPenalty": 0.0,
        "LARGETIME": 10000.0, "MAXTIME": 1000.0, "MAXENG": 1E8, "flagSolver": planner, "CcpBeta": 0.95,
        "taskBeta": 0.95, "solverMaxTime": solver_time, "solverIterMaxTime": 50.0, "flagNotUseUnralavant": True,
        "MAXALPHA": 20.0, "taskNum": int(T), "vehNum": 1 if planner == "TEAMPLANNER_CONDET" else int(n_agents),
        "capNum": 1, "vehTypeNum": 1,
        "vehNumPerType": [int(n_agents)] if planner == "TEAMPLANNER_CONDET" else [1] * int(n_agents),
        "sampleNum": 500, "randomType": 0, "capType": [0],
        "vehicleParamFile": f"{base}/vehicle_param.yaml", "taskParamFile": f"{base}/task_param.yaml",
        "graphFile": f"{base}/graph.yaml"}
    return {"vehicle_param": agent_yaml, "task_param": task_yaml, "planner_param": planner_param, "graph": graph_yaml}


def export_ctasd_yaml(out_dir, depot, task_xy, req, dur, n_agents, **kw):
    """Write vehicle_param.yaml, task_param.yaml, planner_param.yaml, graph.yaml into out_dir (TestSetGenerator.py:79-116)."""
    os.makedirs(out_dir, exist_ok=True)
    docs = ctasd_documents(depot, task_xy, req, dur, n_agents, **kw)
    for name, doc in docs.items():
        with open(os.path.join(out_dir, name + ".yaml"), "w") as f:
            yaml.dump(doc, f, sort_keys=False)
    return docs


def read_ctasd_routes(result_file, param_file):
    """baselines/CTAS-D.py:10-46 -> list over agents of action lists (None = pre_set_route stays None), or None when
    the planner found no solution."""
    with open(param_file) as f:
        param = yaml.safe_load(f)
    num_veh = param["vehNum"] if param["flagSolver"] == "TEAMPLANNER_DET" else param["vehNumPerType"][0]   # :15-18
    with open(result_file) as f:
        data = yaml.safe_load(f)
    if "vehicle" not in data:                                                              # :22-23
        return None
    nodes = [data["vehicle"][f"vv{i + 1}"]["node"] for i in range(num_veh) if f"vv{i + 1}" in data["vehicle"]]   # :26-31
    return [None if r == [0] else list(r)[1:] for r in nodes]                              # :41-45
