#!/usr/bin/env python3
"""Digests of the planner-input yaml files the reference ships for its test set (testSet_20A_50T_CONDET/env_i/*.yaml,
written by TestSetGenerator.py), used to check dcmrta_amd/ctasd_io.py.  Build container only."""
import hashlib
import json
import os

import yaml

REF = os.environ.get("DCMRTA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def canon(doc):
    return hashlib.sha256(json.dumps(doc, sort_keys=True).encode()).hexdigest()


import numpy as np  # noqa: E402

out = {}
graph_num = {}
for i in (0, 7, 23, 49):
    d = f"{REF}/testSet_20A_50T_CONDET/env_{i}"
    docs = {n: yaml.safe_load(open(f"{d}/{n}.yaml")) for n in ("vehicle_param", "task_param", "planner_param", "graph")}
    g = docs["graph"]["vehicle0"]
    edges = [k for k in g if k.startswith("edge")]
    nodes = [k for k in g if k.startswith("node")]
    # the shipped distances come from an older Python's math.hypot (libm): last-ulp differences vs today's are expected,
    # so the graph is stored numerically (structure exact, weights compared with a tolerance) instead of as a hash
    graph_num[f"ends_{i}"] = np.array([[g[k][0], g[k][1], g[k][2], g[k][4]] for k in edges], dtype=np.int32)
    graph_num[f"dist_{i}"] = np.array([g[k][3] for k in edges], dtype=np.float64)
    graph_num[f"time_{i}"] = np.array([g[k][5] for k in edges], dtype=np.float64)
    graph_num[f"node_{i}"] = np.array([g[k] for k in nodes], dtype=np.float64)
    out[str(i)] = dict(sha256={n: canon(v) for n, v in docs.items() if n != "graph"}, planner_param=docs["planner_param"],
                       graph_keys_sha256=canon(list(g.keys())))
json.dump(out, open(f"{OUT}/ctasd_yaml_digest.json", "w"), indent=1)
np.savez_compressed(f"{OUT}/ctasd_graph.npz", **graph_num)
print({k: v["graph_keys_sha256"][:12] for k, v in out.items()})
