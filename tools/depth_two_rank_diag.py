#!/usr/bin/env python3
"""gradsG of tests/dp_worker_depth.py (two ranks, replays from the reloaded initial state at lr = 0) for the four forms of
the exchange: staged / flat x graphs / eager -- pairwise relative differences.  usage: python tools/depth_two_rank_diag.py"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_configs import _run_two_ranks

forms = {"staged_graph": {}, "staged_eager": {"IO_NO_GRAPH": "1"},
         "flat_graph": {"IO_COMM_OVERLAP": "0"}, "flat_graph2": {"IO_COMM_OVERLAP": "0"},
         "flat_eager": {"IO_COMM_OVERLAP": "0", "IO_NO_GRAPH": "1"},
         "flat_graph_1stream": {"IO_COMM_OVERLAP": "0", "IO_DEPTH_STREAMS": "0"}}
g = {}
with tempfile.TemporaryDirectory() as td:
    for tag, env in forms.items():
        d = os.path.join(td, tag)
        try:
            res = _run_two_ranks(d, env, worker="dp_worker_depth.py")
        except AssertionError as ex:
            import json
            print(tag, "WORKER FAILED:", str(ex).strip().splitlines()[-1])
            try:
                r0 = json.load(open(os.path.join(d, "rank0.json")))
                print("   bad tensors:", r0.get("bad_names"))
            except Exception as e2:
                print("   (no rank0.json: %s)" % e2)
            continue
        g[tag] = {k: np.load(os.path.join(d, "%s_rank0.npy" % k)).astype(np.float64) for k in ("gradsA", "gradsG", "paramsD")}
        print(tag, "staged_graphs", res[0]["staged_graphs"], "overlap", res[0]["overlap"], "losses", res[0]["losses"])
for k in ("gradsA", "gradsG"):
    for a in g:
        for b in g:
            if a < b:
                print("%s %-13s vs %-13s %.2e" % (k, a, b, np.sqrt(((g[a][k] - g[b][k]) ** 2).sum()) / np.sqrt((g[b][k] ** 2).sum())))

