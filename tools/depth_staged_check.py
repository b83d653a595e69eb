#!/usr/bin/env python3
"""Values of the staged MiDaS step (per-stage autograd + bucket all-reduces; eager and captured as one hipGraph per stage)
against the flat step (one backward; eager and one hipGraph), step by step on ONE nccl rank (IO_COMM_OVERLAP=force).
usage: python tools/depth_staged_check.py"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
from instaorder_amd import distributed_utils as du, synthetic
du.dist_init_("pytorch", backend="nccl")
from test_gpu_configs import _depthnet

runs = {}
for tag, env in (("staged_graph", {"IO_COMM_OVERLAP": "force"}), ("staged_eager", {"IO_COMM_OVERLAP": "force", "IO_NO_GRAPH": "1"}),
                 ("flat_graph", {"IO_COMM_OVERLAP": "0"}), ("flat_eager", {"IO_COMM_OVERLAP": "0", "IO_NO_GRAPH": "1"})):
    for k in ("IO_COMM_OVERLAP", "IO_NO_GRAPH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m, sd, _ = _depthnet("fp32", 64, 2)
    m.optim.param_groups[0]["lr"] = float(os.environ.get("CHECK_LR", "0"))    # 0: every step starts from the same weights
    m.switch_to("train")
    snaps = [m.optim.flat_params.clone()]
    grads = []
    for i in range(5):
        t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(500 + i, 2, 64).items()}
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
        logs, out = m.step()
        torch.cuda.synchronize()
        snaps.append(m.optim.flat_params.clone())
        grads.append(m.optim.flat_grads.clone())
    runs[tag] = (snaps, grads, bool(getattr(m, "_dp_graphs", None)), bool(getattr(m, "_graph", None)))
    print(tag, "dp_graphs", runs[tag][2], "graph", runs[tag][3])
ref = "flat_eager"
for tag in runs:
    if tag == ref:
        continue
    for i in range(5):
        a, b, p0 = runs[tag][0][i + 1].double(), runs[ref][0][i + 1].double(), runs[ref][0][i].double()
        ga, gb = runs[tag][1][i].double(), runs[ref][1][i].double()
        print("%-13s step %d: params diff / update %.2e   grads diff / norm %.2e" % (
            tag, i, float((a - b).norm() / max(float((b - p0).norm()), 1e-30)), float((ga - gb).norm() / gb.norm())))
dist.destroy_process_group()
