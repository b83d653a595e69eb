#!/usr/bin/env python3
"""Which gradients of the staged MiDaS step differ from the flat step at the SECOND step (WeightPlan active)?"""
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
noplan = os.environ.get("NOPLAN", "0") == "1"
for tag, env in (("staged", {"IO_COMM_OVERLAP": "force", "IO_NO_GRAPH": "1"}), ("flat", {"IO_COMM_OVERLAP": "0", "IO_NO_GRAPH": "1"})):
    for k in ("IO_COMM_OVERLAP", "IO_NO_GRAPH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m, sd, _ = _depthnet("fp32", 64, 2)
    if noplan:
        m._wplan = False
    m.optim.param_groups[0]["lr"] = 0.0       # identical weights in every step: any difference is the step's own
    m.switch_to("train")
    grads = []
    for i in range(3):
        t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(500, 2, 64).items()}
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
        m.step()
        torch.cuda.synchronize()
        grads.append(m.optim.flat_grads.clone())
    runs[tag] = grads
    names = {id(p): n for n, p in m.net.named_parameters()}
    spans = [(names[id(p)], off, k) for p, (off, k) in zip(m.optim._params, m.optim._spans)]
    planned = set(names[i] for i in (m._wplan.entries if m._wplan else {}))
for i in range(3):
    a, b = runs["staged"][i].double(), runs["flat"][i].double()
    print("step %d: staged vs flat %.2e | staged vs staged step0 %.2e | flat vs flat step0 %.2e" % (
        i, float((a - b).norm() / b.norm()), float((a - runs["staged"][0].double()).norm() / b.norm()),
        float((b - runs["flat"][0].double()).norm() / b.norm())))
a, b = runs["staged"][1].double(), runs["flat"][1].double()
bad = []
for n, off, k in spans:
    d = float((a[off:off + k] - b[off:off + k]).norm())
    r = float(b[off:off + k].norm())
    if d > 1e-4 * max(r, 1e-12):
        bad.append((d / max(r, 1e-30), n, n in planned))
print("parameters that differ at step 1: %d of %d" % (len(bad), len(spans)))
for e in sorted(bad, reverse=True)[:25]:
    print("  %.2e %s planned=%s" % e)
dist.destroy_process_group()
