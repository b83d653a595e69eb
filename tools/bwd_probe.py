#!/usr/bin/env python3
"""Per-tensor backward error of the HIP path against an fp64 evaluation by the CPU oracle, in backward order, from the
well-conditioned state of tests/golden/backward_*.npz (debugging aid).  usage: python tools/bwd_probe.py <tag> <algo>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import load_golden, oracle_gradients_fp32_fp64, prestepped_oracle_state, orc, ALGO_CLASSES, ALGO_LR
import instaorder_amd as ia

tag, algo = sys.argv[1], sys.argv[2]
g = load_golden(tag)
state, batch, exact = prestepped_oracle_state(g, algo)
l32, g32, l64, g64 = oracle_gradients_fp32_fp64(state, batch, algo)
cfg = dict(algo=algo, lr=ALGO_LR[algo], weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
           backbone_param=dict(in_channels=5, num_classes=ALGO_CLASSES[algo]), use_rgb=True, overlap_weight=0.1,
           distinct_weight=0.9)
m = getattr(ia, algo)(cfg, dist_model=False)
m.model.load_state_dict({"module." + k: v.clone() for k, v in state.items()}, strict=True)
m.switch_to("train")
m.optim.param_groups[0]["lr"] = 0.0
t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
if algo == "InstaOrderNet_od":
    m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
else:
    m.set_input(t["rgb"], t["modal1"], t["modal2"], t["occ_order"])
m.step()
names = orc.param_names(state)
grads = {n: p.grad.detach().cpu() for n, p in zip(names, m.net.parameters())}
print("exact rebuild:", exact, " lib:", os.environ.get("IO_LIB_PATH", "default"))
for n in reversed(names):
    ref = g64[n].double()
    den = float(ref.norm().clamp_min(1e-300))
    e = float((grads[n].double() - ref).norm()) / den
    ec = float((g32[n].double() - ref).norm()) / den
    print("%-34s hip %.2e  cpu %.2e %s" % (n, e, ec, "  <<<" if e > 3 * ec + 2e-5 else ""))
