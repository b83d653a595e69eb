#!/usr/bin/env python3
"""Per-launch roofline table of ONE training step of the headline configuration (InstaOrderNet_o, 256 pairs, 256x256,
fp32 | bf16): every launch group with its HIP-event time, algorithmic FLOPs / bytes, the time its roofs would allow
(MFMA at the practical ceiling, HBM at the streaming ceiling) and what it loses against them -- sorted by loss.
usage: python tools/per_launch.py [fp32|bf16] [pairs] [full] > profiles/rNN_per_launch_<dtype>.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import instaorder_amd as ia
from instaorder_amd import engine, synthetic

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = 256
MFMA = 139e12 if dtype == "fp32" else 2.4e15        # practical ceilings (DESIGN.md "Measured ceilings"; guide: 95 % of 2.5 PF)
HBM = 5.9e12                                         # mixed read / write streaming on this chip
cfg = dict(algo="InstaOrderNet_o", lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
           backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True, dtype=dtype)
m = ia.InstaOrderNet_o(cfg, dist_model=False)
m._use_graph = False
m.switch_to("train")
base = synthetic.make_pair_batch(1000, 32, S)
dev = {k: torch.from_numpy(np.concatenate([v] * (B // 32), 0)).cuda() for k, v in base.items()}
for it in range(3):
    if it == 2:
        torch.cuda.synchronize()
        engine.prof_begin()
    m.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["occ_order"])
    m.step()
torch.cuda.synchronize()
recs = engine.prof_launches()
engine.prof_end()
tot = sum(r[1] for r in recs)
rows = []
for i, (name, ms, fl, by) in enumerate(recs):
    ideal = max(fl / MFMA, by / HBM) * 1e3
    rows.append((ms - ideal, i, name, ms, fl, by, ideal))
print("# one step: %d launch groups, %.2f ms; ceilings: MFMA %.0f TF/s, HBM %.1f TB/s" % (len(recs), tot, MFMA / 1e12, HBM / 1e12))
print("# %4s %-30s %8s %8s %8s %8s %8s  bound" % ("seq", "class", "ms", "ideal", "lost", "TF/s", "TB/s"))
for lost, i, name, ms, fl, by, ideal in sorted(rows, reverse=True)[:70]:
    print("  %4d %-30s %8.3f %8.3f %8.3f %8.1f %8.2f  %s" % (i, name, ms, ideal, lost, fl / ms / 1e9, by / ms / 1e9,
                                                          "mfma" if fl / MFMA > by / HBM else "hbm"))
if len(sys.argv) > 3 and sys.argv[3] == "full":      # every launch group in execution order, with its work
    print("# full sequence: seq class ms ideal GFLOP MB")
    for lost, i, name, ms, fl, by, ideal in sorted(rows, key=lambda r: r[1]):
        print("S %4d %-30s %8.3f %8.3f %9.2f %9.1f" % (i, name, ms, ideal, fl / 1e9, by / 1e6))
agg = {}
for lost, i, name, ms, fl, by, ideal in rows:
    a = agg.setdefault(name, [0.0, 0.0, 0])
    a[0] += ms; a[1] += ideal; a[2] += 1
print("# per class: ms, ideal ms, launches")
for k, (ms, ideal, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("# %-30s %8.2f %8.2f %5d" % (k, ms, ideal, n))
