#!/usr/bin/env python3
"""Launch-ordered record of ONE training step of the headline configuration: every launch group with its HIP-event time,
FLOPs and bytes, as JSON lines -- for lining up two library variants (IO_LIB_PATH) launch by launch.
usage: python tools/seq_launch.py [fp32|bf16] [pairs] > gpurun_out/seq_<variant>.jsonl ;
       python tools/seq_launch.py --diff a.jsonl b.jsonl [class-substring]"""
import json
import os
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--diff":
    A = [json.loads(l) for l in open(sys.argv[2])]
    Bv = [json.loads(l) for l in open(sys.argv[3])]
    flt = sys.argv[4] if len(sys.argv) > 4 else "conv_nt"
    A = [r for r in A if flt in r["name"]]
    Bv = [r for r in Bv if flt in r["name"]]
    assert len(A) == len(Bv), (len(A), len(Bv))
    ta = tb = 0.0
    for i, (a, b) in enumerate(zip(A, Bv)):
        ta += a["ms"]; tb += b["ms"]
        flag = "  <<<" if abs(a["ms"] - b["ms"]) > 0.05 * max(a["ms"], b["ms"]) and abs(a["ms"] - b["ms"]) > 0.03 else ""
        print("%4d %-28s GF %8.1f  MB %8.1f | %7.3f %7.3f  %+7.3f%s" % (i, a["name"], a["flops"] / 1e9, a["bytes"] / 1e6,
                                                                  a["ms"], b["ms"], b["ms"] - a["ms"], flag))
    print("total %.2f %.2f" % (ta, tb))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import instaorder_amd as ia
from instaorder_amd import engine, synthetic

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = 256
cfg = dict(algo="InstaOrderNet_o", lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
           backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True, dtype=dtype)
m = ia.InstaOrderNet_o(cfg, dist_model=False)
m._use_graph = False
m.switch_to("train")
base = synthetic.make_pair_batch(1000, 32, S)
dev = {k: torch.from_numpy(np.concatenate([v] * (B // 32), 0)).cuda() for k, v in base.items()}
acc = None
REP = 3
for it in range(2 + REP):
    if it >= 2:
        torch.cuda.synchronize()
        engine.prof_begin()
    m.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["occ_order"])
    m.step()
    if it >= 2:
        torch.cuda.synchronize()
        recs = engine.prof_launches()
        engine.prof_end()
        if acc is None:
            acc = [[n, ms, fl, by] for n, ms, fl, by in recs]
        else:
            for a, r in zip(acc, recs):
                a[1] = min(a[1], r[1])
for n, ms, fl, by in acc:
    print(json.dumps(dict(name=n, ms=ms, flops=fl, bytes=by)))
