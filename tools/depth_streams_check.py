#!/usr/bin/env python3
"""InstaDepthNet_od: N training steps from a seeded state; writes the flat parameter buffer (and losses) so that the
single-stream and the multi-stream (IO_DEPTH_STREAMS=1) forms can be compared bit for bit.
usage: python tools/depth_streams_check.py out.npy [steps] [S] [B] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import instaorder_amd as ia
from instaorder_amd import synthetic

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dtype = sys.argv[5] if len(sys.argv) > 5 else "fp32"
torch.manual_seed(1234)
cfg = dict(algo="InstaDepthNet_od", lr=1e-4, weight_decay=1e-4, optim="SGD", pretrained_weight=None, use_rgb=True, dtype=dtype,
           overlap_weight=0.1, distinct_weight=0.9, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=1.0)
m = ia.InstaDepthNet_od(cfg, dist_model=False)
m.switch_to("train")
t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(77, B, S).items()}
losses = []
for _ in range(steps):
    m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
    losses.append(float(m.step()[1]["loss"]))
    if len(losses) == 1:
        torch.cuda.synchronize()
        np.save(out.replace(".npy", "_g1.npy"), m.optim.flat_grads.cpu().numpy())
torch.cuda.synchronize()
np.save(out, m.optim.flat_params.cpu().numpy())
print("losses", losses, "graph", m._graph is not None)
