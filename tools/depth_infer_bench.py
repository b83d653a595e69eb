#!/usr/bin/env python3
"""Inference throughput of the MiDaS-based net through the batched driver (inference.infer_depthnet_batched): one
encoder / decoder pass per image, all 190 pairs of a 20-instance image through the order branches.
usage: python tools/depth_infer_bench.py [fp32|bf16] [S] [n_inst]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import instaorder_amd as ia
from instaorder_amd import inference, synthetic

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
n_inst = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = dict(algo="InstaDepthNet_od", lr=1e-5, weight_decay=1e-4, optim="SGD", pretrained_weight=None, use_rgb=True,
           overlap_weight=0.0, distinct_weight=0.0, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=0.0, dtype=dtype)
m = ia.InstaDepthNet_od(cfg, dist_model=False)
m.switch_to("eval")
items = synthetic.make_images(5, 3, n_inst, S)
data = [synthetic.image_mode_inputs(it["image"], it["modal"], S) for it in items]
data = [(torch.from_numpy(r).cuda(), torch.from_numpy(mk).cuda()) for r, mk in data]
for r, mk in data[:1]:
    inference.infer_depthnet_batched(m, r, mk)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    for r, mk in data:
        res = inference.infer_depthnet_batched(m, r, mk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (reps * len(data))
P = n_inst * (n_inst - 1) // 2
print("%s S=%d: %.1f ms per image of %d instances (%d pairs) -> %.0f pairs/s" % (dtype, S, dt * 1e3, n_inst, P, P / dt))
