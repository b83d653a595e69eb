#!/usr/bin/env python3
"""Which torch (ATen) operators one eager InstaDepthNet_od training step still issues between the HIP launches of
instaorder_amd.ops -- copies, fills, adds of the autograd engine -- by count and input shape.
usage: python tools/depth_op_census.py [B] [S] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import instaorder_amd as ia
from instaorder_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
cfg = dict(algo="InstaDepthNet_od", lr=1e-4, weight_decay=1e-4, optim="SGD", pretrained_weight=None, use_rgb=True, dtype=dtype,
           overlap_weight=0.1, distinct_weight=0.9, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=1.0)
m = ia.InstaDepthNet_od(cfg, dist_model=False)
m._use_graph = False
m.switch_to("train")
t = {k: torch.from_numpy(v.copy()).cuda() for k, v in synthetic.make_depth_batch(77, B, S).items()}


def step():
    m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
    m.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages()
print("# ATen operators of one step, by count")
for e in sorted(ka, key=lambda e: -e.count)[:28]:
    print("%-44s %6d" % (e.key[:44], e.count))
print("# copies / clones / fills / adds by input shape")
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
    if e.key in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::contiguous",
                 "aten::cat", "aten::_to_copy") and e.count >= 2:
        print("%-18s %6d  %s" % (e.key, e.count, str(e.input_shapes)[:110]))
