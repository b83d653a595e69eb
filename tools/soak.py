"""Soak run: 200 optimisation steps over 8 rotating synthetic batches, fp32 and bf16 -- losses stay finite and fall.
usage: python tools/soak.py"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import instaorder_amd as ia
from instaorder_amd import synthetic
for dtype in ("fp32", "bf16"):
    cfg = dict(algo="InstaOrderNet_od", lr=0.01, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls", dtype=dtype,
               backbone_param=dict(in_channels=5, num_classes=[2, 3]), use_rgb=True, overlap_weight=0.1, distinct_weight=0.9)
    m = ia.InstaOrderNet_od(cfg, dist_model=False)
    m.switch_to("train")
    losses = []
    for it in range(200):
        b = synthetic.make_pair_batch(it % 8, 64, 128)     # 8 rotating batches
        t = {k: torch.from_numpy(v) for k, v in b.items()}
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
        out = m.step()
        if it % 20 == 0 or it == 199:
            losses.append(float(out[1]["loss"]))
    print(dtype, " ".join("%.3f" % v for v in losses), "finite params:", bool(torch.isfinite(m.net.flat_params).all()))
