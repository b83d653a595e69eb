#!/usr/bin/env python3
"""Order-prediction accuracy parity on a synthetic InstaOrder-val stand-in (SURVEY.md 8(d) "Accuracy").

No dataset or checkpoint exists offline, so "val order accuracy within 0.1 pp of the reference" is measured as
agreement between the HIP path and the CPU oracle (which is pinned to the reference) on seeded synthetic images:
trained-like weights, BN running statistics warmed by train-mode passes, head centred so decisions fall on both
sides of the thresholds, random ground-truth matrices.  Prints recall / precision / F1 (occlusion) and WHDR
(depth) for both paths and their difference in percentage points.

usage: python tools/synthetic_val.py [n_images] [n_inst] [S] [bf16]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import instaorder_amd as ia
from instaorder_amd import inference as infer
from instaorder_amd import synthetic
from oracle import resnet_oracle as orc        # checker only


def run(n_images=20, n_inst=5, S=256, seed=91, verbose=True, dtype="fp32"):
    algo = "InstaOrderNet_od"
    cfg = dict(algo=algo, lr=1e-4, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
               backbone_param=dict(in_channels=5, num_classes=[2, 3]), use_rgb=True, overlap_weight=0.1,
               distinct_weight=0.9, dtype=dtype)
    m = ia.InstaOrderNet_od(cfg, dist_model=False)
    sd = synthetic.make_state_dict(seed, 5, [2, 3], prefix="module.", style="kaiming")
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    state = orc.state_from_numpy(sd, prefix="module.")
    # warm the running statistics identically on both sides (train-mode forwards, no optimiser step)
    m.switch_to("train")
    for it in range(6):
        b = synthetic.make_pair_batch(seed + 1 + it, 8, S)
        x = torch.cat([torch.from_numpy(b["modal1"]), torch.from_numpy(b["modal2"]), torch.from_numpy(b["rgb"])], 1)
        with torch.no_grad():
            m.model(x.cuda())
            orc.resnet_forward(state, x, True)
    m.switch_to("eval")
    items = synthetic.make_images(seed + 50, n_images, n_inst, S)
    pairs = infer.upper_pairs(n_inst)
    # oracle logits for every pair (both directions), then centre the heads on the oracle's medians
    # (eval mode: samples are independent, so the oracle takes the pairs of several images per call -- on a many-core
    # host a 10-sample batch leaves most threads idle and a 200-image run takes 12 minutes instead of 3)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    xs1, xs2 = [], []
    for item in items:
        rgb, masks = synthetic.image_mode_inputs(item["image"], item["modal"], S)
        r = torch.from_numpy(rgb).expand(len(pairs), -1, -1, -1)
        mi = torch.from_numpy(np.stack([masks[i] for i, j in pairs]))[:, None]
        mj = torch.from_numpy(np.stack([masks[j] for i, j in pairs]))[:, None]
        xs1.append(torch.cat([mi, mj, r], 1))
        xs2.append(torch.cat([mj, mi, r], 1))
    per_call = max(1, 48 // max(len(pairs), 1))
    ologits = []
    for i0 in range(0, len(items), per_call):
        x = torch.cat(xs1[i0:i0 + per_call] + xs2[i0:i0 + per_call], 0)
        with torch.no_grad():
            z = torch.cat(orc.resnet_forward(state, x, False), 1)
        k = len(xs1[i0:i0 + per_call])
        z1, z2 = z[:k * len(pairs)], z[k * len(pairs):]
        for q in range(k):
            ologits.append((z1[q * len(pairs):(q + 1) * len(pairs)], z2[q * len(pairs):(q + 1) * len(pairs)]))
    del xs1, xs2
    med = -torch.cat([torch.cat(p, 0) for p in ologits], 0).median(0).values
    state["fc_occ.bias"] += med[:2]
    state["fc_depth.bias"] += med[2:]
    with torch.no_grad():
        m.net.fc_occ.bias += med[:2].cuda()
        m.net.fc_depth.bias += med[2:].cuda()
    acc = {"hip": [], "oracle": []}
    flips = 0
    flip_margins = []          # the oracle's own decision margin (probability units) wherever the two paths disagree
    for item, (z1, z2) in zip(items, ologits):
        z1, z2 = z1 + med, z2 + med
        d = infer.decide(z1, z2, 2, 3)
        marg = infer.decision_margins(torch.cat([z1, z2], 1), algo)
        o_occ, o_dep = orc.order_matrices(n_inst, pairs, d["i_over_j"], d["j_over_i"], d["depth"])
        rgb, masks = synthetic.image_mode_inputs(item["image"], item["modal"], S)
        res = infer.infer_order_batched(m, torch.from_numpy(rgb), torch.from_numpy(masks), method=algo)
        flips += int((res["occ_order"] != o_occ).sum() + (res["depth_order"] != o_dep).sum())
        for k, (i, j) in enumerate(pairs):
            if res["occ_order"][i, j] != o_occ[i, j]:
                flip_margins.append(float(marg["occ"][k, 0]))
            if res["occ_order"][j, i] != o_occ[j, i]:
                flip_margins.append(float(marg["occ"][k, 1]))
            if res["depth_order"][i, j] != o_dep[i, j]:
                flip_margins.append(float(marg["depth"][k]))
        for name, occ, dep in (("hip", res["occ_order"], res["depth_order"]), ("oracle", o_occ, o_dep)):
            prf = infer.eval_order_recall_precision_f1(occ, item["gt_occ"], 0)
            w = infer.eval_depth_order_whdr(dep, (item["gt_depth"], item["gt_overlap"], item["gt_count"]))
            acc[name].append(list(prf) + [w["ovlOX_all"][0], w["ovlO_all"][0], w["ovlX_all"][0]])
    names = ["recall", "precision", "F1", "WHDR_all", "WHDR_ovl", "WHDR_dist"]
    mh = np.mean(np.asarray(acc["hip"]), 0)
    mo = np.mean(np.asarray(acc["oracle"]), 0)
    if verbose:
        print("%s: images %d x instances %d (pairs %d), S=%d, differing matrix entries: %d"
              % (dtype, n_images, n_inst, n_images * len(pairs), S, flips))
        for n, a, b in zip(names, mh, mo):
            print("  %-10s HIP %8.3f   oracle %8.3f   delta %+.3f pp" % (n, a, b, a - b))
        print("  decisions that differ: %d, largest oracle margin among them: %.2e (probability units)"
              % (len(flip_margins), max(flip_margins) if flip_margins else 0.0))
    run.flip_margins = flip_margins
    return dict(zip(names, (mh - mo).tolist())), flips


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:] if v.isdigit()]
    run(*(a + [20, 5, 256][len(a):]), dtype="bf16" if "bf16" in sys.argv else "fp32")
