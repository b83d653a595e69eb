#!/usr/bin/env python3
"""Order accuracy on a TRAINED model (decisions away from the thresholds): HIP fp32 vs the CPU oracle vs HIP bf16.

tools/synthetic_val.py measures agreement on a random-weight stand-in whose heads are centred on the decision
thresholds -- every flip there has a margin below 1e-3, so accuracy deltas in percentage points say little.  Here the
network is first TRAINED (tools/train_synthetic.py: the whole chain of trainer.py:87-240 on scenes whose occlusion rule is
learnable), then unseen scenes go through the 'patch' pre-processing of inference.py:449-465 ONCE (device renderer) and
the very same network inputs are evaluated by
  * the HIP path in fp32 (infer_order_batched),
  * the CPU oracle (the reference's network arithmetic, pinned by tests/golden) on the same weights,
  * the HIP path in bf16 (same checkpoint loaded into a dtype='bf16' model),
with the reference's decision rule (inference.py:79-117) and metrics (inference.py:794-802).  Prints recall / precision /
F1 per path and the deltas in percentage points, and the oracle's decision margin wherever a path disagrees with it.

usage: python tools/trained_val.py [--iters 400] [--val 40] [--size 128]"""
import argparse
import importlib.util
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import instaorder_amd as ia  # noqa: E402
from instaorder_amd import inference as infer  # noqa: E402
from instaorder_amd import synthetic  # noqa: E402
from oracle import resnet_oracle as orc  # noqa: E402      (checker only)


def _train(iters, batch, size, out, seed):
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(ROOT, "tools", "train_synthetic.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    return ts.main(["--iters", str(iters), "--batch", str(batch), "--size", str(size), "--out", out, "--seed", str(seed)])


def run(iters=400, batch=64, size=128, n_val=40, n_inst=6, verbose=True, seed=0):
    out = tempfile.mkdtemp(prefix="io_trained_val_")
    f1_train = _train(iters, batch, size, out, seed)
    cfg = dict(algo="InstaOrderNet_o", lr=0.01, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
               backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True)
    models = {}
    for dt in ("fp32", "bf16"):
        m = ia.InstaOrderNet_o(dict(cfg, dtype=dt), dist_model=False)
        m.load_state(out, iters)
        m.switch_to("eval")
        models[dt] = m
    state = orc.state_from_numpy({k[len("module."):]: v.detach().cpu().numpy()
                                  for k, v in models["fp32"].model.state_dict().items()})
    rd = synthetic.SyntheticReader(3, n_images=n_val, n_inst=n_inst, max_side=200, min_side=120, empty_every=0, rule="lower")
    acc = {"oracle": [], "fp32": [], "bf16": []}
    flips = {"fp32": [], "bf16": []}
    margins = []
    npairs = 0
    for k in range(rd.get_image_length()):
        modal, _, bboxes, _, fn = rd.get_image_instances(k, with_gt=True)
        gt = rd.get_gt_ordering(k, type="occlusion")
        np.fill_diagonal(gt, -1)
        image = rd.load_image(fn)
        n = modal.shape[0]
        pairs = infer.select_pairs(modal, "all")
        npairs += len(pairs)
        planes = infer._preprocess_pairs(models["fp32"], image, modal, bboxes, pairs, "patch", size)
        rgb, mi, mj = (t.float().cpu() for t in planes)
        with torch.no_grad():
            z1 = orc.resnet_forward(state, torch.cat([mi, mj, rgb], 1), False)
            z2 = orc.resnet_forward(state, torch.cat([mj, mi, rgb], 1), False)
        d = infer.decide(z1, z2, 2, 0)
        marg = infer.decision_margins(torch.cat([z1, z2], 1), "InstaOrderNet_o")["occ"]
        margins.append(np.asarray(marg).reshape(-1))
        o_occ, _ = orc.order_matrices(n, pairs, d["i_over_j"], d["j_over_i"], None)
        acc["oracle"].append(infer.eval_order_recall_precision_f1(o_occ, gt, 0))
        for dt in ("fp32", "bf16"):
            res = infer.infer_order_batched(models[dt], None, torch.from_numpy(np.asarray(modal)), "InstaOrderNet_o",
                                            pairs=pairs, pair_planes=planes)
            acc[dt].append(infer.eval_order_recall_precision_f1(res["occ_order"], gt, 0))
            for q, (i, j) in enumerate(pairs):
                if res["occ_order"][i, j] != o_occ[i, j]:
                    flips[dt].append(float(marg[q, 0]))
                if res["occ_order"][j, i] != o_occ[j, i]:
                    flips[dt].append(float(marg[q, 1]))
    mean = {k: np.mean(np.asarray(v, np.float64), 0) for k, v in acc.items()}
    names = ["recall", "precision", "F1"]
    delta = {dt: dict(zip(names, (mean[dt] - mean["oracle"]).tolist())) for dt in ("fp32", "bf16")}
    margins = np.concatenate(margins)
    if verbose:
        print("oracle decision margins |p - 0.5|: median %.3f, share below 1e-3: %.2f %%" % (
            float(np.median(margins)), 100.0 * float((margins < 1e-3).mean())))
        print("trained %d iterations x %d pairs at %d^2 (validation F1 of the training run %.1f); %d unseen images, %d "
              "pairs, 'patch' inputs" % (iters, batch, size, f1_train, n_val, npairs))
        for i, nme in enumerate(names):
            print("  %-10s oracle %8.3f   HIP fp32 %8.3f (%+.3f pp)   HIP bf16 %8.3f (%+.3f pp)" % (
                nme, mean["oracle"][i], mean["fp32"][i], delta["fp32"][nme], mean["bf16"][i], delta["bf16"][nme]))
        for dt in ("fp32", "bf16"):
            print("  %s: %d of %d decisions differ from the oracle's; the oracle's margin there: %s" % (
                dt, len(flips[dt]), 2 * npairs, ", ".join("%.1e" % v for v in sorted(flips[dt])[-5:]) or "-"))
    return dict(delta=delta, flips=flips, mean=mean, npairs=npairs, f1_train=f1_train, margins=margins)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--val", type=int, default=40)
    ap.add_argument("--size", type=int, default=128)
    a = ap.parse_args()
    run(iters=a.iters, size=a.size, n_val=a.val)
