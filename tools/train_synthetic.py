#!/usr/bin/env python3
"""End-to-end run of the whole chain on synthetic scenes, in the shape of the reference's trainer.py loop
(trainer.py:87-240): model factory -> StepLRScheduler -> DistributedGivenIterationSampler -> device input pipeline
(datasets.SupOcclusionOrderBatches + BatchPrefetcher) -> set_input / step -> reduce_tensors -> checkpoint ->
validation with the 'patch' inference driver + precision / recall / F1 (tools/test.py:402-495, inference.py:794-802),
and a resume from the checkpoint that continues the same index stream and learning-rate schedule.
usage: python tools/train_synthetic.py [--iters 120] [--batch 64] [--size 128] [--dtype fp32|bf16] [--out DIR]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instaorder_amd as ia  # noqa: E402
from instaorder_amd import datasets, distributed_utils, inference, scheduler, synthetic  # noqa: E402


def validate(model, reader, size):
    model.switch_to("eval")
    acc = []
    for k in range(reader.get_image_length()):
        modal, _, bboxes, _, fn = reader.get_image_instances(k, with_gt=True)
        gt = reader.get_gt_ordering(k, type="occlusion")
        np.fill_diagonal(gt, -1)
        order = inference.infer_order_sup_occ(model, reader.load_image(fn), modal, bboxes, "all", "InstaOrderNet_o",
                                              "patch", size)
        acc.append(inference.eval_order_recall_precision_f1(order, gt, 0))
    model.switch_to("train")
    return np.mean(np.asarray(acc, np.float64), 0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=120)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--out", default="")
    ap.add_argument("--resume-at", type=int, default=0, help="stop at this iteration, reload the checkpoint, continue")
    ap.add_argument("--seed", type=int, default=-1, help=">= 0: seed torch's generator first (init_weights draws from it; "
                    "at lr 0.01 the trajectory -- and how well 400 iterations learn -- depends on the draw)")
    a = ap.parse_args(argv)
    if a.seed >= 0:
        torch.manual_seed(a.seed)
    out = a.out or tempfile.mkdtemp(prefix="io_train_")
    mcfg = dict(algo="InstaOrderNet_o", lr=0.01, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
                backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True, dtype=a.dtype,
                lr_steps=[int(a.iters * 0.7)], lr_mults=[0.1], warmup_lr=[], warmup_steps=[])
    dcfg = dict(input_size=a.size, patch_or_image="patch", data_mean=[0.485, 0.456, 0.406],
                data_std=[0.229, 0.224, 0.225], load_rgb=True, use_category=False, dataset="InstaOrder",
                remove_occ_bidirec=0, base_aug=dict(flip=True, shift=[-0.2, 0.2], scale=[0.8, 1.2]))
    train_rd = synthetic.SyntheticReader(1, n_images=48, n_inst=6, max_side=200, min_side=120, empty_every=0, rule="lower")
    val_rd = synthetic.SyntheticReader(2, n_images=8, n_inst=6, max_side=200, min_side=120, empty_every=0, rule="lower")

    def run(first_iter, last_iter, load_from):
        model = ia.InstaOrderNet_o(mcfg, dist_model=False)
        start = -1
        if load_from is not None:
            model.load_state(load_from[0], load_from[1], resume=True)      # single_stage_model.py:54-64
            start = load_from[1] - 1
        sched = scheduler.StepLRScheduler(model.optim, mcfg["lr_steps"], mcfg["lr_mults"], mcfg["lr"],
                                          mcfg["warmup_lr"], mcfg["warmup_steps"], last_iter=start)
        batches = datasets.SupOcclusionOrderBatches(dcfg, "train", "InstaOrderNet_o", train_rd, train_rd.load_image,
                                                    rng=np.random.RandomState(100 + first_iter))
        sampler = distributed_utils.DistributedGivenIterationSampler(batches, a.iters, a.batch, world_size=1, rank=0,
                                                                     last_iter=start)
        idx = np.asarray(list(iter(sampler)), np.int64).reshape(-1, a.batch)
        loader = datasets.BatchPrefetcher(batches, (row.tolist() for row in idx[:last_iter - first_iter]))
        model.switch_to("train")
        t0 = time.time()
        for k, inputs in enumerate(loader):
            curr = first_iter + k + 1
            sched.step(curr - 1)
            model.set_input(*inputs)
            loss = model.step()["loss"]                                     # world size 1: nothing to reduce
            if curr % 20 == 0 or curr == last_iter:
                print("iter %4d  lr %.4g  loss %.4f  (%.0f pairs/s)" % (curr, model.optim.param_groups[0]["lr"],
                                                                      float(loss), a.batch * (k + 1) / (time.time() - t0)),
                      flush=True)
        model.save_state(out, last_iter)                                    # single_stage_model.py:66-72
        return model

    stop = a.resume_at if 0 < a.resume_at < a.iters else a.iters
    model = run(0, stop, None)
    if stop < a.iters:
        print("resuming from %s/ckpt_iter_%d.pth.tar" % (out, stop))
        model = run(stop, a.iters, (out, stop))
    p, r, f1 = validate(model, val_rd, a.size)[:3]
    print("validation ('patch' mode, %d images): recall %.1f precision %.1f F1 %.1f" % (val_rd.get_image_length(), p, r, f1))
    return f1


if __name__ == "__main__":
    main()
