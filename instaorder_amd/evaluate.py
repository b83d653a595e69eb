"""Validation loops of the reference's ``tools/test.py`` ``Tester`` (:165-175 dispatch, :187-287 ``eval_occ_depth_order``,
:288-400 ``eval_depth_order``, :402-475 ``eval_occ_order``) over the batched drivers of ``instaorder_amd.inference``:
per image -- instances from the reader, boxes enlarged as in ``expand_bbox`` (:155-163), ground truth from the reader
(or ``infer_gt_order``), order matrices from the chosen method, P / R / F1 (x100) and WHDR per overlap x equality
subset -- then the reference's means (plain mean for P / R / F1; WHDR over the images where the subset is not empty,
with its 1e-6 in the denominator).  Logging, png dumps and wandb are not reproduced.

Images are independent, so with ``world_size > 1`` each rank evaluates a contiguous slice
(``distributed_utils.shard_range``, the role of ``DistributedSequentialSampler``) and the per-image rows are
all-gathered; every rank returns the same dictionary.
"""
import collections

import numpy as np
import torch

from . import distributed_utils, inference as infer


def expand_bbox(bboxes, enlarge_box):
    """tools/test.py:155-163: xywh boxes -> square boxes of side max(sqrt(w*h*enlarge), 1.1w, 1.1h) around the centre."""
    out = []
    for bbox in bboxes:
        centerx = bbox[0] + bbox[2] / 2.
        centery = bbox[1] + bbox[3] / 2.
        size = max([np.sqrt(bbox[2] * bbox[3] * enlarge_box), bbox[2] * 1.1, bbox[3] * 1.1])
        out.append([int(centerx - size / 2.), int(centery - size / 2.), int(size), int(size)])
    return np.array(out)


WHDR_KEYS = ["%s_%s" % (o, e) for o in ("ovlX", "ovlO", "ovlOX") for e in ("eq", "neq", "all")]


def _gather_rows(rows, n_total, world_size):
    """rows: {image index: 1-D float64 row}; all ranks end up with the [n_total, width] table."""
    width = len(next(iter(rows.values()))) if rows else 0
    if world_size > 1:
        import torch.distributed as dist
        w = torch.tensor([width], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            w = w.cuda()
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        width = int(w.item())
    table = np.zeros((n_total, width + 1), np.float64)          # last column: 1 where this rank filled the row
    for i, r in rows.items():
        table[i, :width] = r
        table[i, width] = 1.0
    if world_size > 1:
        import torch.distributed as dist
        t = torch.from_numpy(table)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)               # rows are disjoint except for wrap-around padding
        table = t.cpu().numpy()
        table[:, :width] /= np.maximum(table[:, width:], 1.0)
    return table[:, :width]


def evaluate(model, data_reader, load_image, data_cfg, order_method, pairs="all", zd=0, disp_select_method="",
             gt_ordering="ann", world_size=1, rank=0, return_orders=False):
    """Dispatch of ``Tester.run`` on ``data_cfg['trainval_dataset']``.  ``order_method``: one of the method strings of
    tools/test.py -- the network methods, or the annotation-free baselines 'area' / 'yaxis' (tools/test.py:306-318,
    421-433) -- or a callable ``(modal_masks, dataset_name) -> order matrix`` (any other model-free rule).  Returns a dict with 'recall', 'precision', 'f1'
    (occlusion datasets) and / or 'WHDR_<ovl>_<eq>' (depth datasets), plus 'num_test_images'; with ``return_orders``
    also 'orders' = {image index: (occlusion matrix | None, depth matrix | None)} of this rank's images."""
    kind = data_cfg["trainval_dataset"]
    want_occ = kind in ("SupOcclusionOrderDataset", "PartialCompDataset", "SupDepthOccOrderDataset")
    want_dep = kind in ("SupDepthOrderDataset", "SupDepthOccOrderDataset")
    if not (want_occ or want_dep):
        raise Exception("No such trainval_dataset: {}".format(kind))
    n = data_reader.get_image_length()
    beg, end, _ = distributed_utils.shard_range(n, world_size, rank) if world_size > 1 else (0, n, n)
    mode, size = data_cfg["patch_or_image"], data_cfg["input_size"]
    dataset = data_cfg["dataset"]
    rows, orders = {}, {}
    for q in range(beg, end):
        i = q % n
        modal, category, bboxes, amodal_gt, image_fn = data_reader.get_image_instances(i, with_gt=True)
        if kind != "SupDepthOccOrderDataset" and data_cfg.get("use_category", False):
            modal = modal * category[:, None, None]
        image = None if (callable(order_method) or order_method in ("area", "yaxis")) else np.asarray(load_image(image_fn))
        boxes = expand_bbox(bboxes, data_cfg["enlarge_box"])
        gt_occ = gt_dep = None
        if want_occ:
            if dataset == "InstaOrder":
                gt_occ = data_reader.get_gt_ordering(i, "occlusion", data_cfg["remove_occ_bidirec"])
            else:
                gt_occ = data_reader.get_gt_ordering(i) if gt_ordering == "ann" else infer.infer_gt_order(modal, amodal_gt)
        if want_dep:
            gt_dep = data_reader.get_gt_ordering(i, "depth", rm_overlap=0 if kind == "SupDepthOccOrderDataset"
                                                 else data_cfg["remove_depth_overlap"])
        pred_occ = pred_dep = None
        if kind == "SupDepthOccOrderDataset":
            if order_method not in ("InstaOrderNet_od", "InstaDepthNet_od"):
                raise Exception("No such order method: {}".format(order_method))
            if order_method == "InstaOrderNet_od":
                pred_occ, pred_dep = infer.infer_order_sup_occ_depth(model, image, modal, boxes, pairs, order_method,
                                                                     mode, size, disp_select_method)
            else:
                rgb, masks = infer.resize_mode_inputs(next(model.model.parameters()).device, image, modal, size) \
                    if mode == "resize" else _identity_inputs(image, modal, size)
                res = infer.infer_depthnet_batched(model, rgb, masks, pairs=infer.select_pairs(modal, pairs))
                pred_occ, pred_dep = res["occ_order"], res["depth_order"]
        elif want_dep:
            if callable(order_method):      # a device-free ordering rule: (modal masks, dataset name) -> matrix
                pred_dep = order_method(modal, dataset)
            elif order_method == "area":    # tools/test.py:306-312: 'larger' for every dataset
                pred_dep = infer.infer_depth_order_area(modal, closer="larger")
            elif order_method == "yaxis":   # :314-318
                pred_dep = infer.infer_depth_order_yaxis(modal, closer="lower" if dataset in ("COCOA", "InstaOrder")
                                                         else "higher")
            elif order_method in ("InstaOrderNet_d", "InstaDepthNet_d", "midas_pretrained"):
                pred_dep, _ = infer.infer_order_sup_depth(model, image, modal, boxes, pairs, order_method, mode, size,
                                                          disp_select_method)
            else:
                raise Exception("No such order method: {}".format(order_method))
        else:
            if callable(order_method):
                pred_occ = order_method(modal, dataset)
            elif order_method == "area":    # tools/test.py:421-427
                pred_occ = infer.infer_occ_order_area(modal, occluder="larger")
            elif order_method == "yaxis":   # :429-433
                pred_occ = infer.infer_occ_order_yaxis(modal, occluder="lower" if dataset in ("COCOA", "InstaOrder")
                                                       else "higher")
            elif order_method in ("InstaOrderNet_o", "OrderNet"):
                pred_occ = infer.infer_order_sup_occ(model, image, modal, boxes, pairs, order_method, mode, size)
            else:
                raise Exception("No such order method: {}".format(order_method))
        row = []
        if want_occ:
            row += list(infer.eval_order_recall_precision_f1(pred_occ, gt_occ, zd))
        if want_dep:
            w = infer.eval_depth_order_whdr(pred_dep, [np.array(g) for g in gt_dep])
            row += [w[k][0] for k in WHDR_KEYS]
        rows[i] = np.asarray(row, np.float64)
        orders[i] = (pred_occ, pred_dep)
    table = _gather_rows(rows, n, world_size)
    out = collections.OrderedDict()
    col = 0
    if want_occ:
        for name in ("recall", "precision", "f1"):
            out[name] = float(table[:, col].sum() / n)                               # tools/test.py:467-469
            col += 1
    if want_dep:
        for k in WHDR_KEYS:
            v = table[:, col]
            valid = v != -1
            out["WHDR_" + k] = float(v[valid].sum() / (valid.sum() + 1e-6))          # tools/test.py:376-379
            col += 1
    out["num_test_images"] = n
    if return_orders:
        out["orders"] = orders
    return out


def _identity_inputs(image, modal, size):
    if image.shape[0] != image.shape[1] or image.shape[0] != size:
        raise NotImplementedError("InstaDepthNet_od evaluation: patch_or_image='resize', or square images of the network size")
    from .synthetic import image_mode_inputs
    rgb, masks = image_mode_inputs(image, modal, size)
    return torch.from_numpy(rgb), torch.from_numpy(masks)
