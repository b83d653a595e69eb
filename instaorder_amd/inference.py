"""Pairwise order inference -- host-side mirror of the hot-path part of the reference's inference.py.

The reference evaluates the O(n^2) instance pairs of an image one at a time (batch 1, two model
calls and two ``.cpu().item()`` syncs per pair: inference.py:439-512, 196-214).  Here all pairs of an
image (both mask orders) form ONE batch for the HIP network, the direction-averaged decision rule
runs on the device, and a single copy brings the decisions back.  The rule itself is unchanged:

  occlusion (inference.py:210-213):  P(i over j) = (out1[:,1] + out2[:,0]) / 2 > 0.5
                                     P(j over i) = (out1[:,0] + out2[:,1]) / 2 > 0.5
  depth     (inference.py:152-167):  argmax( (d1[:,0]+d2[:,1])/2, (d1[:,1]+d2[:,0])/2, (d1[:,2]+d2[:,2])/2 )
                                     0: i closer, 1: i farther, 2: equal

with out* = sigmoid(logits), d* = softmax(logits).  Metrics (recall/precision/F1, WHDR) follow
inference.py:757-802.  Image pre-processing (crop / resize / normalise, cv2) is out of scope: callers
pass the already normalised ``rgb[1,3,S,S]`` and the float masks ``[N,S,S]`` (what the reference's
``patch_or_image == 'image'`` branch produces for a square image of the network size).
"""
import collections

import numpy as np
import torch

from . import distributed_utils, engine


def upper_pairs(n):
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def bordering(a, b):
    """Reference inference.py:691-696: does mask ``a``, dilated once with the 3x3 cross, touch mask ``b``?
    (cv2.dilate with a cross kernel on a binary image = OR of the four one-pixel shifts.)"""
    a = np.asarray(a) == 1
    d = a.copy()
    d[1:] |= a[:-1]
    d[:-1] |= a[1:]
    d[:, 1:] |= a[:, :-1]
    d[:, :-1] |= a[:, 1:]
    return bool(np.any(d & (np.asarray(b) != 0)))


# ---- annotation-free baselines and ground-truth derivation (host; inference.py:272-347, 719-754) -------------------
def _centre_y(mask):
    return np.where(np.asarray(mask) == 1)[0].mean()


def _pairwise(inmodal, need_border, key, direction):
    """Shared double loop of the four heuristics: for i < j (optionally only bordering pairs) order the pair by
    ``key`` (strict <; ties go to (j, i) as in the reference) and mark [a, b] = 1 with (a, b) per ``direction``."""
    n = inmodal.shape[0]
    order = np.zeros((n, n), dtype=np.int64)
    keys = [key(m) for m in inmodal]
    for i in range(n):
        for j in range(i + 1, n):
            if need_border and not bordering(inmodal[i], inmodal[j]):
                continue
            lo, hi = (i, j) if keys[i] < keys[j] else (j, i)          # lo: smaller area / smaller y (higher up)
            a, b = (lo, hi) if direction else (hi, lo)
            order[a, b] = 1
    return order


def infer_occ_order_area(inmodal, occluder="smaller"):
    """inference.py:272-289: among bordering pairs the smaller (or larger) mask occludes."""
    return _pairwise(inmodal, True, lambda m: m.sum(), occluder == "smaller")


def infer_occ_order_yaxis(inmodal, occluder="lower"):
    """inference.py:292-307.  Reference naming quirk kept: its ``lower`` is the mask with the SMALLER mean row index
    (higher up in the image), and occluder='lower' marks that one as the occluder."""
    return _pairwise(inmodal, True, _centre_y, occluder == "lower")


def infer_depth_order_area(inmodal, closer="smaller"):
    """inference.py:310-328: every pair; the smaller (or larger) mask is closer."""
    return _pairwise(inmodal, False, lambda m: m.sum(), closer == "smaller")


def infer_depth_order_yaxis(inmodal, closer="lower"):
    """inference.py:331-346: every pair; closer='lower' marks the mask with the LARGER mean row index (lower in the
    image) as closer -- here the reference's variable names are the right way round."""
    return _pairwise(inmodal, False, _centre_y, closer != "lower")


def infer_gt_order(inmodal, amodal):
    """inference.py:719-739 (KINS / COCOA ground truth): for bordering pairs, i occludes j when i's visible mask covers
    at least as much of j's amodal mask as the other way round (and the overlap is not empty)."""
    n = inmodal.shape[0]
    gt = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(i + 1, n):
            if not bordering(inmodal[i], inmodal[j]):
                continue
            occ_ij = int(((inmodal[i] == 1) & (amodal[j] == 1)).sum())
            occ_ji = int(((inmodal[j] == 1) & (amodal[i] == 1)).sum())
            if occ_ij == 0 and occ_ji == 0:
                continue
            if occ_ij >= occ_ji:
                gt[i, j] = 1
            else:
                gt[j, i] = 1
    return gt


def eval_order(order_matrix, gt_order_matrix):
    """inference.py:742-754: (correct pairs, pairs, correct occluded pairs, occluded pairs, error table)."""
    n = order_matrix.shape[0]
    same = order_matrix == gt_order_matrix
    allpair_true = (same.sum() - n) / 2
    allpair = (n * n - n) / 2
    occpair_true = (same & (gt_order_matrix != 0)).sum() / 2
    occpair = (gt_order_matrix != 0).sum() / 2
    err = np.where(~same)
    show_err = np.concatenate([np.array(err).T + 1, gt_order_matrix[err][:, None], order_matrix[err][:, None]], axis=1)
    return allpair_true, allpair, occpair_true, occpair, show_err


def select_pairs(inmodal, pairs):
    """The pair list of the reference's double loop: every i < j ("all") or only neighbouring instances ("nbor",
    inference.py:446-447: ``bordering(inmodal[i], inmodal[j])`` -- only the first mask is dilated)."""
    n = inmodal.shape[0]
    if pairs == "all":
        return upper_pairs(n)
    if pairs == "nbor":
        return [(i, j) for (i, j) in upper_pairs(n) if bordering(inmodal[i], inmodal[j])]
    raise ValueError("pairs must be 'all' or 'nbor', got %r" % (pairs,))


def _heads(method, net=None):
    if method in ("InstaOrderNet_od",):
        return 2, 3
    if method in ("InstaOrderNet_o",):
        return 2, 0
    if method in ("InstaOrderNet_d",):
        return 0, 3
    if method == "OrderNet":        # one softmax head of 3 (1>2, 2>1, none) or 4 (+ both: OrderNet_ext) classes
        return 0, (int(net.head_dims[0]) if net is not None else 3)
    raise ValueError("method name should be one of OrderNet / InstaOrderNet_o / InstaOrderNet_od / InstaOrderNet_d")


def decide_ordernet(logits1, logits2):
    """net_forward_OrderNet (inference.py:44-76) for P pairs: softmax of both mask orders, direction-averaged class
    probabilities (1 over 2, 2 over 1, none[, both]), argmax -> the two occlusion booleans."""
    o1, o2 = torch.softmax(logits1, 1), torch.softmax(logits2, 1)
    cols = [(o1[:, 1] + o2[:, 0]) / 2, (o1[:, 0] + o2[:, 1]) / 2, (o1[:, 2] + o2[:, 2]) / 2]
    cols.append((o1[:, 3] + o2[:, 3]) / 2 if o1.shape[1] == 4 else torch.zeros_like(cols[0]))
    idx = torch.stack(cols, 1).argmax(1)          # first maximum wins, as np.argmax does
    return {"i_over_j": (idx == 0) | (idx == 3), "j_over_i": (idx == 1) | (idx == 3)}


def decide(logits1, logits2, kocc, kdep):
    """Direction-averaged decisions for P pairs from raw logits of the two mask orders."""
    res = {}
    if kocc:
        o1, o2 = torch.sigmoid(logits1[:, :kocc]), torch.sigmoid(logits2[:, :kocc])
        res["i_over_j"] = (o1[:, 1] + o2[:, 0]) / 2 > 0.5
        res["j_over_i"] = (o1[:, 0] + o2[:, 1]) / 2 > 0.5
    if kdep:
        d1 = torch.softmax(logits1[:, kocc:], 1)
        d2 = torch.softmax(logits2[:, kocc:], 1)
        avg = torch.stack([(d1[:, 0] + d2[:, 1]) / 2, (d1[:, 1] + d2[:, 0]) / 2, (d1[:, 2] + d2[:, 2]) / 2], 1)
        res["depth"] = avg.argmax(1)
    return res


def net_forward_InstaDepthNet(model, image, inmodal1, inmodal2):
    """Reference signature (inference.py:107-137): one pair through InstaDepthNet_od / _d, both mask orders,
    direction-averaged decisions.  image [1,3,H,W] normalised, inmodal* [H,W] arrays in {0,1}.
    Returns (argidx_depth, is_1_over_2, is_2_over_1, disp1, disp2)."""
    dev = next(model.model.parameters()).device
    m1 = torch.as_tensor(np.asarray(inmodal1, dtype=np.float32), device=dev)[None, None]
    m2 = torch.as_tensor(np.asarray(inmodal2, dtype=np.float32), device=dev)[None, None]
    image = image.to(dev)
    with torch.no_grad():
        disp1, dep1, occ1 = model.model(image, m1, m2)
        disp2, dep2, occ2 = model.model(image, m2, m1)
        if occ1 is not None:
            r = decide(torch.cat([occ1, dep1], 1), torch.cat([occ2, dep2], 1), 2, 3)
            return int(r["depth"][0]), bool(r["i_over_j"][0]), bool(r["j_over_i"][0]), disp1, disp2
        r = decide(dep1, dep2, 0, 3)
    return int(r["depth"][0]), 0, 0, disp1, disp2


def decision_margins(pair_logits, method):
    """Distance of each decision from its threshold (|p - 0.5| for the two occlusion directions, gap
    between the two largest averaged depth probabilities) -- used by parity tests to leave aside
    decisions the reference itself takes inside fp32 noise.  pair_logits: [P, 2K] = (order a,b | order b,a)."""
    kocc, kdep = _heads(method)
    K = kocc + kdep
    l1, l2 = pair_logits[:, :K].double(), pair_logits[:, K:].double()
    out = {}
    if kocc:
        o1, o2 = torch.sigmoid(l1[:, :kocc]), torch.sigmoid(l2[:, :kocc])
        out["occ"] = torch.stack([((o1[:, 1] + o2[:, 0]) / 2 - 0.5).abs(), ((o1[:, 0] + o2[:, 1]) / 2 - 0.5).abs()],
                                 1).numpy()
    if kdep:
        d1, d2 = torch.softmax(l1[:, kocc:], 1), torch.softmax(l2[:, kocc:], 1)
        avg = torch.stack([(d1[:, 0] + d2[:, 1]) / 2, (d1[:, 1] + d2[:, 0]) / 2, (d1[:, 2] + d2[:, 2]) / 2], 1)
        top = avg.topk(2, 1).values
        out["depth"] = (top[:, 0] - top[:, 1]).numpy()
    return out


def infer_order_batched(model, rgb, masks, method, pairs=None, max_pairs=256, return_logits=False,
                        world_size=1, rank=0, pair_planes=None):
    """Order matrices of one image.  rgb[1,3,S,S] normalised fp32, masks[N,S,S] in {0,1} (or [.., H, W] with both sides
    multiples of 32: the 'orig' mode).

    ``pair_planes = (rgb[P,3,S,S], modal_i[P,1,S,S], modal_j[P,1,S,S])`` (one entry per pair of ``pairs``, e.g. from
    ``datasets.PairRenderer``) replaces the shared image and masks: the 'patch' pre-processing of the reference crops
    every pair differently (inference.py:449-465).

    With ``world_size > 1`` the pair list is sharded contiguously across ranks
    (``distributed_utils.shard_range``) and the tiny per-pair decisions are all-gathered."""
    net = model.net
    kocc, kdep = _heads(method, net)
    n = masks.shape[0]
    pairs = upper_pairs(n) if pairs is None else list(pairs)
    P = len(pairs)
    SH, S = (masks.shape[-2:] if pair_planes is None else pair_planes[0].shape[-2:])      # S = width; SH != S: 'orig'
    SH, S = int(SH), int(S)
    dev = net.flat_params.device
    if pair_planes is None:
        rgb = rgb.to(dev, torch.float32).contiguous()
        masks = masks.to(dev, torch.float32).contiguous()
    else:
        pair_planes = [t.to(dev, torch.float32).contiguous() for t in pair_planes]
        assert all(t.shape[0] == P for t in pair_planes), "pair_planes: one entry per pair"
    beg, end = 0, P
    if world_size > 1:
        beg, end, _ = distributed_utils.shard_range(P, world_size, rank)
    my = [pairs[k % P] for k in range(beg, end)] if P else []
    K = kocc + kdep
    l1 = torch.empty((len(my), K), device=dev)
    l2 = torch.empty((len(my), K), device=dev)
    HW = SH * S
    was_training = net.training
    net.eval()
    with torch.no_grad():
        for c0 in range(0, len(my), max_pairs):
            chunk = my[c0:c0 + max_pairs]
            p = len(chunk)
            x8 = torch.empty((2 * p, SH, S, 8), device=dev, dtype=engine.TORCH_DTYPE[net.dtype])
            if pair_planes is None:
                ii = torch.tensor([a for a, _ in chunk], device=dev)
                jj = torch.tensor([b for _, b in chunk], device=dev)
                mi, mj = masks[ii].contiguous(), masks[jj].contiguous()
                rgbp, rs = [(rgb, c * HW) for c in range(3)], 0
            else:
                k0 = beg + c0                  # (a sharded list wraps around only when P < world_size)
                sel = [(k0 + t) % P for t in range(p)]
                if sel == list(range(sel[0], sel[0] + p)):
                    rp, mi, mj = [t[sel[0]:sel[0] + p] for t in pair_planes]
                else:
                    idx = torch.tensor(sel, device=dev)
                    rp, mi, mj = [t[idx].contiguous() for t in pair_planes]
                rgbp, rs = [(rp, c * HW) for c in range(3)], 3 * HW
            engine.pack_planes([(mi, 0), (mj, 0)] + rgbp, [HW, HW, rs, rs, rs], p, SH, S, x8[:p])
            engine.pack_planes([(mj, 0), (mi, 0)] + rgbp, [HW, HW, rs, rs, rs], p, SH, S, x8[p:])
            z = net.forward_packed(x8, 1)
            l1[c0:c0 + p], l2[c0:c0 + p] = z[:p], z[p:]
    net.train(was_training)
    if world_size > 1:
        import torch.distributed as dist
        both = torch.cat([l1, l2], 1)
        gathered = [torch.empty_like(both) for _ in range(world_size)]
        dist.all_gather(gathered, both)
        both = torch.cat(gathered, 0)[:P]
        l1, l2 = both[:, :K], both[:, K:]
    ordernet = method == "OrderNet"
    dec = decide_ordernet(l1, l2) if ordernet else decide(l1, l2, kocc, kdep)
    if ordernet:
        kocc, kdep = 2, 0                # an occlusion matrix only
    occ = np.zeros((n, n), dtype=np.int64)
    dep = np.zeros((n, n), dtype=np.int64)
    host = {k: v.cpu().numpy() for k, v in dec.items()}      # one D2H for the whole image
    for k, (i, j) in enumerate(pairs):
        if kocc:
            if host["i_over_j"][k]:
                occ[i, j] = 1
            if host["j_over_i"][k]:
                occ[j, i] = 1
        if kdep:
            d = int(host["depth"][k])
            if d == 0:
                dep[i, j], dep[j, i] = 1, 0
            elif d == 1:
                dep[i, j], dep[j, i] = 0, 1
            else:
                dep[i, j] = dep[j, i] = 2
    res = {"pairs": pairs, "occ_order": occ, "depth_order": dep}
    if return_logits:
        res["pair_logits"] = torch.cat([l1, l2], 1).cpu().numpy()
    return res


_RENDERERS = {}


def _preprocess_pairs(model, image, inmodal, bboxes, pair_list, patch_or_image, input_size):
    """The per-pair pre-processing of inference.py:449-482 on the device (datasets.PairRenderer): 'patch' = square
    crop around the pair (zero padded), INTER_CUBIC; 'image' = zero padding to a square, INTER_LINEAR; masks
    INTER_NEAREST; then x / 255 and the ImageNet mean / std (utils/data_utils.py:9-10, 28-34).  Returns
    (rgb[P,3,S,S], modal_i[P,1,S,S], modal_j[P,1,S,S]).  The 'resize' / 'orig' modes share one image among all pairs
    (resize_mode_inputs / orig_mode_inputs)."""
    from . import datasets
    if patch_or_image not in ("patch", "image"):
        raise ValueError("patch_or_image=%r: one of 'patch', 'image', 'resize', 'orig'" % (patch_or_image,))
    dev = model.net.flat_params.device
    modal = np.ascontiguousarray(inmodal.astype(np.uint8))
    image = np.ascontiguousarray(image.astype(np.uint8))
    _, hh, ww = modal.shape
    items = []
    for i, j in pair_list:
        if patch_or_image == "patch":
            cx, cy, size = datasets.patch_box(bboxes, i, j)
            box = (int(cx - size / 2.), int(cy - size / 2.), int(size), int(size))
            items.append((0, i, j, box, datasets.INTER_CUBIC, False))
        else:
            hw = int(max(hh, ww))
            box = (-((hw - ww) // 2), -((hw - hh) // 2), hw, hw)
            items.append((0, i, j, box, datasets.INTER_LINEAR, False))
    return _renderer(dev, input_size).render([image], [modal], items)


def _renderer(dev, input_size):
    """input_size: S, or (height, width) for the 'orig' mode"""
    from . import datasets
    key = (tuple(int(v) for v in input_size) if isinstance(input_size, (tuple, list)) else int(input_size), str(dev))
    if key not in _RENDERERS:
        _RENDERERS[key] = datasets.PairRenderer(input_size, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], dev)
    return _RENDERERS[key]


def resize_mode_inputs(dev, image, inmodal, input_size):
    """The 'resize' pre-processing (inference.py:484-488, 562-566): ``transform_resize`` = INTER_CUBIC on the float64
    image / 255. + ImageNet normalisation (utils/data_utils.py:37-53) and INTER_NEAREST masks, the whole image squeezed
    to input_size x input_size (a multiple of 32, so MiDaS' Resize keeps it).  One image shared by all pairs:
    returns (rgb[1,3,S,S], masks[N,S,S]) on the device."""
    if input_size % 32:
        raise ValueError("'resize' mode: input_size must be a multiple of 32 (midas/transforms.py:96-105)")
    return _whole_image_inputs(dev, image, inmodal, input_size)


def _whole_image_inputs(dev, image, inmodal, size):
    from . import datasets
    modal = np.ascontiguousarray(inmodal.astype(np.uint8))
    image = np.ascontiguousarray(image.astype(np.uint8))
    n, hh, ww = modal.shape
    box = (0, 0, ww, hh)
    r = _renderer(dev, size)
    rgb, _, _ = r.render([image], [modal], [(0, 0, 0, box, datasets.INTER_CUBIC_F64, False)])
    _, m, _ = r.render([image], [modal], [(0, i, i, box, datasets.INTER_CUBIC_F64, False) for i in range(n)],
                       load_rgb=False)
    return rgb, m[:, 0]


def get_closest_int_multiple_of(orig_num, multiplier):
    """utils/data_utils.py:13-17 (a remainder of exactly half the multiplier rounds UP)"""
    if orig_num % multiplier >= multiplier // 2:
        return orig_num + multiplier - (orig_num % multiplier)
    return orig_num - (orig_num % multiplier)


def orig_mode_inputs(dev, image, inmodal):
    """The 'orig' pre-processing (inference.py:401-407, 490-496, 569-575): the whole image at its own aspect ratio, both
    sides rounded to the closest multiple of 32 -- ``transform_resize(image, ww, hh)`` (INTER_CUBIC on image / 255. in
    float64, ImageNet normalisation; MiDaS' Resize keeps a size that is already a multiple of 32) and INTER_NEAREST masks.
    Returns (rgb[1,3,hh,ww], masks[N,hh,ww]) on the device."""
    _, hh, ww = inmodal.shape
    hh, ww = get_closest_int_multiple_of(int(hh), 32), get_closest_int_multiple_of(int(ww), 32)
    if hh < 32 or ww < 32:
        raise ValueError("'orig' mode: the image rounds to %d x %d; the network needs at least 32 x 32" % (hh, ww))
    return _whole_image_inputs(dev, image, inmodal, (hh, ww))


def _infer_sup(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size):
    pair_list = select_pairs(inmodal, pairs)
    n = inmodal.shape[0]
    if not pair_list:
        z = np.zeros((n, n), dtype=np.int64)
        return {"occ_order": z, "depth_order": z.copy()}
    if patch_or_image in ("resize", "orig"):
        dev = model.net.flat_params.device
        rgb, masks = (resize_mode_inputs(dev, image, inmodal, input_size) if patch_or_image == "resize"
                      else orig_mode_inputs(dev, image, inmodal))
        return infer_order_batched(model, rgb, masks, method, pairs=pair_list)
    planes = _preprocess_pairs(model, image, inmodal, bboxes, pair_list, patch_or_image, input_size)
    return infer_order_batched(model, None, torch.from_numpy(np.asarray(inmodal)), method, pairs=pair_list,
                               pair_planes=planes)


def infer_order_sup_occ(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size=256, use_rgb=True):
    """Reference signature (inference.py:439-512): image uint8 [H,W,3], inmodal [N,H,W], bboxes [N,4] xywh; returns
    the occlusion order matrix (1 at [i, j] = i occludes j)."""
    return _infer_sup(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size)["occ_order"]


def infer_order_sup_occ_depth(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size,
                              disp_select_method=""):
    """Reference signature (inference.py:349-436); returns (occ_order, depth_order)."""
    res = _infer_sup(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size)
    return res["occ_order"], res["depth_order"]


def _matrices(n, pairs, dec, want_occ):
    occ = np.zeros((n, n), dtype=np.int64)
    dep = np.zeros((n, n), dtype=np.int64)
    host = {k: v.cpu().numpy() for k, v in dec.items()}
    for k, (i, j) in enumerate(pairs):
        if want_occ and "i_over_j" in host:
            occ[i, j] = int(host["i_over_j"][k])
            occ[j, i] = int(host["j_over_i"][k])
        d = int(host["depth"][k])
        if d == 0:
            dep[i, j], dep[j, i] = 1, 0
        elif d == 1:
            dep[i, j], dep[j, i] = 0, 1
        else:
            dep[i, j] = dep[j, i] = 2
    return occ, dep


def infer_depthnet_batched(model, rgb, masks, pairs=None, max_pairs=64):
    """All pairs of one image through InstaDepthNet_od / _d (the per-pair loop of inference.py:515-625 calls the whole
    net twice per pair): the image-only encoder + decoder run ONCE per image, the order branches see every pair in
    both mask orders as one batch (eval mode: BatchNorm uses the running estimates, so batching is exact).
    rgb [1,3,S,S] normalised, masks [N,S,S].  Returns dict(pairs, depth_order, occ_order | None, disp [S,S])."""
    from . import midas_net, ops
    net = model.model.module
    dev = next(net.parameters()).device
    n = masks.shape[0]
    pairs = upper_pairs(n) if pairs is None else list(pairs)
    rgb = rgb.to(dev, torch.float32)
    masks = masks.to(dev, torch.float32)
    was_training = net.training
    net.eval()
    has_occ = hasattr(net, "oo_net")
    dep1, dep2, occ1, occ2 = [], [], [], []
    with torch.no_grad():
        disp, feats = net._encode_decode(rgb)
        for c0 in range(0, len(pairs), max_pairs):
            chunk = pairs[c0:c0 + max_pairs]
            p = len(chunk)
            ii = torch.tensor([a for a, _ in chunk], device=dev)
            jj = torch.tensor([b for _, b in chunk], device=dev)
            mi, mj = masks[ii][:, None], masks[jj][:, None]
            x8m = ops.nhwc_from_nchw(torch.cat([torch.cat([mi, mj], 1), torch.cat([mj, mi], 1)], 0), pad_to=8,
                                     dtype=net._act_dtype())
            l1, l2, l3 = (f.expand(2 * p, -1, -1, -1).contiguous() for f in feats)
            if has_occ:
                d = net._order_branch(net.do_net, net.depth_fc, x8m, l1, l2, l3)
                o = net._order_branch(net.oo_net, net.occ_fc, x8m, l1, l2, l3)
                occ1.append(o[:p]); occ2.append(o[p:])
            else:
                d = net._order_branch(net.gdo_net, net.fc, x8m, l1, l2, l3)
            dep1.append(d[:p]); dep2.append(d[p:])
    net.train(was_training)
    d1, d2 = torch.cat(dep1, 0), torch.cat(dep2, 0)
    if has_occ:
        dec = decide(torch.cat([torch.cat(occ1, 0), d1], 1), torch.cat([torch.cat(occ2, 0), d2], 1), 2, 3)
    else:
        dec = decide(d1, d2, 0, 3)
    occ, dep = _matrices(n, pairs, dec, has_occ)
    return {"pairs": pairs, "depth_order": dep, "occ_order": occ if has_occ else None, "disp": disp[0]}


def net_forward_midas_pretrained(pred_disp, inmodal1, inmodal2, disp_select_method):
    """Reference signature (inference.py:79-104): depth order of two instances from a disparity map (0: first is
    closer, 1: farther, 2: equal) by the mean / median of the 5-95 % clipped inverse disparity inside each mask."""
    dev = pred_disp.device
    depth = 1 / (pred_disp + 1e-6)
    m1 = torch.as_tensor(np.asarray(inmodal1).astype(bool), device=dev)
    m2 = torch.as_tensor(np.asarray(inmodal2).astype(bool), device=dev)
    vals = []
    for m in (m1, m2):
        v = depth[m]
        v = torch.clip(v, torch.quantile(v, 0.05), torch.quantile(v, 0.95))
        vals.append(torch.median(v) if disp_select_method == "median" else torch.mean(v))
    if vals[0] < vals[1]:
        return 0
    return 1 if vals[0] > vals[1] else 2


def infer_order_sup_depth(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size, disp_select_method,
                          use_rgb=True):
    """Reference signature (inference.py:515-625); returns (depth order matrix, clipped disparity | None).
    Methods: InstaOrderNet_d (the batched ResNet path) and InstaDepthNet_d / InstaDepthNet_od (batched MiDaS path;
    with ``disp_select_method`` 'mean' / 'median' the order comes from the predicted disparity instead of the head)."""
    if method == "InstaOrderNet_d":
        return _infer_sup(model, image, inmodal, bboxes, pairs, method, patch_or_image, input_size)["depth_order"], None
    if method == "midas_pretrained":
        # inference.py:583-590: `model` is the bare MidasNet; the order comes from its disparity under the two masks
        dev = next(model.parameters()).device
        if patch_or_image == "resize":
            rgb, masks = resize_mode_inputs(dev, image, inmodal, input_size)
        elif patch_or_image == "orig":
            rgb, masks = orig_mode_inputs(dev, image, inmodal)
        elif patch_or_image == "image" and image.shape[0] == image.shape[1] == input_size:
            from .synthetic import image_mode_inputs
            rgb, masks = image_mode_inputs(image, inmodal, input_size)
            rgb, masks = torch.from_numpy(rgb).to(dev), torch.from_numpy(masks)
        else:
            raise NotImplementedError("midas_pretrained: patch_or_image='resize' / 'orig', or 'image' on square images of "
                                      "the network size")
        with torch.no_grad():
            disp = model(rgb.to(dev)).squeeze().float()
        clipped = torch.clip(disp, torch.quantile(disp, 0.05), torch.quantile(disp, 0.95))
        masks_np = masks.cpu().numpy() if torch.is_tensor(masks) else masks
        n = inmodal.shape[0]
        order = np.zeros((n, n), dtype=np.int64)
        for i, j in select_pairs(inmodal, pairs):
            a = net_forward_midas_pretrained(disp, masks_np[i], masks_np[j], disp_select_method)
            if a == 0:
                order[i, j], order[j, i] = 1, 0
            elif a == 1:
                order[i, j], order[j, i] = 0, 1
            else:
                order[i, j] = order[j, i] = 2
        return order, clipped
    if method not in ("InstaDepthNet_d", "InstaDepthNet_od"):
        raise ValueError("method name should be one of {InstaOrderNet_d, midas_pretrained, InstaDepthNet_d, InstaDepthNet_od}")
    # the batched MiDaS path runs the encoder once per IMAGE, which needs one image shared by all pairs
    plist = select_pairs(inmodal, pairs)
    if patch_or_image == "resize":              # what the reference's InstaDepthNet configs use (config.yaml:51)
        dev = next(model.model.parameters()).device
        rgb, masks = resize_mode_inputs(dev, image, inmodal, input_size)
        rgb, masks = rgb.cpu().numpy(), masks.cpu().numpy()
    elif patch_or_image == "orig":              # the whole image at its own aspect ratio (inference.py:569-575)
        dev = next(model.model.parameters()).device
        rgb, masks = orig_mode_inputs(dev, image, inmodal)
        rgb, masks = rgb.cpu().numpy(), masks.cpu().numpy()
    elif patch_or_image == "image" and image.shape[0] == image.shape[1] == input_size:
        from .synthetic import image_mode_inputs
        rgb, masks = image_mode_inputs(image, inmodal, input_size)
    else:
        raise NotImplementedError("InstaDepthNet inference: patch_or_image='resize' / 'orig', or 'image' on square images "
                                  "of the network size (per-pair crops would run the MiDaS encoder once per pair)")
    res = infer_depthnet_batched(model, torch.from_numpy(rgb), torch.from_numpy(masks), pairs=plist)
    if disp_select_method == "":
        return res["depth_order"], None
    # the reference takes the disparity of a call with empty masks; the disparity does not depend on the masks
    disp = res["disp"]
    clipped = torch.clip(disp, torch.quantile(disp, 0.05), torch.quantile(disp, 0.95))
    n = masks.shape[0]
    order = np.zeros((n, n), dtype=np.int64)
    for i, j in res["pairs"]:
        a = net_forward_midas_pretrained(disp, masks[i], masks[j], disp_select_method)
        if a == 0:
            order[i, j], order[j, i] = 1, 0
        elif a == 1:
            order[i, j], order[j, i] = 0, 1
        else:
            order[i, j] = order[j, i] = 2
    return order, clipped


# ---- metrics (host) -------------------------------------------------------------------------------
def eval_order_recall_precision_f1(order_matrix, gt_order_matrix, zd=0):
    """Binary recall / precision / F1 (x100) over entries whose ground truth is not -1
    (inference.py:794-802, sklearn semantics with zero_division=zd)."""
    sel = gt_order_matrix != -1
    y, p = gt_order_matrix[sel].reshape(-1), order_matrix[sel].reshape(-1)
    tp = float(np.sum((y == 1) & (p == 1)))
    fp = float(np.sum((y != 1) & (p == 1)))
    fn = float(np.sum((y == 1) & (p != 1)))
    recall = tp / (tp + fn) if tp + fn > 0 else float(zd)
    precision = tp / (tp + fp) if tp + fp > 0 else float(zd)
    # sklearn: zero_division only when there is nothing to count at all; tp = 0 with errors present is F1 = 0
    f1 = 2 * tp / (2 * tp + fp + fn) if 2 * tp + fp + fn > 0 else float(zd)
    return recall * 100, precision * 100, f1 * 100


def calculate_whdr(order_matrix, gt_order_matrix, score_matrix, mask):
    if mask.sum() == 0:
        return -1
    return ((gt_order_matrix[mask] != order_matrix[mask]) * score_matrix[mask]).sum() / score_matrix[mask].sum() * 100


def eval_depth_order_whdr(order_matrix, gt_order_ovl_count):
    """Weighted human disagreement rate per overlap x equality subset (inference.py:764-791):
    upper triangle, weights 2/count."""
    gt_order, gt_overlap, gt_count = gt_order_ovl_count
    iu = np.triu_indices_from(gt_order, k=1)
    g, ov, cnt, o = gt_order[iu], gt_overlap[iu], gt_count[iu], order_matrix[iu]
    score = 2 / cnt
    m_ovl = collections.OrderedDict([("ovlX", ov == 0), ("ovlO", ov == 1)])
    m_ovl["ovlOX"] = m_ovl["ovlX"] | m_ovl["ovlO"]
    m_eq = collections.OrderedDict([("eq", g == 2), ("neq", (g == 0) | (g == 1))])
    m_eq["all"] = m_eq["eq"] | m_eq["neq"]
    out = collections.defaultdict(list)
    for ko, mo in m_ovl.items():
        for ke, me in m_eq.items():
            out["%s_%s" % (ko, ke)].append(calculate_whdr(o, g, score, mo & me))
    return out
