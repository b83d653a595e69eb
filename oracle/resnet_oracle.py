"""CPU oracle for the InstaOrder pairwise order-prediction hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 restatement of the
reference's algorithm for the hot path; it exists so that the HIP path can be
checked against something that travels to the GPU box (the reference's Python
does not).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
``instaorder_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from /root/reference (in the build container only), runs it on seeded
inputs and commits the outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those
vectors.

Each function cites the reference file:line it restates (paths relative to the
reference root).
"""
from bisect import bisect_right
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5          # nn.BatchNorm2d default, resnet_cls.py:142
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# state handling
# --------------------------------------------------------------------------
def state_from_numpy(sd, prefix=""):
    """OrderedDict[str, ndarray] -> OrderedDict[str, Tensor] (strips ``prefix``)."""
    out = OrderedDict()
    for k, v in sd.items():
        if prefix and k.startswith(prefix):
            k = k[len(prefix):]
        out[k] = torch.from_numpy(np.array(v, copy=True, order="C"))
    return out


def param_names(state):
    """Trainable entries in ``nn.Module.parameters()`` order (everything except
    BN running statistics / counters)."""
    return [k for k in state if not (k.endswith("running_mean") or k.endswith("running_var")
                                     or k.endswith("num_batches_tracked"))]


# --------------------------------------------------------------------------
# backbone: models/backbone/resnet_cls.py
# --------------------------------------------------------------------------
def _bn(x, state, name, training):
    """nn.BatchNorm2d (resnet_cls.py:142, 87-92, 189): batch statistics with biased
    variance for the normalisation, running estimates updated with momentum 0.1
    and the unbiased variance, counter incremented, when training."""
    rm, rv = state[name + ".running_mean"], state[name + ".running_var"]
    if training:
        state[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, state[name + ".weight"], state[name + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def _bottleneck(x, state, p, stride, has_down, training):
    """Bottleneck.forward (resnet_cls.py:96-116); stride sits on the 3x3 (:88)."""
    out = F.conv2d(x, state[p + ".conv1.weight"])
    out = F.relu(_bn(out, state, p + ".bn1", training))
    out = F.conv2d(out, state[p + ".conv2.weight"], stride=stride, padding=1)
    out = F.relu(_bn(out, state, p + ".bn2", training))
    out = F.conv2d(out, state[p + ".conv3.weight"])
    out = _bn(out, state, p + ".bn3", training)
    if has_down:
        idt = F.conv2d(x, state[p + ".downsample.0.weight"], stride=stride)
        idt = _bn(idt, state, p + ".downsample.1", training)
    else:
        idt = x
    return F.relu(out + idt)


def resnet_forward(state, x, training):
    """ResNet.forward (resnet_cls.py:203-222).  Returns logits ``[B,C]`` or the pair
    ``(occ[B,2], depth[B,3])`` when the state carries the two heads (:153-157)."""
    x = F.conv2d(x, state["conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(x, state, "bn1", training))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, blocks in enumerate(LAYERS):
        for b in range(blocks):
            stride = 2 if (b == 0 and li > 0) else 1
            x = _bottleneck(x, state, "layer%d.%d" % (li + 1, b), stride, b == 0, training)
    x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    if "fc_occ.weight" in state:
        return (F.linear(x, state["fc_occ.weight"], state["fc_occ.bias"]),
                F.linear(x, state["fc_depth.weight"], state["fc_depth.bias"]))
    return F.linear(x, state["fc.weight"], state["fc.bias"])


# --------------------------------------------------------------------------
# label mirroring: models/supervised_order.py
# --------------------------------------------------------------------------
def mirror_occ(occ_order):
    """occ_order2 = columns of occ_order1 swapped (supervised_order.py:514-516, 46-48)."""
    return occ_order[:, [1, 0]]


def mirror_depth(depth_order):
    """0<->1, 2 stays 2 (supervised_order.py:39-42); also OrderNet's 3 stays 3 (:456-460)."""
    d2 = 1 - depth_order
    d2 = torch.where(depth_order >= 2, depth_order, d2)
    return d2


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def loss_o(state, batch, world_size, training):
    """InstaOrderNet_o.step / forward_only loss (supervised_order.py:518-543):
    sigmoid then nn.BCELoss (mean over B*2) for both mask orders, summed, divided
    by the world size.  Returns (loss, out1, out2, logits1, logits2)."""
    x1 = torch.cat([batch["modal1"], batch["modal2"], batch["rgb"]], 1)
    x2 = torch.cat([batch["modal2"], batch["modal1"], batch["rgb"]], 1)
    z1 = resnet_forward(state, x1, training)
    z2 = resnet_forward(state, x2, training)
    o1, o2 = torch.sigmoid(z1), torch.sigmoid(z2)
    y1 = batch["occ_order"]
    loss = (F.binary_cross_entropy(o1, y1) + F.binary_cross_entropy(o2, mirror_occ(y1))) / world_size
    return loss, o1, o2, z1, z2


def _ce_on_probs(q, t):
    """nn.CrossEntropyLoss applied to *probabilities* (log-softmax of a softmax),
    as the reference does (supervised_order.py:54, 68-72)."""
    return F.cross_entropy(q, t)


def loss_od(state, batch, world_size, training, overlap_weight, distinct_weight):
    """InstaOrderNet_od.step + calculate_loss (supervised_order.py:59-95).
    Returns (loss, loss_occ, loss_depth, (occ1, dep1, occ2, dep2) raw logits)."""
    x1 = torch.cat([batch["modal1"], batch["modal2"], batch["rgb"]], 1)
    x2 = torch.cat([batch["modal2"], batch["modal1"], batch["rgb"]], 1)
    zo1, zd1 = resnet_forward(state, x1, training)
    zo2, zd2 = resnet_forward(state, x2, training)
    q1, q2 = F.softmax(zd1, dim=1), F.softmax(zd2, dim=1)
    p1, p2 = torch.sigmoid(zo1), torch.sigmoid(zo2)
    d1 = batch["depth_order"]
    d2 = mirror_depth(d1)
    ov = batch["is_overlap"] == 1
    di = batch["is_overlap"] == 0
    l_ov = torch.zeros(())
    l_di = torch.zeros(())
    if int(ov.sum()) > 0:
        l_ov = _ce_on_probs(q1[ov], d1[ov]) + _ce_on_probs(q2[ov], d2[ov])
    if int(di.sum()) > 0:
        l_di = _ce_on_probs(q1[di], d1[di]) + _ce_on_probs(q2[di], d2[di])
    loss_depth = l_ov * overlap_weight + l_di * distinct_weight
    y1 = batch["occ_order"]
    loss_occ = F.binary_cross_entropy(p1, y1) + F.binary_cross_entropy(p2, mirror_occ(y1))
    loss = (loss_depth + loss_occ) / world_size
    return loss, loss_occ, loss_depth, (zo1, zd1, zo2, zd2)


def loss_softmax_ce(state, batch, world_size, training, weights=None):
    """OrderNet.step / forward_only (supervised_order.py:465-493) and
    InstaOrderNet_d.forward_only (:394-409): softmax then CrossEntropyLoss on both
    mask orders, /world_size; class ids come from ``depth_order``.
    InstaOrderNet_d.step (:413-438) instead weights the overlap / distinct subsets
    (``weights=(overlap_weight, distinct_weight)``), each a mean over its own size,
    skipped when empty -- its forward_only does NOT (reference quirk, kept)."""
    x1 = torch.cat([batch["modal1"], batch["modal2"], batch["rgb"]], 1)
    x2 = torch.cat([batch["modal2"], batch["modal1"], batch["rgb"]], 1)
    z1 = resnet_forward(state, x1, training)
    z2 = resnet_forward(state, x2, training)
    t1 = batch["depth_order"]
    t2 = mirror_depth(t1)
    q1, q2 = F.softmax(z1, 1), F.softmax(z2, 1)
    if weights is None:
        loss = (_ce_on_probs(q1, t1) + _ce_on_probs(q2, t2)) / world_size
    else:
        ov = batch["is_overlap"] == 1
        di = batch["is_overlap"] == 0
        l_ov = torch.zeros(())
        l_di = torch.zeros(())
        if int(ov.sum()) > 0:
            l_ov = _ce_on_probs(q1[ov], t1[ov]) + _ce_on_probs(q2[ov], t2[ov])
        if int(di.sum()) > 0:
            l_di = _ce_on_probs(q1[di], t1[di]) + _ce_on_probs(q2[di], t2[di])
        loss = (l_ov * weights[0] + l_di * weights[1]) / world_size
    return loss, z1, z2


# --------------------------------------------------------------------------
# optimiser / scheduler
# --------------------------------------------------------------------------
def sgd_step(state, grads, momentum_bufs, lr, weight_decay, momentum=0.9):
    """torch.optim.SGD as configured at single_stage_model.py:35-38: one parameter
    group (BN affine and biases included), coupled L2, dampening 0, no Nesterov;
    the momentum buffer starts as the first (decayed) gradient."""
    for k, g in grads.items():
        p = state[k]
        d = g.add(p, alpha=weight_decay)
        if k not in momentum_bufs:
            momentum_bufs[k] = d.clone()
        else:
            momentum_bufs[k].mul_(momentum).add_(d)
        p.add_(momentum_bufs[k], alpha=-lr)


def step_lr(it, base_lr, milestones, lr_mults, warmup_lr=(), warmup_steps=()):
    """StepLRScheduler._get_new_lr with _WarmUpLRScheduler._get_warmup_lr
    (utils/scheduler.py:58-109) for a single param group whose initial lr is
    ``base_lr``: piecewise-linear warm-up through (warmup_steps, warmup_lr), then
    base * prod(lr_mults[:number of milestones <= it])."""
    pos = bisect_right(list(warmup_steps), it)
    if pos < len(warmup_steps):
        if pos == 0:
            cur = base_lr + it * (warmup_lr[0] - base_lr) / warmup_steps[0]
        else:
            cur = warmup_lr[pos - 1] + (it - warmup_steps[pos - 1]) * \
                (warmup_lr[pos] - warmup_lr[pos - 1]) / (warmup_steps[pos] - warmup_steps[pos - 1])
        return (cur / base_lr) * base_lr
    cum = [1.0]
    for m in lr_mults:
        cum.append(cum[-1] * m)
    k = bisect_right(list(milestones), it)
    if len(warmup_lr) == 0:
        scale = cum[k]
    else:
        scale = warmup_lr[-1] * cum[k] / base_lr
    return base_lr * scale


def train_step(state, momentum_bufs, batch, algo, lr, weight_decay, world_size=1,
               overlap_weight=0.1, distinct_weight=0.9, allreduce=None):
    """One ``step()`` of the wrapper classes: forward x2, loss, backward, gradient
    SUM all-reduce (utils/distributed_utils.py:27-31), SGD.  ``allreduce`` is a
    callable applied to every gradient tensor in place (None on one rank).
    Returns (dict of scalar losses, OrderedDict of gradients)."""
    names = param_names(state)
    leaves = []
    for k in names:
        state[k] = state[k].detach().requires_grad_(True)
        leaves.append(state[k])
    tb = {k: torch.as_tensor(v) for k, v in batch.items()}
    if algo == "InstaOrderNet_o":
        loss = loss_o(state, tb, world_size, True)[0]
        logs = {"loss": loss.detach()}
    elif algo == "InstaOrderNet_od":
        loss, lo, ld, _ = loss_od(state, tb, world_size, True, overlap_weight, distinct_weight)
        logs = {"loss": loss.detach(), "loss_occ": lo.detach(), "loss_depth": ld.detach()}
    elif algo == "InstaOrderNet_d":
        loss = loss_softmax_ce(state, tb, world_size, True, (overlap_weight, distinct_weight))[0]
        logs = {"loss": loss.detach()}
    elif algo == "OrderNet":
        loss = loss_softmax_ce(state, tb, world_size, True)[0]
        logs = {"loss": loss.detach()}
    else:
        raise ValueError(algo)
    gl = torch.autograd.grad(loss, leaves)
    grads = OrderedDict()
    for k, g in zip(names, gl):
        state[k] = state[k].detach()
        grads[k] = g.detach().clone()
        if allreduce is not None:
            allreduce(grads[k])
    with torch.no_grad():
        sgd_step(state, grads, momentum_bufs, lr, weight_decay)
    return logs, grads


# --------------------------------------------------------------------------
# inference decision rules and metrics: inference.py
# --------------------------------------------------------------------------
def decide_occ(out1, out2):
    """net_forward_occ (inference.py:196-214): average the two directions, threshold
    at 0.5.  ``out*`` are sigmoid probabilities [B,2].  Returns two bool arrays
    (first over second, second over first)."""
    a = (out1[:, 1] + out2[:, 0]) / 2
    b = (out1[:, 0] + out2[:, 1]) / 2
    return (a > 0.5), (b > 0.5)


def decide_depth(q1, q2):
    """net_forward_occ_depth / net_forward_depth (inference.py:140-193): class
    0 = first closer, 1 = first farther, 2 = equal; argmax of direction-averaged
    softmax probabilities (first maximum wins, as np.argmax)."""
    closer = (q1[:, 0] + q2[:, 1]) / 2
    farther = (q1[:, 1] + q2[:, 0]) / 2
    equal = (q1[:, 2] + q2[:, 2]) / 2
    return torch.stack([closer, farther, equal], 1).argmax(1)


def order_matrices(n, pairs, i_over_j, j_over_i, depth_idx=None):
    """infer_order_sup_occ / infer_order_sup_occ_depth (inference.py:349-512):
    fill the N x N matrices from per-pair decisions over the upper triangle."""
    occ = np.zeros((n, n), np.int64)
    dep = np.zeros((n, n), np.int64)
    for k, (i, j) in enumerate(pairs):
        if bool(i_over_j[k]):
            occ[i, j] = 1
        if bool(j_over_i[k]):
            occ[j, i] = 1
        if depth_idx is not None:
            d = int(depth_idx[k])
            if d == 0:
                dep[i, j], dep[j, i] = 1, 0
            elif d == 1:
                dep[i, j], dep[j, i] = 0, 1
            else:
                dep[i, j] = dep[j, i] = 2
    return occ, dep


def recall_precision_f1(order, gt, zero_division=0):
    """eval_order_recall_precision_f1 (inference.py:794-802): sklearn binary scores
    over entries whose ground truth is not -1, as percentages."""
    sel = gt != -1
    y, p = gt[sel].reshape(-1), order[sel].reshape(-1)
    tp = float(((y == 1) & (p == 1)).sum())
    fp = float(((y != 1) & (p == 1)).sum())
    fn = float(((y == 1) & (p != 1)).sum())
    rec = tp / (tp + fn) if tp + fn > 0 else float(zero_division)
    pre = tp / (tp + fp) if tp + fp > 0 else float(zero_division)
    # sklearn applies zero_division to F1 only when tp + fp + fn == 0; tp = 0 beside errors is F1 = 0
    f1 = 2 * tp / (2 * tp + fp + fn) if 2 * tp + fp + fn > 0 else float(zero_division)
    return rec * 100, pre * 100, f1 * 100


def whdr(order, gt_order, gt_overlap, gt_count):
    """eval_depth_order_whdr + calculate_whdr (inference.py:757-791): upper triangle
    only, weights 2/count, split by overlap x {eq, neq, all}; -1 for empty subsets."""
    iu = np.triu_indices_from(gt_order, k=1)
    o, g, ov, cnt = order[iu], gt_order[iu], gt_overlap[iu], gt_count[iu]
    score = 2 / cnt
    m_ovl = {"ovlX": ov == 0, "ovlO": ov == 1}
    m_ovl["ovlOX"] = m_ovl["ovlX"] | m_ovl["ovlO"]
    m_eq = {"eq": g == 2, "neq": (g == 0) | (g == 1)}
    m_eq["all"] = m_eq["eq"] | m_eq["neq"]
    out = {}
    for ko, mo in m_ovl.items():
        for ke, me in m_eq.items():
            m = mo & me
            if m.sum() == 0:
                out[ko + "_" + ke] = -1
            else:
                out[ko + "_" + ke] = float(((g[m] != o[m]) * score[m]).sum() / score[m].sum() * 100)
    return out
