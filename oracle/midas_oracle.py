"""CPU oracle for the MiDaS-based nets -- TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

Plain-PyTorch functional restatement of ``InstaDepthNet_od.forward`` / ``InstaDepthNet_d.forward``
(midas/midas_net.py:60-113, 166-212), of their building blocks (midas/blocks.py:71-82 encoder arrangement,
:121-160 ResidualConvUnit with its in-place ReLU, :163-195 FeatureFusionBlock, :97-118 Interpolate; the grouped
Bottleneck of models/backbone/resnet_cls.py:75-116, 309-320) and of the wrapper losses
(models/supervised_order.py:129-184, 207-235).  It works on a flat ``state`` dict {key: tensor} with the reference's
state_dict keys (without the ``module.`` prefix).  Parity: pinned by tests/golden/depthnet_*.npz, which
tests/golden/make_golden.py produces by running the unmodified reference (torch.hub.load patched to the reference's own
resnext101_32x8d, as SURVEY.md 8(c) prescribes; pretrained MiDaS weights do not exist here).
"""
import numpy as np
import torch
import torch.nn.functional as F


def state_from_numpy(sd, prefix="module.", dtype=torch.float32, requires_grad=True):
    out, seen = {}, {}
    for k, v in sd.items():
        kk = k[len(prefix):] if k.startswith(prefix) else k
        if id(v) in seen:                       # aliased keys share one tensor
            out[kk] = out[seen[id(v)]]
            continue
        t = torch.from_numpy(np.ascontiguousarray(v))
        if t.is_floating_point():
            t = t.to(dtype).clone()
            leaf = kk.rsplit(".", 1)[-1]
            t.requires_grad_(requires_grad and leaf in ("weight", "bias"))
        out[kk] = t
        seen[id(v)] = kk
    return out


def _bn(st, name, x, training):
    return F.batch_norm(x, st[name + ".running_mean"], st[name + ".running_var"], st[name + ".weight"], st[name + ".bias"],
                        training, 0.1, 1e-5)


def _bottleneck(st, name, x, stride, groups, training):
    out = F.relu(_bn(st, name + ".bn1", F.conv2d(x, st[name + ".conv1.weight"]), training))
    out = F.conv2d(out, st[name + ".conv2.weight"], stride=stride, padding=1, groups=groups)
    out = F.relu(_bn(st, name + ".bn2", out, training))
    out = _bn(st, name + ".bn3", F.conv2d(out, st[name + ".conv3.weight"]), training)
    identity = x
    if name + ".downsample.0.weight" in st:
        identity = _bn(st, name + ".downsample.1", F.conv2d(x, st[name + ".downsample.0.weight"], stride=stride), training)
    return F.relu(out + identity)


def _stage(st, name, x, nblocks, stride, groups, training):
    for i in range(nblocks):
        x = _bottleneck(st, "%s.%d" % (name, i), x, stride if i == 0 else 1, groups, training)
    return x


def _layer1(st, net, x, nblocks, groups, training):
    """layer1 = Sequential(conv1, bn1, relu, maxpool, layer1) (midas/blocks.py:73-76, midas_net.py:146-147)."""
    y = F.conv2d(x, st[net + ".layer1.0.weight"], stride=2, padding=3)
    y = F.relu(_bn(st, net + ".layer1.1", y, training))
    y = F.max_pool2d(y, 3, 2, 1)
    return _stage(st, net + ".layer1.4", y, nblocks, 1, groups, training)


def _rcu(st, name, x):
    r = F.relu(x)                                   # nn.ReLU(inplace=True) on the input: the skip adds relu(x)
    out = F.conv2d(r, st[name + ".conv1.weight"], st[name + ".conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), st[name + ".conv2.weight"], st[name + ".conv2.bias"], padding=1)
    return out + r


def _fusion(st, name, *xs):
    out = xs[0]
    if len(xs) == 2:
        out = out + _rcu(st, name + ".resConfUnit1", xs[1])
    out = _rcu(st, name + ".resConfUnit2", out)
    return F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)


def _branch(st, net, fc, masks, l1, l2, l3, training):
    f1 = _layer1(st, net, masks, 3, 1, training)
    f2 = _stage(st, net + ".layer2", f1 + l1, 4, 2, 1, training)
    f3 = _stage(st, net + ".layer3", f2 + l2, 6, 2, 1, training)
    f4 = _stage(st, net + ".layer4", f3 + l3, 3, 2, 1, training)
    pooled = torch.flatten(F.adaptive_avg_pool2d(f4, 1), 1)
    return F.linear(pooled, st[fc + ".weight"], st[fc + ".bias"])


def forward(st, img, mask1, mask2, training, variant="od", non_negative=True):
    """-> (disp[B,H,W], depth_order[B,3], occ_order[B,2] | None).  Running statistics in ``st`` are advanced in place in
    training mode, exactly as the modules would."""
    l1 = _layer1(st, "pretrained", img, 3, 32, training)
    l2 = _stage(st, "pretrained.layer2", l1, 4, 2, 32, training)
    l3 = _stage(st, "pretrained.layer3", l2, 23, 2, 32, training)
    l4 = _stage(st, "pretrained.layer4", l3, 3, 2, 32, training)
    rn = [F.conv2d(l, st["scratch.layer%d_rn.weight" % (i + 1)], padding=1) for i, l in enumerate((l1, l2, l3, l4))]
    p4 = _fusion(st, "scratch.refinenet4", rn[3])
    p3 = _fusion(st, "scratch.refinenet3", p4, rn[2])
    p2 = _fusion(st, "scratch.refinenet2", p3, rn[1])
    p1 = _fusion(st, "scratch.refinenet1", p2, rn[0])
    y = F.conv2d(p1, st["scratch.output_conv.0.weight"], st["scratch.output_conv.0.bias"], padding=1)
    y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.relu(F.conv2d(y, st["scratch.output_conv.2.weight"], st["scratch.output_conv.2.bias"], padding=1))
    y = F.conv2d(y, st["scratch.output_conv.4.weight"], st["scratch.output_conv.4.bias"])
    if non_negative:
        y = F.relu(y)
    disp = torch.squeeze(y, dim=1)
    masks = torch.cat([mask1, mask2], 1)
    if variant == "od":
        dep = _branch(st, "do_net", "depth_fc", masks, l1, l2, l3, training)
        occ = _branch(st, "oo_net", "occ_fc", masks, l1, l2, l3, training)
        return disp, dep, occ
    dep = _branch(st, "gdo_net", "fc", masks, l1, l2, l3, training)
    return disp, dep, None


# ---- losses (models/supervised_order.py:129-184, 207-235) -----------------------------------------------------------------
def smooth_loss(disp, img):
    mn = disp.min(2, True)[0].min(3, True)[0]
    mx = disp.max(2, True)[0].max(3, True)[0]
    disp = (disp - mn) / (mx + 1e-7)
    disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def disp_order_count(disp1, disp2, modal1, modal2, depth_order1, is_overlap):
    from scipy import ndimage
    total = 0
    for bb in range(modal1.shape[0]):
        if int(is_overlap[bb]) != 0:
            continue
        e1 = torch.from_numpy(ndimage.binary_erosion(modal1[bb, 0].numpy()).astype(bool))
        e2 = torch.from_numpy(ndimage.binary_erosion(modal2[bb, 0].numpy()).astype(bool))
        d1, d2 = disp1[bb, 0].detach(), disp2[bb, 0].detach()
        if int(depth_order1[bb]) == 0:
            total += (d1[e1] <= d1[e2].max()).sum() + (d1[e1].min() <= d1[e2]).sum()
            total += (d2[e1] >= d2[e2].max()).sum() + (d2[e1].min() >= d2[e2]).sum()
        elif int(depth_order1[bb]) == 1:
            total += (d1[e1] >= d1[e2].max()).sum() + (d1[e1].min() >= d1[e2]).sum()
            total += (d2[e1] <= d2[e2].max()).sum() + (d2[e1].min() <= d2[e2]).sum()
    return float(total) / float(disp1.shape[2] * disp1.shape[3])


def losses(outs1, outs2, batch, params, world_size=1, variant="od"):
    """-> (logs dict, differentiable total).  outs = forward(...) of the two mask orders."""
    disp1, dep1, occ1 = outs1
    disp2, dep2, occ2 = outs2
    disp1, disp2 = disp1.unsqueeze(1), disp2.unsqueeze(1)
    dep1, dep2 = F.softmax(dep1, 1), F.softmax(dep2, 1)
    d1 = batch["depth_order"]
    d2 = d1.clone()
    d2[d1 == 0] = 1
    d2[d1 == 1] = 0
    ov, di = batch["is_overlap"] == 1, batch["is_overlap"] == 0
    ce = torch.nn.CrossEntropyLoss()
    lo = ld = 0
    if ov.sum() > 0:
        lo = (ce(dep1[ov], d1[ov]) + ce(dep2[ov], d2[ov])) * params["overlap_weight"] / world_size
    if di.sum() > 0:
        ld = (ce(dep1[di], d1[di]) + ce(dep2[di], d2[di])) * params["distinct_weight"] / world_size
    locc = 0
    if variant == "od" and params["occ_order_weight"] != 0:
        o1 = batch["occ_order"]
        o2 = o1[:, [1, 0]]
        bce = torch.nn.BCELoss()
        locc = (bce(torch.sigmoid(occ1), o1) + bce(torch.sigmoid(occ2), o2)) / world_size
    ls = 0
    if params["smooth_weight"] != 0:
        ls = (smooth_loss(disp1, batch["rgb"]) + smooth_loss(disp2, batch["rgb"])) * params["smooth_weight"] / world_size
    ldo = 0
    if params["dorder_weight"] != 0:
        ldo = disp_order_count(disp1, disp2, batch["modal1"], batch["modal2"], d1, batch["is_overlap"]) \
            * params["dorder_weight"] / world_size
    total = lo + ld + locc + ls + ldo
    logs = {"loss_overlap": lo, "loss_distinct": ld, "loss_smooth": ls, "loss_disp_order": ldo}
    if variant == "od":
        logs["loss_occ"] = locc
    return logs, total
