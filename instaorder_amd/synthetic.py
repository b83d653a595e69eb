"""Deterministic synthetic inputs for the pairwise order-prediction path.

Everything here is seeded through ``numpy.random.RandomState`` so the same
arrays are regenerated bit-for-bit in this container (golden generation,
CPU tests) and on the GPU box (parity tests, smoke, bench).  Nothing is read
from disk and nothing depends on torch's RNG.

Tensor contract follows the reference datasets
(datasets/occ_order_dataset.py:202-279, datasets/depth_occ_order_dataset.py:197-240):
``rgb[B,3,S,S]`` fp32 (already mean/std normalised), ``modal1/modal2[B,1,S,S]``
fp32 in {0,1}, ``occ_order[B,2]`` fp32 in {0,1} laid out as
``[b_over_a, a_over_b]``, ``depth_order[B]`` int64 in {0,1,2},
``count[B]`` fp32, ``is_overlap[B]`` int64 in {0,1}.
"""
from collections import OrderedDict
import math

import numpy as np

LAYERS = (3, 4, 6, 3)          # ResNet-50 bottleneck counts (resnet_cls.py:266-267)
PLANES = (64, 128, 256, 512)
EXPANSION = 4


def state_specs(in_channels=5, num_classes=2):
    """(name, shape, kind) for every state_dict entry of ``resnet50_cls`` in
    registration order (resnet_cls.py:121-177).  kind is one of
    conv / bn_weight / bn_bias / bn_mean / bn_var / bn_count / fc_weight / fc_bias."""
    specs = []

    def conv(name, cout, cin, k):
        specs.append((name + ".weight", (cout, cin, k, k), "conv"))

    def bn(name, c):
        specs.append((name + ".weight", (c,), "bn_weight"))
        specs.append((name + ".bias", (c,), "bn_bias"))
        specs.append((name + ".running_mean", (c,), "bn_mean"))
        specs.append((name + ".running_var", (c,), "bn_var"))
        specs.append((name + ".num_batches_tracked", (), "bn_count"))

    conv("conv1", 64, in_channels, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip(PLANES, LAYERS)):
        for b in range(blocks):
            p = "layer%d.%d" % (li + 1, b)
            conv(p + ".conv1", planes, inplanes, 1)
            bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3)
            bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * EXPANSION, planes, 1)
            bn(p + ".bn3", planes * EXPANSION)
            if b == 0:
                conv(p + ".downsample.0", planes * EXPANSION, inplanes, 1)
                bn(p + ".downsample.1", planes * EXPANSION)
            inplanes = planes * EXPANSION
    feat = 512 * EXPANSION
    if isinstance(num_classes, (list, tuple)):
        specs.append(("fc_occ.weight", (num_classes[0], feat), "fc_weight"))
        specs.append(("fc_occ.bias", (num_classes[0],), "fc_bias"))
        specs.append(("fc_depth.weight", (num_classes[1], feat), "fc_weight"))
        specs.append(("fc_depth.bias", (num_classes[1],), "fc_bias"))
    else:
        specs.append(("fc.weight", (num_classes, feat), "fc_weight"))
        specs.append(("fc.bias", (num_classes,), "fc_bias"))
    return specs


def make_state_dict(seed, in_channels=5, num_classes=2, gain=0.02, prefix="", style="xavier"):
    """Seeded weights.  ``style='xavier'`` has the statistics of the reference
    initialisation (utils/common_utils.py:35-65: xavier-normal gain 0.02 on
    conv/linear, BN weight ~ N(1, 0.02), zero biases) -- the regime training starts
    from, where conv outputs have variance ~5e-5 and eval-mode BN with fresh running
    statistics wipes out the input.  ``style='kaiming'`` has the statistics of a
    trained-like network (He fan-out convs as resnet_cls.py:162-164, BN weight
    ~ N(1, 0.1), small BN biases, 1/sqrt(fan_in) heads): activations are O(1), so
    eval-mode outputs depend on the input.  Returns OrderedDict name -> ndarray
    (OIHW fp32; ``num_batches_tracked`` int64)."""
    rng = np.random.RandomState(seed)
    kaiming = style == "kaiming"
    out = OrderedDict()
    for name, shape, kind in state_specs(in_channels, num_classes):
        if kind == "conv":
            cout, cin, kh, kw = shape
            std = math.sqrt(2.0 / (cout * kh * kw)) if kaiming else gain * math.sqrt(2.0 / ((cin + cout) * kh * kw))
            v = (rng.standard_normal(shape) * std).astype(np.float32)
        elif kind == "fc_weight":
            std = math.sqrt(1.0 / shape[1]) if kaiming else gain * math.sqrt(2.0 / (shape[0] + shape[1]))
            v = (rng.standard_normal(shape) * std).astype(np.float32)
        elif kind == "bn_weight":
            v = (1.0 + (0.1 if kaiming else gain) * rng.standard_normal(shape)).astype(np.float32)
            if kaiming and (name.endswith("bn3.weight") or name.endswith("downsample.1.weight")):
                v *= np.float32(0.5)      # keep the residual sum from growing with depth
        elif kind == "bn_bias" and kaiming:
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind in ("bn_bias", "fc_bias", "bn_mean"):
            v = np.zeros(shape, np.float32)
        elif kind == "bn_var":
            v = np.ones(shape, np.float32)
        elif kind == "bn_count":
            v = np.zeros(shape, np.int64)
        else:
            raise ValueError(kind)
        out[prefix + name] = v
    return out


def make_checkpoint_state(seed, in_channels=5, num_classes=2, prefix="module."):
    """Seeded content of a mid-training checkpoint: trained-like weights, NON-trivial BatchNorm buffers (running mean /
    variance, per-layer step counters) and one momentum buffer per parameter, in ``named_parameters`` order -- what
    ``{'step','state_dict','optimizer'}`` of single_stage_model.py:66-72 carries.  Returns (state_dict, momentum list,
    lr, step): tests/golden/make_golden.py::case_checkpoint pours it into the REFERENCE model + its torch.optim.SGD and
    lets the reference write the file; the tests pour it into this package's model."""
    sd = make_state_dict(seed, in_channels, num_classes, prefix=prefix, style="kaiming")
    rng = np.random.RandomState(seed + 7919)
    mom = []
    for (name, shape, kind) in state_specs(in_channels, num_classes):
        k = prefix + name
        if kind == "bn_mean":
            sd[k] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == "bn_var":
            sd[k] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif kind == "bn_count":
            sd[k] = np.array(rng.randint(1, 1000), dtype=np.int64).reshape(shape)
        else:
            mom.append((1e-3 * rng.standard_normal(shape)).astype(np.float32))
    return sd, mom, 3.3e-4, 4321


def _mask(rng, S):
    """One filled rectangle or ellipse covering 5-40 % of an SxS image."""
    area = rng.uniform(0.05, 0.40) * S * S
    aspect = rng.uniform(0.5, 2.0)
    h = min(S, max(2, int(round(math.sqrt(area * aspect)))))
    w = min(S, max(2, int(round(area / h))))
    top = rng.randint(0, S - h + 1)
    left = rng.randint(0, S - w + 1)
    m = np.zeros((S, S), np.float32)
    if rng.rand() < 0.5:
        m[top:top + h, left:left + w] = 1.0
    else:
        yy, xx = np.mgrid[0:S, 0:S]
        cy, cx = top + (h - 1) / 2.0, left + (w - 1) / 2.0
        m[((yy - cy) / (h / 2.0)) ** 2 + ((xx - cx) / (w / 2.0)) ** 2 <= 1.0] = 1.0
    return m


def make_pair_batch(seed, B, S=256):
    """One training batch of B instance pairs (dict of ndarrays)."""
    rng = np.random.RandomState(seed)
    rgb = rng.standard_normal((B, 3, S, S)).astype(np.float32)
    modal1 = np.stack([_mask(rng, S) for _ in range(B)])[:, None]
    modal2 = np.stack([_mask(rng, S) for _ in range(B)])[:, None]
    occ_order = (rng.rand(B, 2) < 0.3).astype(np.float32)
    depth_order = rng.randint(0, 3, size=B).astype(np.int64)
    is_overlap = (rng.rand(B) < 0.5).astype(np.int64)
    count = np.full((B,), 2.0, np.float32)
    return dict(rgb=rgb, modal1=modal1, modal2=modal2, occ_order=occ_order,
                depth_order=depth_order, count=count, is_overlap=is_overlap)


def make_images(seed, n_images, n_inst, S=256):
    """Synthetic 'validation images': uint8 RGB image [S,S,3], modal masks
    [n_inst,S,S] uint8, xywh boxes, and random ground-truth order matrices
    (occlusion in {0,1} with -1 on nothing; depth in {0,1,2}; overlap; count)."""
    rng = np.random.RandomState(seed)
    items = []
    for _ in range(n_images):
        image = rng.randint(0, 256, size=(S, S, 3)).astype(np.uint8)
        modal = np.stack([_mask(rng, S) for _ in range(n_inst)]).astype(np.uint8)
        bboxes = []
        for m in modal:
            ys, xs = np.where(m > 0)
            bboxes.append([xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1])
        bboxes = np.asarray(bboxes, np.int64)
        gt_occ = (rng.rand(n_inst, n_inst) < 0.3).astype(np.int64)
        np.fill_diagonal(gt_occ, 0)
        gt_depth = np.zeros((n_inst, n_inst), np.int64)
        gt_overlap = np.zeros((n_inst, n_inst), np.int64)
        gt_count = np.ones((n_inst, n_inst), np.int64) * 2
        for i in range(n_inst):
            for j in range(i + 1, n_inst):
                d = rng.randint(0, 3)
                gt_depth[i, j] = d
                gt_depth[j, i] = d if d == 2 else 1 - d
                gt_overlap[i, j] = gt_overlap[j, i] = int(rng.rand() < 0.5)
                gt_count[i, j] = gt_count[j, i] = rng.randint(1, 4)
        items.append(dict(image=image, modal=modal, bboxes=bboxes, gt_occ=gt_occ,
                          gt_depth=gt_depth, gt_overlap=gt_overlap, gt_count=gt_count))
    return items


class SyntheticReader(object):
    """Stand-in for ``datasets/reader.py:InstaOrderDataset`` (the COCO / InstaOrder json is not available): seeded
    scenes with non-square images of different sizes, rectangle / ellipse instances, and InstaOrder-style annotations
    -- an occlusion matrix ({0,1}, bidirectional pairs allowed, some images without any occluding pair) and depth
    relations "i<j" / "i=j" with overlap flag and count.  Implements the reader methods the dataset classes call
    (occ_order_dataset.py:182-200, depth_occ_order_dataset.py:150-160, 197-205) plus ``load_image``."""

    def __init__(self, seed, n_images=6, n_inst=5, max_side=160, min_side=72, empty_every=4, rule="random"):
        """rule 'random': occlusion entries are coin flips (fixtures); 'lower': i occludes j iff their boxes
        intersect and the centre of i lies lower in the image -- a relation a network can learn from the two masks
        (tools/train_synthetic.py)."""
        rng = np.random.RandomState(seed)
        self.scenes = []
        for k in range(n_images):
            H, W = int(rng.randint(min_side, max_side + 1)), int(rng.randint(min_side, max_side + 1))
            image = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
            modal = np.zeros((n_inst, H, W), np.uint8)
            for i in range(n_inst):
                h, w = int(rng.randint(8, H // 2 + 1)), int(rng.randint(8, W // 2 + 1))
                top, left = int(rng.randint(0, H - h + 1)), int(rng.randint(0, W - w + 1))
                if rng.rand() < 0.5:
                    modal[i, top:top + h, left:left + w] = 1
                else:
                    yy, xx = np.mgrid[0:H, 0:W]
                    cy, cx = top + (h - 1) / 2.0, left + (w - 1) / 2.0
                    modal[i][((yy - cy) / (h / 2.0)) ** 2 + ((xx - cx) / (w / 2.0)) ** 2 <= 1.0] = 1
            bboxes = []
            for m in modal:
                ys, xs = np.where(m > 0)
                bboxes.append([xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1])
            occ = (rng.rand(n_inst, n_inst) < 0.3).astype(np.int64)
            np.fill_diagonal(occ, 0)
            if rule == "lower":
                bb = np.asarray(bboxes, np.float64)
                cy = bb[:, 1] + bb[:, 3] / 2.0
                for i in range(n_inst):
                    for j in range(n_inst):
                        touch = (bb[i, 0] < bb[j, 0] + bb[j, 2] and bb[j, 0] < bb[i, 0] + bb[i, 2] and
                                 bb[i, 1] < bb[j, 1] + bb[j, 3] and bb[j, 1] < bb[i, 1] + bb[i, 3])
                        occ[i, j] = int(i != j and touch and cy[i] > cy[j])
            if empty_every and k % empty_every == empty_every - 1:
                occ[:] = 0                                   # an image without occluding pairs: the dataset re-draws
            depth = -np.ones((n_inst, n_inst), np.int64)
            overlap = -np.ones((n_inst, n_inst), np.int64)
            count = -np.ones((n_inst, n_inst), np.int64)
            relations = []
            for i in range(n_inst):
                for j in range(i + 1, n_inst):
                    if rng.rand() < 0.25:
                        continue                             # not every pair is annotated
                    a, b = (i, j) if rng.rand() < 0.5 else (j, i)
                    equal = rng.rand() < 0.2
                    ov, ct = int(rng.rand() < 0.5), int(rng.randint(1, 4))
                    if equal:
                        depth[a, b] = depth[b, a] = 2
                    else:
                        depth[a, b], depth[b, a] = 1, 0      # a closer than b
                    overlap[a, b] = overlap[b, a] = ov
                    count[a, b] = count[b, a] = ct
                    relations.append("%d%s%d" % (a, "=" if equal else "<", b))
            self.scenes.append(dict(image=image, modal=modal, bboxes=np.asarray(bboxes, np.int64),
                                    category=rng.randint(1, 80, size=n_inst).astype(np.int64), occ=occ, depth=depth,
                                    overlap=overlap, count=count, relations=relations))
        self.depth_all = [(k, r) for k, sc in enumerate(self.scenes) for r in sc["relations"]]

    def get_image_length(self):
        return len(self.scenes)

    def get_geometric_length(self):
        return len(self.depth_all)

    def get_imgId_and_depth(self, i):
        return self.depth_all[i]

    def get_image_instances(self, idx, with_gt=False, **kw):
        sc = self.scenes[idx]
        return sc["modal"], sc["category"], sc["bboxes"], None, "scene%d" % idx

    def get_gt_ordering(self, idx, type="occlusion", rm_bidirec=0, rm_overlap=0):
        sc = self.scenes[idx]
        if type == "occlusion":
            return sc["occ"].copy()
        return [sc["depth"].copy(), sc["overlap"].copy(), sc["count"].copy()]

    def load_image(self, image_fn):
        return self.scenes[int(str(image_fn).rsplit("scene", 1)[1])]["image"]


def image_mode_inputs(image, modal, input_size):
    """The reference's ``patch_or_image == 'image'`` preprocessing
    (inference.py:467-482) for an image that is already square and already
    ``input_size`` wide, where every resize is the identity: returns
    rgb[1,3,S,S] fp32 normalised with the ImageNet mean/std
    (utils/data_utils.py:9-10,28-34) and the float masks [N,S,S]."""
    assert image.shape[0] == image.shape[1] == input_size
    mean = np.asarray([0.485, 0.456, 0.406], np.float32)[:, None, None]
    std = np.asarray([0.229, 0.224, 0.225], np.float32)[:, None, None]
    rgb = image.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    rgb = ((rgb - mean) / std)[None]
    return rgb.astype(np.float32), modal.astype(np.float32)


# ---- MiDaS-based nets (InstaDepthNet_od / _d) -----------------------------------------------------------------------
def make_spec_state_dict(seed, spec, prefix=""):
    """Seeded, well-conditioned values for an arbitrary module: ``spec`` is a list of (key, shape, alias_of) in
    ``state_dict`` order, alias_of = the first key that shares the tensor (the order branches of InstaDepthNet_*
    expose conv1 / bn1 under two names) or None.  Filters: N(0, sqrt(2 / fan_in)); the last BatchNorm of every
    bottleneck (``bn3``) gets weight ~ 0.2 so that rounding noise adds up instead of multiplying through 33 + 16
    blocks; other BN weights ~ N(1, 0.05); BN biases / conv biases ~ N(0, 0.05); running_mean ~
    N(0, 0.1), running_var ~ U(0.5, 1.5).  Every tensor has its own RandomState(seed, crc32(key))."""
    import zlib
    out = {}
    for key, shape, alias in spec:
        if alias is not None:
            out[prefix + key] = out[prefix + alias]
            continue
        rng = np.random.RandomState([seed & 0x7FFFFFFF, zlib.crc32(key.encode()) & 0x7FFFFFFF])
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            v = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            v = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif len(shape) == 2:
            v = (rng.standard_normal(shape) * np.sqrt(1.0 / shape[1])).astype(np.float32)
        elif leaf == "weight":          # BatchNorm weight
            centre = 0.2 if ".bn3." in "." + key else 1.0
            v = (centre + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:                           # biases
            v = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        out[prefix + key] = v
    return out


def make_depth_batch(seed, B, S):
    """Inputs of InstaDepthNet_*.set_input: the pair batch plus what the disparity losses look at (masks with an
    interior after erosion, is_overlap with both values present)."""
    b = make_pair_batch(seed, B, S)
    b["is_overlap"] = (np.arange(B) % 2).astype(np.int64)
    b["depth_order"] = (np.arange(B) % 3).astype(np.int64)
    return b
