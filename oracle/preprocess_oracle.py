"""CPU oracle for the input pipeline in front of the order networks -- TEST INFRASTRUCTURE ONLY (imported by tests/
and tests/golden/make_golden.py, nothing else).

Two layers:

1. ``resize`` -- numpy restatement of ``cv2.resize`` for the three uses the reference makes of it on 8-bit data:
   INTER_NEAREST on masks, INTER_LINEAR / INTER_CUBIC on uint8 RGB (datasets/occ_order_dataset.py:82-87, 105-106,
   123, 153-156, 169-170; inference.py:366-392, 455-481).  The algorithm lives in a third-party dependency that is
   NOT in /root/reference and NOT installed in this image: ``opencv-python`` (requirements.txt:4, unpinned).  What is
   restated is the published fixed-point algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp``
   (``resizeNN``; ``resizeGeneric_`` with ``HResizeLinear/VResizeLinear<uchar,int,short>`` and
   ``HResizeCubic/VResizeCubic`` + ``FixedPtCast<int,uchar,22>``, 11-bit coefficients, A = -0.75):
       scale = 1 / (dst / src)                                   (double)
       nearest:  sx = min(floor(dx * scale), src - 1)
       linear :  fx = float((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; sx < 0 -> (0, 0);
                 sx >= src-1 -> (src-1, 0); alpha = rint((1-fx, fx) * 2048) as short;
                 rows: H[x] = S[sx]*a0 + S[sx+1]*a1;  out = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2
       cubic  :  same fx / sx without the border reset; 4 taps sx-1..sx+2 clamped to the edge; float coefficients of
                 the Keys kernel (A = -0.75), alpha = rint(c * 2048); out = sat_u8((sum_k b_k * H_k + 2^21) >> 22)
   **Parity unpinned** for this layer: there is no cv2 here to run, and the reference holds no fixture for it.  The
   known closed forms it is checked against (tests/test_oracle_golden.py): identity at equal size, 2x2 box average for
   an exact 2x linear down-scale (OpenCV's INTER_AREA shortcut gives the same numbers), nearest index rule.
   A SECOND, independent source since round 5 (tests/test_datasets_cpu.py): ``torch.nn.functional.interpolate`` -- float
   arithmetic on the same half-pixel geometry (``align_corners=False``; bicubic with A = -0.75, nearest with OpenCV's
   floor rule) -- agrees with this restatement to within the 8-bit fixed-point rounding (<= 1 grey level, linear and
   cubic; identical nearest pixels; the float64 cubic to 1e-4) over 7 size pairs and the crops of the dataset fixtures'
   scenes.  That bounds a wrong geometry or kernel constant; it does not pin the fixed-point rounding itself, so the
   status stays "parity unpinned".

2. The item assembly of ``SupOcclusionOrderDataset`` / ``SupDepthOccOrderDataset`` (crop box arithmetic, padding, flip,
   normalisation, label layout, and the ORDER of the np.random draws): ``pair_plan`` / ``render_pair`` below follow
   datasets/occ_order_dataset.py:81-180 and utils/data_utils.py:61-124.  This layer IS pinned: tests/golden/
   dataset_items.npz is produced by the reference's own dataset classes (tests/golden/make_golden.py, with
   ``cv2.resize`` bound to ``resize`` of this file because cv2 does not exist here).
"""
import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2
INTER_CUBIC_F64 = 3      # not a cv2 constant: INTER_CUBIC applied to the float64 image x / 255. (render_pair)
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _scale(src, dst):
    return 1.0 / (float(dst) / float(src))          # resize.cpp: scale_x = 1. / inv_scale_x (doubles)


def _coords(src, dst):
    """float fx and int sx per destination index (the float/floor split of resizeGeneric_)."""
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * _scale(src, dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return (f - s.astype(np.float32)).astype(np.float32), s


def _short(c):
    return np.clip(np.rint(c.astype(np.float32) * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int64)


def linear_taps(src, dst):
    fx, sx = _coords(src, dst)
    lo, hi = sx < 0, sx >= src - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, src - 1, sx))
    idx = np.stack([sx, np.minimum(sx + 1, src - 1)], 1)
    coef = np.stack([_short(np.float32(1) - fx), _short(fx)], 1)
    return idx, coef


def cubic_taps(src, dst):
    fx, sx = _coords(src, dst)
    A = np.float32(-0.75)
    one = np.float32(1)
    x1 = fx + one
    c0 = ((A * x1 - np.float32(5) * A) * x1 + np.float32(8) * A) * x1 - np.float32(4) * A
    c1 = ((A + np.float32(2)) * fx - (A + np.float32(3))) * fx * fx + one
    xr = one - fx
    c2 = ((A + np.float32(2)) * xr - (A + np.float32(3))) * xr * xr + one
    c3 = one - c0 - c1 - c2
    coef = np.stack([_short(c) for c in (c0, c1, c2, c3)], 1)
    idx = np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, coef


def resize(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, (width, height), interpolation=...) for uint8 HxW / HxWxC (any dtype for nearest)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    if interpolation == INTER_NEAREST:
        xs = np.minimum(np.floor(np.arange(dw) * _scale(sw, dw)).astype(np.int64), sw - 1)
        ys = np.minimum(np.floor(np.arange(dh) * _scale(sh, dh)).astype(np.int64), sh - 1)
        return np.ascontiguousarray(img[ys][:, xs])
    assert img.dtype == np.uint8, "fixed-point path: 8-bit images"
    squeeze = img.ndim == 2
    src = img[:, :, None] if squeeze else img
    taps = linear_taps if interpolation == INTER_LINEAR else cubic_taps
    xi, xa = taps(sw, dw)
    yi, ya = taps(sh, dh)
    s = src.astype(np.int64)
    H = (s[:, xi, :] * xa[None, :, :, None]).sum(2)                     # [sh, dw, C] horizontal pass (int)
    R = H[yi]                                                           # [dh, taps, dw, C]
    if interpolation == INTER_LINEAR:
        b = ya[:, :, None, None]
        out = (((b[:, 0] * (R[:, 0] >> 4)) >> 16) + ((b[:, 1] * (R[:, 1] >> 4)) >> 16) + 2) >> 2
    else:
        out = ((R * ya[:, :, None, None]).sum(1) + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_cubic_f64(img, dsize):
    """cv2.resize(float64 HxWxC image, (width, height), interpolation=cv2.INTER_CUBIC): the non-fixed-point path of
    resizeGeneric_ (HResizeCubic<double,double,float> / VResizeCubic<double,double,float>): float coefficients, double
    products accumulated left to right, no rounding step.  Same pinning status as ``resize`` (unpinned)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    assert img.dtype == np.float64 and img.ndim == 3

    def taps(src, dst):
        fx, sx = _coords(src, dst)
        A, one = np.float32(-0.75), np.float32(1)
        x1 = fx + one
        c0 = ((A * x1 - np.float32(5) * A) * x1 + np.float32(8) * A) * x1 - np.float32(4) * A
        c1 = ((A + np.float32(2)) * fx - (A + np.float32(3))) * fx * fx + one
        xr = one - fx
        c2 = ((A + np.float32(2)) * xr - (A + np.float32(3))) * xr * xr + one
        c3 = one - c0 - c1 - c2
        coef = np.stack([c0, c1, c2, c3], 1).astype(np.float64)
        return np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, src - 1), coef

    xi, xa = taps(sw, dw)
    yi, ya = taps(sh, dh)
    H = img[:, xi[:, 0], :] * xa[None, :, 0, None]
    for k in range(1, 4):
        H = H + img[:, xi[:, k], :] * xa[None, :, k, None]
    out = H[yi[:, 0]] * ya[:, 0, None, None]
    for k in range(1, 4):
        out = out + H[yi[:, k]] * ya[:, k, None, None]
    return out


def constrain_to_multiple_of(x, multiple, max_val=None):
    """midas/transforms.py:96-105 for resize_method 'upper_bound' (min_val = 0)."""
    y = int(np.round(x / multiple) * multiple)
    if max_val is not None and y > max_val:
        y = int(np.floor(x / multiple) * multiple)
    return y


def transform_resize(image, width, height):
    """utils/data_utils.py:37-53: Resize(width, height, keep_aspect_ratio=False, ensure_multiple_of=32,
    resize_method='upper_bound', INTER_CUBIC) on image / 255., NormalizeImage(ImageNet mean / std), PrepareForNet ->
    float32 [3, H', W']."""
    h, w = image.shape[:2]
    new_h = constrain_to_multiple_of((height / h) * h, 32, max_val=height)
    new_w = constrain_to_multiple_of((width / w) * w, 32, max_val=width)
    x = resize_cubic_f64(image / 255., (new_w, new_h))
    x = (x - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32)


# ---- utils/data_utils.py -------------------------------------------------------------------------------------------
def combine_bbox(bboxes):
    """utils/data_utils.py:61-73 (xywh rows -> enclosing xywh)."""
    l = bboxes[:, 0].min()
    u = bboxes[:, 1].min()
    r = (bboxes[:, 0] + bboxes[:, 2]).max()
    b = (bboxes[:, 1] + bboxes[:, 3]).max()
    return np.array([l, u, r - l, b - u])


def crop_padding(img, roi, pad_value=0):
    """utils/data_utils.py:105-124: the roi of img, zero (pad_value) outside the image."""
    x, y, w, h = [int(v) for v in roi]
    H, W = img.shape[:2]
    out = np.full((h, w) + img.shape[2:], pad_value, dtype=img.dtype)
    x0, x1, y0, y1 = max(x, 0), min(x + w, W), max(y, 0), min(y + h, H)
    if x1 > x0 and y1 > y0:
        out[y0 - y:y1 - y, x0 - x:x1 - x] = img[y0:y1, x0:x1]
    return out


# ---- datasets/occ_order_dataset.py:81-180 ---------------------------------------------------------------------------
def pair_plan(mode, modal_shape, bboxes, idx1, idx2, phase, base_aug, rng, randshift=True):
    """The crop rectangle (x, y, w, h) in image coordinates, the RGB interpolation and the flip flag of one item, with
    the reference's np.random draws in the reference's order.  mode: 'patch' (_get_pair, :138-180), 'image'
    (_get_pair_image, :98-130) or 'resize' (_get_pair_resize, :81-96)."""
    _, hh, ww = modal_shape
    if mode == "patch":
        bbox = combine_bbox(np.asarray(bboxes)[(idx1, idx2), :])
        centerx = bbox[0] + bbox[2] / 2.
        centery = bbox[1] + bbox[3] / 2.
        size = max([np.sqrt(bbox[2] * bbox[3] * 2.), bbox[2] * 1.1, bbox[3] * 1.1])
        if phase == "train":
            if randshift:
                centerx += rng.uniform(*base_aug["shift"]) * size
                centery += rng.uniform(*base_aug["shift"]) * size
            size /= rng.uniform(*base_aug["scale"])
        box = (int(centerx - size / 2.), int(centery - size / 2.), int(size), int(size))
        interp = INTER_CUBIC
    elif mode == "image":
        hw = int(max(hh, ww))
        box = (-((hw - ww) // 2), -((hw - hh) // 2), hw, hw)
        interp = INTER_LINEAR
    elif mode == "resize":
        box = (0, 0, ww, hh)
        interp = INTER_LINEAR
    else:
        raise ValueError(mode)
    flip = bool(base_aug["flip"] and rng.rand() > 0.5)
    return box, interp, flip


def render_pair(image, mask1, mask2, box, interp, flip, sz, mean, std):
    """(rgb[3,sz,sz] fp32 normalised, modal1[sz,sz], modal2[sz,sz] in the mask dtype) of one planned item."""
    m1 = resize(crop_padding(mask1, box), (sz, sz), INTER_NEAREST)
    m2 = resize(crop_padding(mask2, box), (sz, sz), INTER_NEAREST)
    if interp == INTER_CUBIC_F64:
        x = resize_cubic_f64(crop_padding(image, box) / 255., (sz, sz))
        x = (x - list(mean)) / list(std)
        if flip:
            m1, m2, x = m1[:, ::-1], m2[:, ::-1], x[:, ::-1, :]
        return (np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32), np.ascontiguousarray(m1),
                np.ascontiguousarray(m2))
    rgb = resize(crop_padding(image, box), (sz, sz), interp)
    if flip:
        m1, m2, rgb = m1[:, ::-1], m2[:, ::-1], rgb[:, ::-1, :]
    x = rgb.astype(np.float32).transpose(2, 0, 1) / np.float32(255.)
    x = (x - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    return x.astype(np.float32), np.ascontiguousarray(m1), np.ascontiguousarray(m2)
