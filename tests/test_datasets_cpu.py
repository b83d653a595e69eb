"""Input pipeline, host side (no GPU): the oracle's resize against closed forms, and the item logic of
instaorder_amd.datasets (pair choice, crop arithmetic, flip, direction swap, labels, np.random draw order) against
items produced by the reference's own dataset classes (tests/golden/dataset_items.npz, made by
tests/golden/make_golden.py with cv2.resize bound to the oracle's restatement -- cv2 is absent in this image)."""
import json
import os

import numpy as np
import pytest

from instaorder_amd import datasets, synthetic
from oracle import preprocess_oracle as po

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataset_items.npz")


def test_resize_closed_forms():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (40, 52, 3)).astype(np.uint8)
    for it in (po.INTER_NEAREST, po.INTER_LINEAR, po.INTER_CUBIC):
        assert (po.resize(img, (52, 40), it) == img).all()                      # identity at equal size
    box = (img.astype(int).reshape(20, 2, 26, 2, 3).sum((1, 3)) + 2) >> 2       # exact 2x: the INTER_AREA shortcut
    assert (po.resize(img, (26, 20), po.INTER_LINEAR) == box).all()
    row = np.arange(4, dtype=np.uint8)[None, :]
    assert po.resize(row, (2, 1), po.INTER_NEAREST).tolist() == [[0, 2]]        # sx = floor(dx * scale)
    assert po.resize(row[:, :2], (4, 1), po.INTER_NEAREST).tolist() == [[0, 0, 1, 1]]
    flat = np.full((9, 7, 3), 201, np.uint8)                                    # coefficients sum to 2048
    for it in (po.INTER_LINEAR, po.INTER_CUBIC):
        assert (po.resize(flat, (23, 31), it) == 201).all()
    up = po.resize(img, (100, 77), po.INTER_CUBIC).astype(int)                  # overshoot saturates, stays 8-bit
    assert up.min() >= 0 and up.max() <= 255 and abs(up.mean() - img.mean()) < 1.0


def test_crop_padding_matches_definition():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    for roi in [(-5, -7, 40, 40), (3, 4, 10, 8), (25, 15, 10, 10), (100, 100, 5, 5), (-50, 0, 10, 10)]:
        out = po.crop_padding(img, roi)
        x, y, w, h = roi
        for yy in range(h):
            for xx in range(w):
                iy, ix = y + yy, x + xx
                want = img[iy, ix] if (0 <= iy < 20 and 0 <= ix < 30) else 0
                assert (out[yy, xx] == want).all()


def _variants():
    z = np.load(GOLD)
    cfg = json.loads(str(z["config_json"]))
    rows = [str(v).split("|") for v in z["variants"]]
    return z, cfg, rows


@pytest.mark.parametrize("k", range(6))
def test_item_logic_matches_reference_datasets(k):
    z, cfg, rows = _variants()
    name, kind, algo, mode, phase, seed = rows[k]
    cfg = dict(cfg, patch_or_image=mode)
    rd = synthetic.SyntheticReader(int(z["reader_seed"]))
    cls = {"occ": datasets.SupOcclusionOrderBatches, "depth_occ": datasets.SupDepthOccOrderBatches,
           "depth": datasets.SupDepthOrderBatches}[kind]
    ds = cls(cfg, phase, algo, rd, rd.load_image)
    np.random.seed(int(seed))
    n = z[name + "_f0"].shape[0]
    plans = [ds.plan(i) for i in range(n)]
    S = cfg["input_size"]
    for i, p in enumerate(plans):
        rgb, m1, m2 = po.render_pair(rd.load_image(p["image_fn"]), p["modal"][p["idx1"]], p["modal"][p["idx2"]],
                                     p["box"], p["interp"], p["flip"], S, cfg["data_mean"], cfg["data_std"])
        assert np.array_equal(rgb, z[name + "_f0"][i]), (name, i)
        assert np.array_equal(m1.astype(np.float32), z[name + "_f1"][i, 0]), (name, i)
        assert np.array_equal(m2.astype(np.float32), z[name + "_f2"][i, 0]), (name, i)
        if kind == "occ":
            assert np.array_equal(np.asarray(p["target"], np.float64), z[name + "_f3"][i].astype(np.float64))
        else:
            assert p["depth"] == int(z[name + "_f3"][i])
            assert int(p["count"]) == int(z[name + "_f4"][i]) and int(p["is_overlap"]) == int(z[name + "_f5"][i])
            if kind == "depth_occ":
                assert np.array_equal(np.asarray(p["occ"], np.float64), z[name + "_f6"][i].astype(np.float64))


def test_transform_resize_matches_reference_chain():
    """the 'resize' inference transform: the reference's own utils.data_utils.transform_resize (MiDaS Resize ->
    NormalizeImage -> PrepareForNet; cv2.resize bound to the oracle's float64 cubic) == oracle.transform_resize"""
    z = np.load(GOLD)
    rd = synthetic.SyntheticReader(int(z["reader_seed"]))
    for k, (w, h) in enumerate([(64, 64), (96, 64)]):
        got = po.transform_resize(rd.scenes[k]["image"], w, h)
        assert got.dtype == np.float32 and np.array_equal(got, z["transform_resize_%d" % k])
    flat = np.full((30, 50, 3), 77, np.uint8)              # constant image: cubic weights sum to one (within 1 ulp of fp32)
    out = po.resize_cubic_f64(flat / 255., (64, 32))
    assert np.abs(out - 77 / 255.).max() < 1e-6


# ---- a second, independent source for the resize layer (the row stays "parity unpinned": cv2 itself is absent) ------------
def _torch_resize(img, dsize, mode, dtype):
    """torch.nn.functional.interpolate of an HxWxC image to (width, height) = dsize -- PyTorch's own implementation of the
    same published sampling geometry (half-pixel centres: src = (dst + 0.5) * scale - 0.5, border taps clamped, Keys
    A = -0.75 for bicubic; 'nearest' = floor(dst * scale))."""
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(np.ascontiguousarray(img)).to(dtype).permute(2, 0, 1)[None]
    kw = {} if mode == "nearest" else dict(align_corners=False, antialias=False)
    y = F.interpolate(x, size=(int(dsize[1]), int(dsize[0])), mode=mode, **kw)
    return y[0].permute(1, 2, 0).numpy()


SIZES = [((40, 52), (64, 64)), ((97, 61), (64, 64)), ((256, 256), (64, 64)), ((33, 47), (256, 256)),
         ((300, 180), (256, 256)), ((480, 640), (384, 384)), ((64, 64), (96, 128))]


@pytest.mark.parametrize("src,dst", SIZES)
def test_resize_against_torch_interpolate(src, dst):
    """utils/data_utils.py:104-124 / datasets/occ_order_dataset.py:138-180 resize crops with cv2.resize; cv2 is not in this
    image, so the oracle's restatement of OpenCV's arithmetic is bounded here by a second implementation of the same
    geometry that IS available: the 8-bit fixed-point paths stay within +-1 LSB of the float result (11-bit coefficients,
    two roundings), the float64 cubic agrees to float32 coefficient precision, nearest picks the same source pixels."""
    import torch
    rng = np.random.RandomState(src[0] * 131 + dst[0])
    img = rng.randint(0, 256, (src[0], src[1], 3)).astype(np.uint8)
    # a smooth component too, so that not every pixel is an extremum
    yy, xx = np.mgrid[0:src[0], 0:src[1]]
    img[:, :, 1] = (127 + 120 * np.sin(yy / 7.0) * np.cos(xx / 5.0)).astype(np.uint8)
    dsize = (dst[1], dst[0])
    near = po.resize(img, dsize, po.INTER_NEAREST)
    assert np.array_equal(near, _torch_resize(img, dsize, "nearest", torch.float32).astype(np.uint8))
    lin = po.resize(img, dsize, po.INTER_LINEAR).astype(np.float64)
    ref = _torch_resize(img, dsize, "bilinear", torch.float64)
    assert np.abs(lin - ref).max() <= 1.0 + 1e-9, np.abs(lin - ref).max()
    assert np.abs(lin - ref).mean() < 0.35           # (rounding to 8 bits alone: 0.25)
    cub = po.resize(img, dsize, po.INTER_CUBIC).astype(np.float64)
    refc = np.clip(_torch_resize(img, dsize, "bicubic", torch.float64), 0, 255)
    assert np.abs(cub - refc).max() <= 1.0 + 1e-9, np.abs(cub - refc).max()
    assert np.abs(cub - refc).mean() < 0.35
    f64 = po.resize_cubic_f64(img.astype(np.float64) / 255.0, dsize)
    reff = _torch_resize(img.astype(np.float64) / 255.0, dsize, "bicubic", torch.float64)
    # OpenCV evaluates the source coordinate and the Keys coefficients in float32 (restated so), PyTorch in double: a
    # coordinate near 640 carries 3e-5 of rounding, hence the bound (0.03 of an 8-bit step on [0, 1] data)
    assert np.abs(f64 - reff).max() < 1e-4, np.abs(f64 - reff).max()


def test_dataset_item_crops_within_one_lsb_of_float_resize():
    """The same bound on what the pipeline really resizes: the crops of the dataset-item fixtures' scenes (patch mode,
    random boxes with padding) at the fixtures' input size."""
    import torch
    rd = synthetic.SyntheticReader(77, n_images=3, n_inst=4, max_side=160, min_side=96, empty_every=0)
    rng = np.random.RandomState(3)
    worst = 0.0
    for k in range(3):
        image = np.asarray(rd.load_image("scene%d" % k))
        H, W = image.shape[:2]
        for _ in range(6):
            w, h = int(rng.randint(20, W)), int(rng.randint(20, H))
            x, y = int(rng.randint(-w // 3, W - w // 2)), int(rng.randint(-h // 3, H - h // 2))
            crop = po.crop_padding(image, (x, y, w, h))
            for it, mode in ((po.INTER_LINEAR, "bilinear"), (po.INTER_CUBIC, "bicubic")):
                got = po.resize(crop, (64, 64), it).astype(np.float64)
                ref = np.clip(_torch_resize(crop, (64, 64), mode, torch.float64), 0, 255)
                worst = max(worst, float(np.abs(got - ref).max()))
    assert worst <= 1.0 + 1e-9, worst
