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
