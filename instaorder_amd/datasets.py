"""Device-side counterpart of the reference's training datasets + DataLoader collate
(datasets/occ_order_dataset.py, datasets/depth_occ_order_dataset.py): the per-item DECISIONS (which pair, the random
shift / scale / flip / direction swap, the labels) stay on the host and follow the reference line by line, including
the order of its ``np.random`` draws; the per-item PIXEL WORK (``crop_padding`` + three ``cv2.resize`` + flip +
normalise) runs in one HIP launch per batch (``io_pair_planes_u8``, csrc/preprocess.hip) on uint8 sources that are
uploaded once through pinned memory.  ``batch(indices)`` returns the tuple the reference's DataLoader yields, already
on the GPU, so it plugs straight into ``model.set_input(*batch)`` (trainer.py:167, 235).

Annotation parsing (datasets/reader.py: COCO / InstaOrder json via pycocotools) is out of scope -- a ``data_reader``
object with the reader's methods is passed in: ``get_image_length()``, ``get_image_instances(idx, with_gt=True)`` ->
(modal[n,H,W] uint8, category[n], bboxes[n,4] xywh, amodal, image_fn), ``get_gt_ordering(idx, type=..., ...)``, and for
the depth datasets ``get_geometric_length()`` / ``get_imgId_and_depth(i)``; plus ``load_image(image_fn)`` -> uint8 HxWx3.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

INTER_LINEAR, INTER_CUBIC, INTER_CUBIC_F64 = 1, 2, 3     # io_pair_desc.interp


# ---- utils/data_utils.py:61-73 --------------------------------------------------------------------------------------
def combine_bbox(bboxes):
    l = bboxes[:, 0].min()
    u = bboxes[:, 1].min()
    r = (bboxes[:, 0] + bboxes[:, 2]).max()
    b = (bboxes[:, 1] + bboxes[:, 3]).max()
    return np.array([l, u, r - l, b - u])


def patch_box(bboxes, i, j):
    """The square crop around a pair before augmentation: centre and side (occ_order_dataset.py:139-142,
    inference.py:450-453)."""
    bbox = combine_bbox(np.asarray(bboxes)[(i, j), :])
    centerx = bbox[0] + bbox[2] / 2.
    centery = bbox[1] + bbox[3] / 2.
    size = max([np.sqrt(bbox[2] * bbox[3] * 2.), bbox[2] * 1.1, bbox[3] * 1.1])
    return centerx, centery, size


def crop_plan(mode, modal_shape, bboxes, idx1, idx2, phase, base_aug, rng, randshift=True):
    """Crop rectangle (x, y, w, h), image interpolation and flip flag of one item -- ``_get_pair`` (:138-180),
    ``_get_pair_image`` (:98-130), ``_get_pair_resize`` (:81-96) without their pixel work; draws from ``rng`` in the
    reference's order (shift x, shift y, scale, flip)."""
    _, hh, ww = modal_shape
    if mode == "patch":
        centerx, centery, size = patch_box(bboxes, idx1, idx2)
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
        box = (0, 0, int(ww), int(hh))
        interp = INTER_LINEAR
    else:
        raise Exception("No such patch_or_image: {}".format(mode))
    flip = bool(base_aug["flip"] and rng.rand() > 0.5)
    return box, interp, flip


# ---- the device renderer --------------------------------------------------------------------------------------------
class PairRenderer(object):
    """uint8 images + instance masks -> (rgb[P,3,S,S], modal1[P,1,S,S], modal2[P,1,S,S]) fp32 on the GPU.
    ``input_size`` = S, or (height, width) for outputs that are not square (the 'orig' inference mode)."""

    def __init__(self, input_size, mean, std, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("instaorder_amd.datasets.PairRenderer needs a GPU (there is no CPU path)")
        if isinstance(input_size, (tuple, list)):
            self.SH, self.S = int(input_size[0]), int(input_size[1])
        else:
            self.SH = self.S = int(input_size)
        self.device = torch.device(device)
        self.mean = (C.c_double * 3)(*[float(v) for v in mean])
        self.std = (C.c_double * 3)(*[float(v) for v in std])
        self._pinned = [None, None]
        self._events = [None, None]
        self._turn = 0

    def _staging(self, nbytes):
        """two pinned staging buffers used alternately; a buffer is reused only after its upload has completed"""
        k = self._turn
        self._turn ^= 1
        if self._events[k] is not None:
            self._events[k].synchronize()
        if self._pinned[k] is None or self._pinned[k].numel() < nbytes:
            self._pinned[k] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return k, self._pinned[k]

    def render(self, images, masks, items, load_rgb=True):
        """images: list of uint8 [H,W,3] arrays (entries may be None when load_rgb is False); masks: list of uint8
        [n,H,W] arrays, one per image; items: list of (image_index, idx1, idx2, (x, y, w, h), interp, flip)."""
        P = len(items)
        if P == 0:
            raise ValueError("PairRenderer.render: empty batch")
        S, SH = self.S, self.SH
        # arena layout: every referenced image once, every referenced mask once (16-byte aligned)
        off, cursor = {}, 0

        def place(key, nbytes):
            nonlocal cursor
            if key not in off:
                off[key] = cursor
                cursor = (cursor + nbytes + 15) // 16 * 16
            return off[key]

        desc = (_lib.PairDesc * P)()
        for k, (ii, i1, i2, box, interp, flip) in enumerate(items):
            n, H, W = masks[ii].shape
            if masks[ii].dtype != np.uint8:
                raise TypeError("instance masks must be uint8 (got %s)" % masks[ii].dtype)
            d = desc[k]
            d.image_off = place(("img", ii), H * W * 3) if load_rgb else 0
            d.mask1_off = place(("m", ii, int(i1)), H * W)
            d.mask2_off = place(("m", ii, int(i2)), H * W)
            d.H, d.W = H, W
            d.x, d.y, d.w, d.h = [int(v) for v in box]
            d.flip = int(bool(flip))
            d.interp = int(interp)
            if load_rgb and (images[ii].shape != (H, W, 3) or images[ii].dtype != np.uint8):
                raise ValueError("image %d: expected uint8 [%d,%d,3], got %s %s" % (ii, H, W, images[ii].dtype,
                                                                                   images[ii].shape))
        nbytes = max(cursor, 16)
        dbytes = C.sizeof(desc)
        slot, stage = self._staging(nbytes + dbytes)
        host = stage.numpy()
        for key, o in off.items():
            src = images[key[1]] if key[0] == "img" else masks[key[1]][key[2]]
            host[o:o + src.size] = np.ascontiguousarray(src).reshape(-1)
        host[nbytes:nbytes + dbytes] = np.frombuffer(desc, dtype=np.uint8)
        dev = stage[:nbytes + dbytes].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._events[slot] = ev
        rgb = torch.empty((P, 3, SH, S), device=self.device) if load_rgb else None
        m1 = torch.empty((P, 1, SH, S), device=self.device)
        m2 = torch.empty((P, 1, SH, S), device=self.device)
        rc = _lib.lib().io_pair_planes_u8_hw(
            C.c_void_p(dev.data_ptr()), C.c_size_t(nbytes), C.c_void_p(dev.data_ptr() + nbytes),
            C.cast(desc, C.c_void_p), P, SH, S, C.cast(self.mean, C.c_void_p), C.cast(self.std, C.c_void_p),
            C.c_void_p(rgb.data_ptr()) if load_rgb else None, C.c_void_p(m1.data_ptr()), C.c_void_p(m2.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(rc, "io_pair_planes_u8_hw")
        if rgb is None:
            rgb = torch.zeros((P, 3, SH, S), device=self.device)     # occ_order_dataset.py:231-232
        return rgb, m1, m2


# ---- item logic of the datasets ----------------------------------------------------------------------------------------
class _Batches(object):
    def __init__(self, config, phase, algo, data_reader, load_image, device="cuda:0", rng=None):
        self.config = config
        self.phase = phase
        self.algo = algo
        self.data_reader = data_reader
        self.load_image = load_image
        self.sz = config["input_size"]
        self.mode = config["patch_or_image"]
        if self.mode not in ("patch", "image", "resize"):
            raise Exception("No such patch_or_image: {}".format(self.mode))
        self.rng = np.random if rng is None else rng          # the reference draws from the global numpy generator
        self.device = device
        self._renderer = None

    @property
    def renderer(self):
        if self._renderer is None:                             # created on first use: planning alone needs no GPU
            self._renderer = PairRenderer(self.sz, self.config["data_mean"], self.config["data_std"], self.device)
        return self._renderer

    def _instances(self, idx):
        modal, category, bboxes, amodal, image_fn = self.data_reader.get_image_instances(idx, with_gt=True)
        if self.config.get("use_category", False):
            modal = modal * category[:, None, None]               # occ_order_dataset.py:184-185
            if modal.max() > 255:
                raise ValueError("use_category with category ids above 255 does not fit the uint8 mask arena")
        return np.ascontiguousarray(modal.astype(np.uint8)), bboxes, image_fn

    def _crop(self, modal, bboxes, idx1, idx2):
        return crop_plan(self.mode, modal.shape, bboxes, idx1, idx2, self.phase, self.config["base_aug"], self.rng,
                         randshift=True)

    def _render(self, plans):
        """plans: list of dicts with modal, image_fn, idx1, idx2, box, interp, flip (idx1/idx2 already in output order)"""
        load_rgb = bool(self.config["load_rgb"])
        images, masks, items, seen = [], [], [], {}
        for p in plans:
            key = id(p["modal"])
            if key not in seen:
                seen[key] = len(masks)
                masks.append(p["modal"])
                images.append(np.asarray(self.load_image(p["image_fn"])) if load_rgb else None)
            items.append((seen[key], p["idx1"], p["idx2"], p["box"], p["interp"], p["flip"]))
        return self.renderer.render(images, masks, items, load_rgb=load_rgb)


class SupOcclusionOrderBatches(_Batches):
    """``SupOcclusionOrderDataset`` (occ_order_dataset.py:21-279) for algo 'InstaOrderNet_o' and 'OrderNet'."""

    def __len__(self):
        return self.data_reader.get_image_length()

    def _get_pair_ind(self, idx):
        """occ_order_dataset.py:182-200 (dataset 'InstaOrder' / COCOA branches)."""
        modal, bboxes, image_fn = self._instances(idx)
        if self.config["dataset"] == "KINS":
            from . import inference as infer
            amodal = self.data_reader.get_image_instances(idx, with_gt=True)[3]
            gt = infer.infer_gt_order(modal, amodal)
        elif self.config["dataset"] == "InstaOrder":
            gt = self.data_reader.get_gt_ordering(idx, type="occlusion", rm_bidirec=self.config["remove_occ_bidirec"])
        else:
            gt = self.data_reader.get_gt_ordering(idx)
        np.fill_diagonal(gt, -1)
        pairs = np.where(gt == 1)
        non_pairs = np.where(gt == 0)
        if len(pairs[0]) == 0:
            return self._get_pair_ind(self.rng.choice(len(self)))
        return modal, bboxes, image_fn, pairs, non_pairs, gt

    def plan(self, idx):
        """One item of ``__getitem__`` (:202-279) up to, but not including, the pixel work."""
        rng = self.rng
        modal, bboxes, image_fn, pairs, non_pairs, gt = self._get_pair_ind(idx)
        label = None
        if rng.rand() < 0.7 or len(non_pairs[0]) == 0:
            r = rng.choice(len(pairs[0]))
            idx1, idx2 = pairs[0][r], pairs[1][r]
            if self.algo == "OrderNet":
                label = 1
                if self.config.get("extend_bidirec", False) and gt[idx2, idx1]:
                    label = 3
        else:
            r = rng.choice(len(non_pairs[0]))
            idx1, idx2 = non_pairs[0][r], non_pairs[1][r]
            label = 2
        box, interp, flip = self._crop(modal, bboxes, idx1, idx2)
        a_over_b, b_over_a = gt[idx1, idx2], gt[idx2, idx1]
        keep = rng.rand() < 0.5
        if self.algo == "OrderNet":
            if not keep:
                label = 0 if label == 1 else label
            target = label
        elif self.algo == "InstaOrderNet_o":
            target = [b_over_a, a_over_b] if keep else [a_over_b, b_over_a]
        else:
            raise Exception("SupOcclusionOrderDataset serves OrderNet / InstaOrderNet_o, not {}".format(self.algo))
        if not keep:
            idx1, idx2 = idx2, idx1
        return dict(modal=modal, image_fn=image_fn, idx1=int(idx1), idx2=int(idx2), box=box, interp=interp,
                    flip=flip, target=target)

    def batch(self, indices):
        """(rgb, modal1, modal2, occ_order) as the DataLoader over the reference dataset yields them, on the GPU."""
        plans = [self.plan(i) for i in indices]
        rgb, m1, m2 = self._render(plans)
        dev = rgb.device
        if self.algo == "OrderNet":
            target = torch.tensor([p["target"] for p in plans], dtype=torch.long).to(dev)
        else:
            target = torch.tensor(np.asarray([p["target"] for p in plans], dtype=np.float32)).to(dev)
        return rgb, m1, m2, target


class SupDepthOccOrderBatches(_Batches):
    """``SupDepthOccOrderDataset`` (depth_occ_order_dataset.py:20-240) for algo 'InstaOrderNet_od' /
    'InstaDepthNet_od': one item per annotated depth relation "i<j" / "i=j"."""
    WITH_OCC = True

    def __len__(self):
        return self.data_reader.get_geometric_length()

    def _get_pair_ind(self, img_id):
        """depth_occ_order_dataset.py:150-160 (no category scaling, no re-draw in this class)."""
        modal, category, bboxes, amodal, image_fn = self.data_reader.get_image_instances(img_id, with_gt=True)
        modal = np.ascontiguousarray(modal.astype(np.uint8))
        gt_depth = self.data_reader.get_gt_ordering(img_id, type="depth", rm_overlap=self.config["remove_depth_overlap"])
        gt_occ = self.data_reader.get_gt_ordering(img_id, type="occlusion", rm_bidirec=self.config["remove_occ_bidirec"])
        return modal, bboxes, image_fn, gt_depth, gt_occ

    def plan(self, idx):
        rng = self.rng
        img_id, depth_order = self.data_reader.get_imgId_and_depth(idx)
        modal, bboxes, image_fn, (gt_depth, gt_overlap, gt_count), gt_occ = self._get_pair_ind(img_id)
        split_char = "<" if "<" in depth_order else "="
        idx1, idx2 = list(map(int, depth_order.split(split_char)))
        box, interp, flip = self._crop(modal, bboxes, idx1, idx2)
        if gt_depth[idx1, idx2] == -1:
            depth_label = -1
        elif gt_depth[idx1, idx2] == 1 and gt_depth[idx2, idx1] == 0:
            depth_label = 0
        elif gt_depth[idx1, idx2] == 2:
            depth_label = 2
        else:
            raise Exception("inconsistent depth annotation for {} in image {}".format(depth_order, img_id))
        depth_count = gt_count[idx1, idx2]                      # indexed before the direction swap (:222-223)
        is_overlap = gt_overlap[idx1, idx2]
        occ = None
        if self.WITH_OCC:
            a_over_b, b_over_a = gt_occ[idx1, idx2], gt_occ[idx2, idx1]
        if rng.rand() < 0.5:
            if self.WITH_OCC:
                occ = [b_over_a, a_over_b]
        else:
            depth_label = 1 if depth_label == 0 else depth_label
            if self.WITH_OCC:
                occ = [a_over_b, b_over_a]
            idx1, idx2 = idx2, idx1
        return dict(modal=modal, image_fn=image_fn, idx1=int(idx1), idx2=int(idx2), box=box, interp=interp,
                    flip=flip, depth=int(depth_label), count=depth_count, is_overlap=is_overlap, occ=occ)

    def batch(self, indices):
        """(rgb, modal1, modal2, depth_order, count, is_overlap[, occ_order]) on the GPU (depth_occ_order_dataset.py:
        234-240, depth_order_dataset.py:238-244 + default collate)."""
        plans = [self.plan(i) for i in indices]
        rgb, m1, m2 = self._render(plans)
        dev = rgb.device
        depth = torch.tensor([p["depth"] for p in plans], dtype=torch.long).to(dev)
        count = torch.tensor(np.asarray([p["count"] for p in plans])).to(dev)
        ovl = torch.tensor(np.asarray([p["is_overlap"] for p in plans])).to(dev)
        if not self.WITH_OCC:
            return rgb, m1, m2, depth, count, ovl
        occ = torch.tensor(np.asarray([p["occ"] for p in plans], dtype=np.float32)).to(dev)
        return rgb, m1, m2, depth, count, ovl, occ


class SupDepthOrderBatches(SupDepthOccOrderBatches):
    """``SupDepthOrderDataset`` (depth_order_dataset.py:22-244) for algo 'InstaOrderNet_d' / 'InstaDepthNet_d': the
    depth relation items without the occlusion label; category scaling and the re-draw of images that carry no depth
    annotation as in :180-197 (the reference passes the re-drawn RELATION index on as an image id -- kept)."""
    WITH_OCC = False

    def _get_pair_ind(self, img_id):
        modal, bboxes, image_fn = self._instances(img_id)
        gt = self.data_reader.get_gt_ordering(img_id, type="depth", rm_overlap=self.config["remove_depth_overlap"])
        if gt[0].sum() == gt[0].shape[0] * gt[0].shape[1] * (-1):
            return self._get_pair_ind(self.rng.choice(len(self)))
        return modal, bboxes, image_fn, gt, None


class BatchPrefetcher(object):
    """Builds the next batches on a worker thread and a side HIP stream while the GPU runs the current step -- the role
    of the reference's DataLoader workers (trainer.py:95-101: ``workers`` processes + default collate), with the
    uint8 upload and the render kernel off the compute stream.  ``index_batches``: iterable of index lists."""

    def __init__(self, batches, index_batches, depth=2):
        import queue
        import threading
        self.batches = batches
        self._q = queue.Queue(maxsize=depth)
        self._stream = torch.cuda.Stream(device=batches.device)
        self._err = None
        self._stop = False

        def work():
            try:
                with torch.cuda.stream(self._stream):
                    for idx in index_batches:
                        if self._stop:
                            break
                        out = self.batches.batch(idx)
                        ev = torch.cuda.Event()
                        ev.record(self._stream)
                        self._q.put((out, ev))
            except BaseException as e:          # surfaced on the consumer side
                self._err = e
            self._q.put(None)

        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        out, ev = item
        cur = torch.cuda.current_stream(self.batches.device)
        cur.wait_event(ev)
        for t in out:
            t.record_stream(cur)
        return out

    def close(self):
        self._stop = True
        while self._thread.is_alive():
            try:
                self._q.get(timeout=0.1)
            except Exception:
                pass
