#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference; never on the GPU box).
The reference is imported unmodified; the shims below exist because this
container has no GPU and lacks a few third-party packages the reference imports
at module top but never calls on the path exercised here.

Inputs and weights are regenerated from seeds by ``instaorder_amd.synthetic``;
only the reference's OUTPUTS are stored (losses, logits, gradient norms +
sampled elements, post-step weights norms + samples, BN running statistics,
order matrices, metrics, scheduler values).

usage:  python tests/golden/make_golden.py [case ...]
"""
import os
import sys
import types
from unittest import mock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from instaorder_amd import synthetic  # noqa: E402

NSAMP = 64


def sample_idx(n):
    return (np.arange(NSAMP, dtype=np.int64) * 2654435761) % max(n, 1)


def install_shims():
    np.int = int  # inference.py:352,441,518 use the removed alias
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class Normalize:
        def __init__(self, mean, std):
            self.m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
            self.s = torch.tensor(std, dtype=torch.float32)[:, None, None]

        def __call__(self, x):
            return (x - self.m) / self.s

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tvt.Normalize, tvt.Compose = Normalize, Compose
    tv.transforms, tv.models, tv.utils = tvt, mock.MagicMock(), mock.MagicMock()
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.models": tv.models, "torchvision.utils": tv.utils})
    for n in ["skimage", "skimage.morphology", "skimage.io", "skimage.draw", "pycocotools",
              "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval", "cvbase"]:
        sys.modules[n] = mock.MagicMock(name=n)
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST = cv2.INTER_LINEAR = cv2.INTER_CUBIC = cv2.INTER_AREA = 0

    def resize(img, size, interpolation=None):
        assert tuple(img.shape[:2]) == tuple(size[::-1]), "identity resize only"
        return img

    cv2.resize = resize
    sys.modules["cv2"] = cv2
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    os.environ.setdefault("HOME", "/root")
    sys.path.insert(0, REF)


def init_dist(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def load_cfg(algo):
    import yaml
    with open(os.path.join(REF, "experiments/InstaOrder", algo, "config.yaml")) as f:
        return yaml.safe_load(f)


def model_cfg(algo):
    """the 'model' section of experiments/InstaOrder/<algo>/config.yaml"""
    return load_cfg(algo)["model"]


def build(algo, seed, dist_model=True, style="xavier"):
    import models
    cfg = model_cfg(algo)
    m = getattr(models, algo)(cfg, dist_model=dist_model)
    nc = cfg["backbone_param"]["num_classes"]
    sd = synthetic.make_state_dict(seed, 5, nc, prefix="module.", style=style)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m, cfg


def set_input(m, algo, batch):
    t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
    if algo == "InstaOrderNet_od":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"],
                    t["is_overlap"], t["occ_order"])
    elif algo == "InstaOrderNet_d":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"])
    elif algo == "OrderNet":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"])  # class ids 0..2
    else:
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["occ_order"])


def snapshot(m, what):
    """norms + sampled elements of params ('p') or grads ('g'); BN buffers for 'bn'."""
    names, norms, samp = [], [], []
    if what in ("p", "g"):
        for k, p in m.model.named_parameters():
            t = p.detach() if what == "p" else p.grad.detach()
            a = t.contiguous().view(-1).double().numpy()
            names.append(k)
            norms.append(np.sqrt((a * a).sum()))
            samp.append(a[sample_idx(a.size)].astype(np.float32))
        return np.array(names), np.array(norms), np.stack(samp)
    rm = [b.detach().view(-1).numpy() for k, b in m.model.named_buffers() if k.endswith("running_mean")]
    rv = [b.detach().view(-1).numpy() for k, b in m.model.named_buffers() if k.endswith("running_var")]
    nb = [int(b) for k, b in m.model.named_buffers() if k.endswith("num_batches_tracked")]
    return np.concatenate(rm), np.concatenate(rv), np.array(nb)


def unpack_step(ret):
    if isinstance(ret, tuple):
        logs, l = ret
        out = {k: float(v) for k, v in logs.items()}
        out["loss"] = float(l["loss"])
        return out
    return {"loss": float(ret["loss"])}


def eval_logits(m, algo, batch):
    t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
    with torch.no_grad():
        o = m.model(torch.cat([t["modal1"], t["modal2"], t["rgb"]], 1))
    if isinstance(o, tuple):
        return np.concatenate([x.numpy() for x in o], 1)
    return o.numpy()


def case_train(algo, S, B, seed, steps, tag, style="xavier"):
    """eval forward at init -> `steps` training steps -> eval forward."""
    m, cfg = build(algo, seed, style=style)
    out = {}
    b0 = synthetic.make_pair_batch(seed + 100, B, S)
    m.switch_to("eval")
    out["eval0_logits"] = eval_logits(m, algo, b0)
    set_input(m, algo, b0)
    out["eval0_loss"] = np.float64(float(m.forward_only()[1]["loss"]))
    m.switch_to("train")
    for it in range(steps):
        batch = synthetic.make_pair_batch(seed + 100 + it, B, S)
        set_input(m, algo, batch)
        logs = unpack_step(m.step())
        for k, v in logs.items():
            out["step%d_%s" % (it, k)] = np.float64(v)
        if it == 0:
            n, gn, gs = snapshot(m, "g")
            out["names"], out["grad_norms"], out["grad_samples"] = n, gn, gs
        if it in (0, steps - 1):
            _, pn, ps = snapshot(m, "p")
            out["step%d_param_norms" % it], out["step%d_param_samples" % it] = pn, ps
            rm, rv, nb = snapshot(m, "bn")
            out["step%d_running_mean" % it], out["step%d_running_var" % it] = rm, rv
            out["step%d_num_batches" % it] = nb
    m.switch_to("eval")
    out["eval1_logits"] = eval_logits(m, algo, b0)
    set_input(m, algo, b0)
    out["eval1_loss"] = np.float64(float(m.forward_only()[1]["loss"]))
    out["meta"] = np.array([S, B, seed, steps])
    out["lr"] = np.float64(cfg["lr"])
    out["weight_decay"] = np.float64(cfg["weight_decay"])
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, {k: float(v) for k, v in out.items() if k.endswith("loss")})


def _ws2_worker(rank, world, port, S, B, seed, q):
    install_shims()
    init_dist(rank, world, port)
    algo = "InstaOrderNet_o"
    # every rank is CONSTRUCTED with different weights (the reference initialises randomly inside
    # SingleStageModel.__init__, before DistModule wraps the net); DistModule must broadcast rank 0's.
    import models
    import utils as ref_utils
    sd = synthetic.make_state_dict(seed + rank * 7, 5, 2)

    def seeded_init(net, init_type="xavier", init_gain=0.02):
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)

    ref_utils.init_weights = seeded_init
    cfg = model_cfg(algo)
    m = getattr(models, algo)(cfg, dist_model=True)
    batch = synthetic.make_pair_batch(seed + 200 + rank, B, S)
    m.switch_to("train")
    set_input(m, algo, batch)
    logs = unpack_step(m.step())
    _, pn, ps = snapshot(m, "p")
    _, gn, gs = snapshot(m, "g")
    rm, rv, nb = snapshot(m, "bn")
    q.put((rank, logs["loss"], pn, ps, gn, gs, rm, rv))


def case_ws2(S, B, seed, tag):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ws2_worker, args=(r, 2, 29533, S, B, seed, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get() for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join()
    out = {"meta": np.array([S, B, seed, 2])}
    for r in res:
        rank = r[0]
        out["rank%d_loss" % rank] = np.float64(r[1])
        out["rank%d_param_norms" % rank], out["rank%d_param_samples" % rank] = r[2], r[3]
        out["rank%d_grad_norms" % rank], out["rank%d_grad_samples" % rank] = r[4], r[5]
        out["rank%d_running_mean" % rank], out["rank%d_running_var" % rank] = r[6], r[7]
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, out["rank0_loss"], out["rank1_loss"])


def case_plumbing(algo, seed, n_images, n_inst, S, warm_steps, tag):
    """config 1: images x instances -> O(n^2) pair loop of the reference's
    inference.py ('image' mode) -> order matrices -> metrics."""
    import inference as infer
    m, cfg = build(algo, seed, dist_model=False, style="kaiming")
    # train-mode forward passes (no optimiser step) so that the BN running statistics
    # are meaningful while the head stays undecided (decisions on both sides of 0.5)
    m.switch_to("train")
    for it in range(warm_steps):
        b = synthetic.make_pair_batch(seed + 300 + it, 8, S)
        with torch.no_grad():
            m.model(torch.cat([torch.from_numpy(b["modal1"]), torch.from_numpy(b["modal2"]),
                               torch.from_numpy(b["rgb"])], 1))
    m.switch_to("eval")
    items = synthetic.make_images(seed + 400, n_images, n_inst, S)
    out = {"meta": np.array([S, n_images, n_inst, seed, warm_steps])}
    # centre the head: bias := -median(logit) per output over all pairs/directions, so the
    # decisions are not all on one side of 0.5.  The biases are INPUT data, stored below.
    zs = []
    with torch.no_grad():
        for item in items:
            rgb, masks = synthetic.image_mode_inputs(item["image"], item["modal"], S)
            for i in range(n_inst):
                for j in range(n_inst):
                    if i != j:
                        o = m.model(torch.cat([torch.from_numpy(masks[i])[None, None],
                                               torch.from_numpy(masks[j])[None, None], torch.from_numpy(rgb)], 1))
                        zs.append(torch.cat(o, 1)[0].numpy() if isinstance(o, tuple) else o[0].numpy())
    med = -np.median(np.stack(zs), 0).astype(np.float32)
    net = m.model.module
    with torch.no_grad():
        if algo == "InstaOrderNet_o":
            net.fc.bias.copy_(torch.from_numpy(med))
        else:
            net.fc_occ.bias.copy_(torch.from_numpy(med[:2]))
            net.fc_depth.bias.copy_(torch.from_numpy(med[2:]))
    out["head_bias"] = med
    for ii, item in enumerate(items):
        if algo == "InstaOrderNet_o":
            om = infer.infer_order_sup_occ(m, item["image"], item["modal"], item["bboxes"], "all",
                                           algo, "image", S, True)
            out["occ_%d" % ii] = om
        else:
            om, dm = infer.infer_order_sup_occ_depth(m, item["image"], item["modal"], item["bboxes"],
                                                     "all", algo, "image", S, "")
            out["occ_%d" % ii], out["depth_%d" % ii] = om, dm
            w = infer.eval_depth_order_whdr(dm, (item["gt_depth"], item["gt_overlap"], item["gt_count"]))
            keys = sorted(w.keys())
            out["whdr_keys"] = np.array(keys)
            out["whdr_%d" % ii] = np.array([float(w[k][0]) for k in keys])
        out["prf_%d" % ii] = np.array(infer.eval_order_recall_precision_f1(om, item["gt_occ"], 0))
        # the direction-averaged probabilities behind the decisions (for near-threshold slack)
        rgb, masks = synthetic.image_mode_inputs(item["image"], item["modal"], S)
        probs = []
        with torch.no_grad():
            for i in range(n_inst):
                for j in range(i + 1, n_inst):
                    mi = torch.from_numpy(masks[i])[None, None]
                    mj = torch.from_numpy(masks[j])[None, None]
                    r = torch.from_numpy(rgb)
                    o1 = m.model(torch.cat([mi, mj, r], 1))
                    o2 = m.model(torch.cat([mj, mi, r], 1))
                    if isinstance(o1, tuple):
                        o1 = torch.cat(o1, 1)
                        o2 = torch.cat(o2, 1)
                    probs.append(np.concatenate([o1.numpy()[0], o2.numpy()[0]]))
        out["pair_logits_%d" % ii] = np.stack(probs)
    _, pn, ps = snapshot(m, "p")
    out["param_norms"] = pn
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, [out["occ_%d" % i].tolist() for i in range(n_images)])


def case_scheduler(tag):
    import utils
    out = {}
    p = [torch.nn.Parameter(torch.zeros(1))]
    its = [0, 1, 100, 31999, 32000, 32001, 47999, 48000, 85999]
    opt = torch.optim.SGD(p, lr=0.001, momentum=0.9, weight_decay=1e-4)
    s = utils.StepLRScheduler(opt, [32000, 48000], [0.1, 0.1], 0.001, [], [], -1)
    lrs = []
    for it in its:
        s.step(it)
        lrs.append(opt.param_groups[0]["lr"])
    out["its"], out["lrs_plain"] = np.array(its), np.array(lrs, np.float64)
    opt = torch.optim.SGD(p, lr=0.001, momentum=0.9, weight_decay=1e-4)
    s = utils.StepLRScheduler(opt, [300, 600], [0.1, 0.5], 0.001, [0.004, 0.01], [50, 200], -1)
    its2 = [0, 10, 49, 50, 51, 125, 199, 200, 299, 300, 599, 600, 1000]
    lrs = []
    for it in its2:
        s.step(it)
        lrs.append(opt.param_groups[0]["lr"])
    out["its_warm"], out["lrs_warm"] = np.array(its2), np.array(lrs, np.float64)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, out["lrs_plain"], out["lrs_warm"])


def case_decisions(tag):
    """Direction-averaged decision rules on crafted logits, incl. near-threshold
    values and argmax ties, through the reference's own net_forward_* functions."""
    import inference as infer
    rng = np.random.RandomState(7)
    n = 64
    occ1 = rng.uniform(-0.2, 0.2, (n, 2)).astype(np.float32)
    occ2 = rng.uniform(-0.2, 0.2, (n, 2)).astype(np.float32)
    dep1 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    dep2 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    # exact symmetric cases: averaged probability exactly 0.5 -> not ">" 0.5
    occ1[0], occ2[0] = [0.3, -0.7], [0.7, -0.3]
    occ1[1], occ2[1] = [0.0, 0.0], [0.0, 0.0]
    dep1[0], dep2[0] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]        # three-way tie -> class 0
    dep1[1], dep2[1] = [1.0, 0.0, 1.0], [0.0, 1.0, 1.0]        # tie closer/equal

    class Fake:
        def __init__(self):
            self.k = 0
            self.calls = 0

        def model(self, x):
            first = (self.calls % 2 == 0)
            self.calls += 1
            o = torch.from_numpy((occ1 if first else occ2)[self.k:self.k + 1])
            d = torch.from_numpy((dep1 if first else dep2)[self.k:self.k + 1])
            return (o, d) if self.mode == "od" else (o if self.mode == "o" else d)

    img = torch.zeros(1, 3, 8, 8)
    mk = np.zeros((8, 8), np.float32)
    fk = Fake()
    res_o, res_od, res_d = [], [], []
    for k in range(n):
        fk.k = k
        fk.mode, fk.calls = "o", 0
        res_o.append(infer.net_forward_occ(fk, img, mk, mk, True))
        fk.mode, fk.calls = "od", 0
        res_od.append(infer.net_forward_occ_depth(fk, img, mk, mk))
        fk.mode, fk.calls = "d", 0
        res_d.append(infer.net_forward_depth(fk, img, mk, mk, True))
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), occ1=occ1, occ2=occ2, dep1=dep1, dep2=dep2,
                        res_o=np.array(res_o).astype(np.int64), res_od=np.array(res_od).astype(np.int64),
                        res_d=np.array(res_d).astype(np.int64))
    print(tag, np.array(res_o)[:4].tolist(), np.array(res_od)[:4].tolist())


def case_decisions_ordernet(tag):
    """net_forward_OrderNet (inference.py:44-76) on crafted 3- and 4-class logits incl. ties, and the reference's
    sklearn-based eval_order_recall_precision_f1 (inference.py:794-802) on matrices that hit every zero-division
    corner (no positives at all; tp = 0 beside false positives / negatives) with zd = 0 and 1."""
    import inference as infer
    rng = np.random.RandomState(9)
    n = 48
    out = {}
    for K in (3, 4):
        a = rng.uniform(-1, 1, (n, K)).astype(np.float32)
        b = rng.uniform(-1, 1, (n, K)).astype(np.float32)
        a[0], b[0] = 0.25, 0.25                                   # all classes tie -> index 0 (1 over 2)
        a[1], b[1] = [0, 0, 1] + [0] * (K - 3), [0, 0, 1] + [0] * (K - 3)
        if K == 4:
            a[2], b[2] = [0, 0, 0, 2], [0, 0, 0, 2]               # both
            a[3], b[3] = [0, 0, 1, 1], [0, 0, 1, 1]               # none ties both -> none

        class Fake:
            calls, k = 0, 0

            def model(self, x):
                first = self.calls % 2 == 0
                self.calls += 1
                return torch.from_numpy((a if first else b)[self.k:self.k + 1])

        fk, res = Fake(), []
        img, mk = torch.zeros(1, 3, 8, 8), np.zeros((8, 8), np.float32)
        for k in range(n):
            fk.k, fk.calls = k, 0
            res.append(infer.net_forward_OrderNet(fk, img, mk, mk))
        out["l1_%d" % K], out["l2_%d" % K], out["res_%d" % K] = a, b, np.array(res).astype(np.int64)
    gts, prs, scores = [], [], []
    cases = [([[-1, 0], [0, -1]], [[0, 0], [0, 0]]),      # nothing to count at all
             ([[-1, 1], [0, -1]], [[0, 0], [1, 0]]),      # tp = 0, fp = 1, fn = 1 (pair predicted the wrong way round)
             ([[-1, 1], [1, -1]], [[0, 0], [0, 0]]),      # tp = 0, fn = 2, no predicted positive
             ([[-1, 0], [0, -1]], [[0, 1], [1, 0]]),      # tp = 0, fp = 2, no true positive
             ([[-1, 1], [0, -1]], [[0, 1], [0, 0]])]      # perfect
    for gt, pr in cases:
        gt, pr = np.array(gt), np.array(pr)
        gts.append(gt)
        prs.append(pr)
        scores.append([infer.eval_order_recall_precision_f1(pr, gt, zd) for zd in (0, 1)])
    for nn in (4, 7):
        for _ in range(6):
            gt, pr = rng.randint(-1, 2, (nn, nn)), rng.randint(0, 2, (nn, nn))
            gts.append(np.pad(gt, ((0, 7 - nn), (0, 7 - nn)), constant_values=-1))
            prs.append(np.pad(pr, ((0, 7 - nn), (0, 7 - nn))))
            scores.append([infer.eval_order_recall_precision_f1(pr, gt, zd) for zd in (0, 1)])
    gts = [np.pad(g, ((0, 7 - g.shape[0]), (0, 7 - g.shape[0])), constant_values=-1) for g in gts]
    prs = [np.pad(p, ((0, 7 - p.shape[0]), (0, 7 - p.shape[0]))) for p in prs]
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), gt=np.stack(gts), pred=np.stack(prs),
                        scores=np.array(scores, np.float64), **out)
    print(tag, out["res_3"][:3].tolist(), out["res_4"][:4].tolist(), np.array(scores)[:5].tolist())


BWD_TENSORS = ["conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer1.0.bn2.weight",
               "layer1.0.downsample.0.weight", "layer1.2.conv3.weight", "layer2.0.conv2.weight", "layer2.0.bn3.bias",
               "layer2.0.downsample.0.weight", "layer2.3.conv1.weight", "layer3.0.conv1.weight", "layer3.2.bn1.weight",
               "layer3.5.conv2.weight", "layer3.5.conv3.weight", "layer4.0.conv2.weight", "layer4.0.downsample.1.weight",
               "layer4.2.conv1.weight", "layer4.2.conv3.weight", "fc.weight", "fc.bias"]
BWD_MAX = 1 << 16         # elements stored per tensor (an evenly strided subset of larger ones)


def bwd_subset(n):
    return np.arange(n) if n <= BWD_MAX else (np.arange(BWD_MAX, dtype=np.int64) * n) // BWD_MAX


def case_backward(tag, algo="InstaOrderNet_o", S=64, B=16, seed=61, pre=10, lr=None):
    """Backward parity on the REAL network (ReLUs switching) in a WELL-CONDITIONED state.  At a random initialisation
    the reference's own fp32 gradients sit 1-2 % from an fp64 evaluation of the same graph, whatever the batch
    (DESIGN.md section 4); ten SGD steps of the reference's recipe away from its initialisation (xavier gain 0.02,
    lr 1e-3, momentum 0.9, wd 1e-4) they agree to ~1e-6.  So: `pre` reference steps in fp32, then the gradient of one
    more batch twice -- in fp32 (what the reference ships) and in fp64 from the same weights (the anchor).  Stored per
    parameter tensor: the fp64 gradient norm, the reference's fp32-vs-fp64 distance, norms of the pre-stepped weights
    (so that a test can rebuild that state with the CPU oracle and check it), and the fp64 gradient itself (rounded to
    fp32) on BWD_TENSORS."""
    m, cfg = build(algo, seed, style="xavier")
    if lr is not None:          # (InstaOrderNet_od's config starts at 1e-4, where ten steps do not get far enough)
        cfg = dict(cfg, lr=lr)
        for g in m.optim.param_groups:
            g["lr"] = lr
    m.switch_to("train")
    for it in range(pre):
        set_input(m, algo, synthetic.make_pair_batch(seed + 300 + it, B, S))
        m.step()
    state = {k: v.detach().clone() for k, v in m.model.state_dict().items()}

    def run(double):
        mm, _ = build(algo, seed, style="xavier")
        if double:
            mm.model.double()
        mm.model.load_state_dict({k: (v.double() if (double and v.is_floating_point()) else v.clone())
                                  for k, v in state.items()}, strict=True)
        batch = synthetic.make_pair_batch(seed + 100, B, S)
        if double:
            batch = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in batch.items()}
        mm.switch_to("train")
        for g in mm.optim.param_groups:
            g["lr"] = 0.0
        set_input(mm, algo, batch)
        logs = unpack_step(mm.step())
        return logs, [(k[len("module."):], p.grad.detach().double().numpy().copy()) for k, p in mm.model.named_parameters()]
    l32, g32 = run(False)
    l64, g64 = run(True)
    names = [k for k, _ in g64]
    n64 = np.array([np.sqrt((g * g).sum()) for _, g in g64])
    dist = np.array([np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-300)
                     for (_, a), (_, b) in zip(g32, g64)])
    _, pn, ps = snapshot(m, "p")
    rm, rv, nb = snapshot(m, "bn")
    out = dict(names=np.array(names), norms64=n64, ref_dist=dist, loss32=np.float64(l32["loss"]),
               loss64=np.float64(l64["loss"]), meta=np.array([S, B, seed, pre]), tensors=np.array(BWD_TENSORS),
               lr=np.float64(cfg["lr"]), weight_decay=np.float64(cfg["weight_decay"]),
               pre_param_norms=pn, pre_param_samples=ps, pre_running_mean=rm, pre_running_var=rv)
    d32 = dict(g32)
    tensors = [k for k in BWD_TENSORS + ["fc_occ.weight", "fc_occ.bias", "fc_depth.weight", "fc_depth.bias"] if k in names]
    out["tensors"] = np.array(tensors)
    for k, g in g64:
        if k in tensors:
            idx = bwd_subset(g.size)
            sub64, sub32 = g.reshape(-1)[idx], d32[k].reshape(-1)[idx]
            out["g64/" + k] = sub64.astype(np.float32)
            out["refsub/" + k] = np.float64(np.sqrt(((sub32 - sub64) ** 2).sum()) / np.sqrt((sub64 ** 2).sum()))
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, "loss32 %.7f loss64 %.7f" % (l32["loss"], l64["loss"]),
          "reference fp32-vs-fp64 gradient distance per tensor: median %.2e max %.2e" % (np.median(dist), dist.max()))
    for k in tensors:
        print("   %-32s ref dist %.2e (subset %.2e)" % (k, dist[names.index(k)], float(out["refsub/" + k])))


def case_checkpoint(tag):
    """single_stage_model.py:54-72 / common_utils.py:128-149: a checkpoint WRITTEN BY THE REFERENCE (its own
    ``save_state`` on its own model + torch.optim.SGD, filled with instaorder_amd.synthetic.make_checkpoint_state) is
    loaded into the instaorder_amd model here, on the CPU, through ``load_state(..., resume=True)``; the digest of what
    arrived in the flat buffers is the fixture (the 190 MB file itself is not).  The reverse direction is asserted
    on the spot: a file written by the package's ``save_state`` loads into the reference model + optimiser, bit-exact."""
    import shutil
    import tempfile
    import instaorder_amd as ia
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import checkpoint_digest              # the definition the tests use
    algo, seed = "InstaOrderNet_od", 51
    m, cfg = build(algo, seed, style="kaiming")
    sd, mom, lr, step = synthetic.make_checkpoint_state(seed, 5, [2, 3])
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    params = m.optim.param_groups[0]["params"]
    assert len(params) == len(mom) == 163
    for p, b in zip(params, mom):
        assert tuple(p.shape) == b.shape
        m.optim.state[p]["momentum_buffer"] = torch.from_numpy(b.copy())
    m.optim.param_groups[0]["lr"] = lr
    d = tempfile.mkdtemp()
    try:
        m.save_state(d, step)                                            # the reference writes the file
        ck = torch.load(os.path.join(d, "ckpt_iter_%d.pth.tar" % step), map_location="cpu", weights_only=False)
        layout = dict(keys=np.array(list(ck["state_dict"].keys())),
                      shapes=np.array([",".join(str(x) for x in v.shape) for v in ck["state_dict"].values()]),
                      dtypes=np.array([str(v.dtype) for v in ck["state_dict"].values()]),
                      group_keys=np.array(sorted(k for k in ck["optimizer"]["param_groups"][0] if k != "params")),
                      n_state=np.int64(len(ck["optimizer"]["state"])),
                      state_keys=np.array(sorted(ck["optimizer"]["state"][0].keys())))
        mine = ia.InstaOrderNet_od(dict(cfg, dtype="fp32"), dist_model=False)
        assert mine.load_state(d, step, resume=True) is None             # (the wrapper returns nothing, as the reference)
        dg = checkpoint_digest(mine)
        # ... and the values really are the seeded ones, tensor by tensor
        for (k, v), (k2, v2) in zip(mine.model.state_dict().items(), sd.items()):
            assert k == k2 and np.array_equal(v.cpu().numpy(), v2), k
        for v, b in zip(mine.optim._views, mom):
            assert np.array_equal(v.cpu().numpy(), b)
        # reverse: package -> reference
        d2 = tempfile.mkdtemp()
        mine.save_state(d2, step + 1)
        m2, _ = build(algo, seed + 1, style="xavier")
        # the reference maps every storage with .cuda() (common_utils.py:129-130); no GPU here: identity, as the shims
        # above already do for tensors and modules
        torch.UntypedStorage.cuda = lambda self, *a, **k: self
        torch.storage.TypedStorage.cuda = lambda self, *a, **k: self
        m2.load_state(d2, step + 1, resume=True)
        for (k, v), v2 in zip(m2.model.state_dict().items(), sd.values()):
            assert np.array_equal(v.numpy(), v2), k
        for p, b in zip(m2.optim.param_groups[0]["params"], mom):
            assert np.array_equal(m2.optim.state[p]["momentum_buffer"].numpy(), b)
        assert m2.optim.param_groups[0]["lr"] == lr
        shutil.rmtree(d2)
    finally:
        shutil.rmtree(d)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), meta=np.array([seed, step]), **layout,
                        **{k: (np.array(v) if isinstance(v, str) else v) for k, v in dg.items()})
    print(tag, dg["sha_params"][:16], dg["sha_momentum"][:16], "lr", float(dg["lr"]))


# ---- MiDaS-based nets -----------------------------------------------------------------------------
DEPTH_LOSS_WEIGHTS = dict(overlap_weight=0.1, distinct_weight=0.9, dorder_weight=1.0, smooth_weight=0.1,
                          occ_order_weight=1.0)


def case_depthnet(algo, S, B, seed, tag):
    """InstaDepthNet_od / _d of the unmodified reference: torch.hub.load (network fetch of the un-vendored WSL
    ResNeXt, midas/blocks.py:85-87) is replaced by the reference's own resnext101_32x8d (same architecture,
    SURVEY 8(c)); no pretrained MiDaS weights exist here, so weights are synthetic."""
    from models.backbone import resnet_cls as ref_resnet
    torch.hub.load = lambda repo, name, **kw: ref_resnet.resnext101_32x8d(in_channels=3)
    import models
    cfg = model_cfg(algo)
    cfg["pretrained_weight"] = None
    cfg.update(DEPTH_LOSS_WEIGHTS)
    cfg["lr"] = 1e-3
    m = getattr(models, algo)(cfg, dist_model=True)
    ref_sd = m.model.state_dict()
    first, spec = {}, []
    for k, v in ref_sd.items():
        kk = k[len("module."):]
        ptr = v.data_ptr() if v.numel() else id(v)
        alias = first.get(ptr) if v.dim() > 0 else None
        if alias is None and v.dim() > 0:
            first[ptr] = kk
        spec.append((kk, tuple(v.shape), alias))
    sd = synthetic.make_spec_state_dict(seed, spec, prefix="module.")
    m.model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    out = {"keys": np.array([k for k, _, _ in spec]),
           "shapes": np.array([",".join(str(d) for d in sh) for _, sh, _ in spec]),
           "aliases": np.array([a or "" for _, _, a in spec])}
    batch = synthetic.make_depth_batch(seed + 100, B, S)
    t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}

    def feed():
        if algo == "InstaDepthNet_od":
            m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
        else:
            m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"])

    m.switch_to("eval")
    with torch.no_grad():
        d, dep, occ = m.model(t["rgb"], t["modal1"], t["modal2"])
    out["eval_disp"], out["eval_dep"] = d.numpy(), dep.numpy()
    if occ is not None:
        out["eval_occ"] = occ.numpy()
    feed()
    logs, l = m.forward_only()
    for k, v in logs.items():
        out["evalfo_" + k] = np.float64(float(v))
    out["evalfo_loss"] = np.float64(float(l["loss"]))
    m.switch_to("train")
    feed()
    logs, l = m.step()
    for k, v in logs.items():
        out["step_" + k] = np.float64(float(v))
    out["step_loss"] = np.float64(float(l["loss"]))
    names, gn, gs = [], [], []
    for k, p in m.model.named_parameters():
        g = p.grad
        a = (g if g is not None else torch.zeros_like(p)).detach().contiguous().view(-1).double().numpy()
        names.append(k[len("module."):])
        gn.append(np.sqrt((a * a).sum()))
        gs.append(a[sample_idx(a.size)].astype(np.float32))
    out["names"], out["grad_norms"], out["grad_samples"] = np.array(names), np.array(gn), np.stack(gs)
    _, pn, ps = snapshot(m, "p")
    out["step_param_norms"], out["step_param_samples"] = pn, ps
    rm, rv, nb = snapshot(m, "bn")
    out["step_running_mean"], out["step_running_var"], out["step_num_batches"] = rm, rv, nb
    out["meta"] = np.array([S, B, seed])
    out["lr"], out["weight_decay"] = np.float64(cfg["lr"]), np.float64(cfg["weight_decay"])
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, {k: float(v) for k, v in out.items() if k.startswith("step_loss") or k.startswith("evalfo_loss")},
          "params", len(names), "state entries", len(spec))


DATASET_VARIANTS = [   # (name, dataset class, algo, patch_or_image, phase, seed)
    ("o_patch_train", "occ", "InstaOrderNet_o", "patch", "train", 101),
    ("o_image_train", "occ", "InstaOrderNet_o", "image", "train", 102),
    ("ordernet_patch_val", "occ", "OrderNet", "patch", "val", 103),
    ("od_patch_train", "depth_occ", "InstaOrderNet_od", "patch", "train", 104),
    ("od_resize_train", "depth_occ", "InstaOrderNet_od", "resize", "train", 105),
    ("d_patch_train", "depth", "InstaOrderNet_d", "patch", "train", 106),
]
DATASET_S = 40
DATASET_READER_SEED = 77


def dataset_config(mode):
    """the `data` sections of experiments/InstaOrder/InstaOrderNet_{o,od}/config.yaml with a small input size"""
    cfg = dict(load_cfg("InstaOrderNet_o")["data"])
    cfg.update(load_cfg("InstaOrderNet_od")["data"])
    cfg.update(input_size=DATASET_S, patch_or_image=mode, extend_bidirec=True)
    return cfg


def case_dataset_items(tag):
    """Items of the reference's own SupOcclusionOrderDataset / SupDepthOccOrderDataset over synthetic annotations
    (synthetic.SyntheticReader in place of the COCO json reader) with ``cv2.resize`` bound to the oracle's restatement
    of it (cv2 is not installed here): pins crop arithmetic, padding, flip, normalisation, label layout and the order
    of the np.random draws."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import preprocess_oracle as po
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC = po.INTER_NEAREST, po.INTER_LINEAR, po.INTER_CUBIC

    def cv2_resize(img, size, interpolation=po.INTER_LINEAR):
        if img.dtype == np.float64:
            assert interpolation == po.INTER_CUBIC
            return po.resize_cubic_f64(img, size)
        return po.resize(img, size, interpolation)

    cv2.resize = cv2_resize
    from datasets import reader as ref_reader
    from datasets import occ_order_dataset, depth_occ_order_dataset, depth_order_dataset
    rd = synthetic.SyntheticReader(DATASET_READER_SEED)
    ref_reader.InstaOrderDataset = lambda annot_fn: rd
    out = {}
    for name, kind, algo, mode, phase, seed in DATASET_VARIANTS:
        cls = {"occ": occ_order_dataset.SupOcclusionOrderDataset,
               "depth_occ": depth_occ_order_dataset.SupDepthOccOrderDataset,
               "depth": depth_order_dataset.SupDepthOrderDataset}[kind]
        ds = cls(dataset_config(mode), phase, algo)
        ds._load_image = lambda fn: rd.load_image(fn)
        np.random.seed(seed)
        n = min(len(ds), 8)
        items = [ds[i] for i in range(n)]
        for f in range(len(items[0])):
            col = [np.asarray(it[f].numpy() if torch.is_tensor(it[f]) else it[f]) for it in items]
            out["%s_f%d" % (name, f)] = np.stack(col)
        print(name, "items", n, "fields", len(items[0]))
    # the 'resize' inference transform (utils/data_utils.py:37-53 through midas/transforms.py) on two scenes
    from utils.data_utils import transform_resize
    for k, (w, h) in enumerate([(64, 64), (96, 64)]):
        out["transform_resize_%d" % k] = transform_resize(rd.scenes[k]["image"], w, h)
        print("transform_resize", k, out["transform_resize_%d" % k].shape)
    import json
    out["config_json"] = np.array(json.dumps(dataset_config("patch")))
    out["variants"] = np.array(["|".join(str(v) for v in row) for row in DATASET_VARIANTS])
    out["reader_seed"] = np.array(DATASET_READER_SEED)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)


def case_heuristics(tag):
    """inference.py's annotation-free baselines, infer_gt_order and eval_order on synthetic scenes; cv2.dilate (absent)
    is scipy's binary_dilation with the same cross."""
    from scipy import ndimage
    cv2 = sys.modules["cv2"]
    cv2.dilate = lambda a, k, iterations=1: ndimage.binary_dilation(a.astype(bool), structure=k.astype(bool),
                                                                     iterations=iterations).astype(np.uint8)
    import inference as ref_inf
    rd = synthetic.SyntheticReader(88, n_images=4, n_inst=6, empty_every=0)
    out = {}
    for k, sc in enumerate(rd.scenes):
        m = sc["modal"]
        rng = np.random.RandomState(k)
        amodal = np.stack([ndimage.binary_dilation(x.astype(bool), iterations=int(rng.randint(1, 6))).astype(np.uint8)
                           for x in m])
        out["amodal_%d" % k] = amodal
        out["occ_area_s_%d" % k] = ref_inf.infer_occ_order_area(m, "smaller")
        out["occ_area_l_%d" % k] = ref_inf.infer_occ_order_area(m, "larger")
        out["occ_y_lo_%d" % k] = ref_inf.infer_occ_order_yaxis(m, "lower")
        out["occ_y_hi_%d" % k] = ref_inf.infer_occ_order_yaxis(m, "higher")
        out["dep_area_s_%d" % k] = ref_inf.infer_depth_order_area(m, "smaller")
        out["dep_area_l_%d" % k] = ref_inf.infer_depth_order_area(m, "larger")
        out["dep_y_lo_%d" % k] = ref_inf.infer_depth_order_yaxis(m, "lower")
        out["dep_y_hi_%d" % k] = ref_inf.infer_depth_order_yaxis(m, "higher")
        gt = ref_inf.infer_gt_order(m, amodal)
        out["gt_%d" % k] = gt
        ev = ref_inf.eval_order(out["occ_area_s_%d" % k], gt)
        out["eval_%d" % k] = np.asarray(ev[:4], np.float64)
        out["eval_err_%d" % k] = ev[4]
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, "scenes", len(rd.scenes), "gt ones", [int(out["gt_%d" % k].sum()) for k in range(len(rd.scenes))])


TESTER_SCENARIOS = [   # (name, trainval_dataset, order_method, patch_or_image, algo or None)
    ("occ_area", "SupOcclusionOrderDataset", "area", "patch", None),
    ("occ_yaxis", "SupOcclusionOrderDataset", "yaxis", "patch", None),
    ("dep_area", "SupDepthOrderDataset", "area", "resize", None),
    ("dep_yaxis", "SupDepthOrderDataset", "yaxis", "resize", None),
    ("occ_net_patch", "SupOcclusionOrderDataset", "InstaOrderNet_o", "patch", "InstaOrderNet_o"),
    ("occ_net_image", "SupOcclusionOrderDataset", "InstaOrderNet_o", "image", "InstaOrderNet_o"),
    ("od_net_resize", "SupDepthOccOrderDataset", "InstaOrderNet_od", "resize", "InstaOrderNet_od"),
]
# the 'orig' mode (inference.py:401-407, 490-496): whole images at their own aspect ratio -- the scenes of the reader round
# to 128 x 128 (twice), 96 x 128 and 128 x 160 network inputs
TESTER_SCENARIOS_ORIG = [
    ("occ_net_orig", "SupOcclusionOrderDataset", "InstaOrderNet_o", "orig", "InstaOrderNet_o"),
    ("od_net_orig", "SupDepthOccOrderDataset", "InstaOrderNet_od", "orig", "InstaOrderNet_od"),
]
TESTER_S, TESTER_SEED, TESTER_READER_SEED, TESTER_WARM = 64, 31, 91, 6


def case_tester(tag, TESTER_SCENARIOS=TESTER_SCENARIOS):
    """The reference's own tools/test.py Tester loops (eval_occ_order / eval_depth_order / eval_occ_depth_order) over
    synthetic scenes: heuristics and the supervised nets in 'patch' / 'image' / 'resize' mode (cv2.resize = the oracle's
    restatement, cv2.dilate = scipy's binary_dilation).  Recorded: every predicted order matrix, the pair logits
    behind it, and the aggregated metrics the Tester logs."""
    import importlib.util
    import logging
    from argparse import Namespace
    from scipy import ndimage
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import preprocess_oracle as po
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC = po.INTER_NEAREST, po.INTER_LINEAR, po.INTER_CUBIC

    def cv2_resize(img, size, interpolation=po.INTER_LINEAR):
        if img.dtype == np.float64:
            return po.resize_cubic_f64(img, size)
        return po.resize(img, size, interpolation)

    cv2.resize = cv2_resize
    cv2.dilate = lambda a, k, iterations=1: ndimage.binary_dilation(a.astype(bool), structure=k.astype(bool),
                                                                     iterations=iterations).astype(np.uint8)
    sys.modules.setdefault("wandb", mock.MagicMock(name="wandb"))
    spec = importlib.util.spec_from_file_location("ref_tools_test", os.path.join(REF, "tools", "test.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    import inference as infer
    rd = synthetic.SyntheticReader(TESTER_READER_SEED, n_images=4, n_inst=5, empty_every=0)

    class _Img:
        def __init__(self, a):
            self.a = a

        def convert(self, mode):
            return self.a

    T.Image = types.SimpleNamespace(open=lambda path: _Img(rd.load_image(os.path.basename(path))))
    data_cfg = dict(load_cfg("InstaOrderNet_o")["data"])
    data_cfg.update(load_cfg("InstaOrderNet_od")["data"])
    data_cfg.update(input_size=TESTER_S, val_image_root="/nowhere")
    out = {"data_cfg_json": np.array(__import__("json").dumps(data_cfg)),
           "scenarios": np.array(["|".join(str(v) for v in r) for r in TESTER_SCENARIOS]),
           "meta": np.array([TESTER_S, TESTER_SEED, TESTER_READER_SEED, TESTER_WARM])}
    models_cache = {}

    def get_model(algo):
        if algo not in models_cache:
            m, _ = build(algo, TESTER_SEED, dist_model=False, style="kaiming")
            m.switch_to("train")
            for it in range(TESTER_WARM):
                b = synthetic.make_pair_batch(TESTER_SEED + 300 + it, 8, TESTER_S)
                with torch.no_grad():
                    m.model(torch.cat([torch.from_numpy(b["modal1"]), torch.from_numpy(b["modal2"]),
                                       torch.from_numpy(b["rgb"])], 1))
            m.switch_to("eval")
            models_cache[algo] = m
        return models_cache[algo]

    for name, kind, method, mode, algo in TESTER_SCENARIOS:
        cfg = dict(data_cfg, trainval_dataset=kind, patch_or_image=mode)
        t = object.__new__(T.Tester)
        t.args = Namespace(data=cfg, model=dict(use_rgb=True), order_method=method, pairs="all", zd=0, save_pngs=0,
                           disp_select_method="", order_th=0.1, load_model="x")
        T.args = t.args                                   # the loops also read the module-level `args` (test.py:423)
        t.data_reader, t.data_length, t.dataset = rd, rd.get_image_length(), "InstaOrder"
        t.data_root, t.gt_ordering, t.curr_step = "/nowhere", "ann", 5
        t.logger = logging.getLogger("tester_golden")
        logged = {}
        t.wb_logger = types.SimpleNamespace(log=lambda d, step=None: logged.update(d))
        logits = []
        hook = None
        if algo is not None:
            t.model = get_model(algo)
            net = t.model.model.module
            for b in ([net.fc] if algo == "InstaOrderNet_o" else [net.fc_occ, net.fc_depth]):
                b.bias.data.zero_()
            hook = t.model.model.register_forward_hook(
                lambda mod, inp, o: logits.append((torch.cat(o, 1) if isinstance(o, tuple) else o)[0].detach().numpy().copy()))
        run = {"SupOcclusionOrderDataset": t.eval_occ_order, "SupDepthOrderDataset": t.eval_depth_order,
               "SupDepthOccOrderDataset": t.eval_occ_depth_order}[kind]
        if algo is not None:
            run()                                         # pass 1: logits with a zero head bias -> centre the head
            med = -np.median(np.stack(logits), 0).astype(np.float32)
            if algo == "InstaOrderNet_o":
                net.fc.bias.data.copy_(torch.from_numpy(med))
            else:
                net.fc_occ.bias.data.copy_(torch.from_numpy(med[:2]))
                net.fc_depth.bias.data.copy_(torch.from_numpy(med[2:]))
            out[name + "_head_bias"] = med
            logits.clear()
            logged.clear()
        preds = []
        wrapped = {}
        for fn in ("infer_occ_order_area", "infer_occ_order_yaxis", "infer_depth_order_area", "infer_depth_order_yaxis",
                   "infer_order_sup_occ", "infer_order_sup_occ_depth", "infer_order_sup_depth"):
            orig = getattr(infer, fn)
            wrapped[fn] = orig

            def rec(*a, _o=orig, **k):
                r = _o(*a, **k)
                preds.append(r)
                return r

            setattr(infer, fn, rec)
        try:
            run()
        finally:
            for fn, orig in wrapped.items():
                setattr(infer, fn, orig)
            if hook is not None:
                hook.remove()
        for i, r in enumerate(preds):
            if isinstance(r, tuple):
                out["%s_pred_occ_%d" % (name, i)], out["%s_pred_dep_%d" % (name, i)] = np.asarray(r[0]), np.asarray(r[1])
            else:
                out["%s_pred_%d" % (name, i)] = np.asarray(r)
        if logits:
            out[name + "_logits"] = np.stack(logits)      # call order: image-major, pair-major, (a,b) then (b,a)
        for k, v in logged.items():
            if isinstance(v, (int, float, np.floating, np.integer)):
                out["%s_log_%s" % (name, k.replace("/", "."))] = np.float64(v)
        print(name, {k: round(float(v), 3) for k, v in logged.items() if isinstance(v, (float, np.floating))})
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)


def case_init_stats(tag):
    """utils/common_utils.py:35-65 as models/single_stage_model.py:24 applies it (init_weights(net, 'xavier'), gain
    0.02) to the reference's resnet50_cls: per-tensor std / mean of every parameter over `reps` seeded draws, pooled
    (the statistics are what a port must reproduce; the draws themselves depend on torch's generator)."""
    from models.backbone import resnet_cls
    from utils import common_utils
    reps = 3
    acc = {}
    for r in range(reps):
        torch.manual_seed(100 + r)
        net = resnet_cls.resnet50_cls(in_channels=5, num_classes=[2, 3])
        common_utils.init_weights(net, init_type="xavier")
        for k, p in net.named_parameters():
            a = p.detach().double().reshape(-1)
            acc.setdefault(k, []).append((float(a.mean()), float((a * a).mean()), a.numel()))
    names = list(acc)
    mean = np.array([np.mean([m for m, _, _ in acc[k]]) for k in names])
    rms = np.array([np.sqrt(np.mean([q for _, q, _ in acc[k]])) for k in names])
    numel = np.array([acc[k][0][2] for k in names])
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), names=np.array(names), mean=mean, rms=rms, numel=numel,
                        reps=np.int64(reps))
    print(tag, "conv1.weight rms %.3e  bn1.weight mean %.4f rms-1 %.3e  fc_occ.weight rms %.3e" % (
        rms[names.index("conv1.weight")], mean[names.index("bn1.weight")],
        np.sqrt(max(rms[names.index("bn1.weight")] ** 2 - mean[names.index("bn1.weight")] ** 2, 0)),
        rms[names.index("fc_occ.weight")]))


CASES = {
    "init_stats": lambda: case_init_stats("init_stats"),
    # well-scaled states (eval logits O(0.1 .. 1), informative losses) at the bench's and the reference _od's input size
    "o_S256_B4_k": lambda: case_train("InstaOrderNet_o", 256, 4, 25, 1, "o_S256_B4_k", "kaiming"),
    "od_S256_B4_k": lambda: case_train("InstaOrderNet_od", 256, 4, 26, 1, "od_S256_B4_k", "kaiming"),
    "od_S384_B2_k": lambda: case_train("InstaOrderNet_od", 384, 2, 27, 1, "od_S384_B2_k", "kaiming"),
    "tester": lambda: case_tester("tester"),
    "tester_orig": lambda: case_tester("tester_orig", TESTER_SCENARIOS_ORIG),
    "heuristics": lambda: case_heuristics("heuristics"),
    "dataset_items": lambda: case_dataset_items("dataset_items"),
    "o_S64_B4": lambda: case_train("InstaOrderNet_o", 64, 4, 11, 3, "o_S64_B4"),
    "od_S64_B6": lambda: case_train("InstaOrderNet_od", 64, 6, 12, 3, "od_S64_B6"),
    "d_S64_B6": lambda: case_train("InstaOrderNet_d", 64, 6, 13, 1, "d_S64_B6"),
    "ordernet_S64_B4": lambda: case_train("OrderNet", 64, 4, 14, 1, "ordernet_S64_B4"),
    "o_S64_B4_k": lambda: case_train("InstaOrderNet_o", 64, 4, 21, 2, "o_S64_B4_k", "kaiming"),
    "od_S64_B6_k": lambda: case_train("InstaOrderNet_od", 64, 6, 22, 2, "od_S64_B6_k", "kaiming"),
    "o_S256_B4": lambda: case_train("InstaOrderNet_o", 256, 4, 15, 1, "o_S256_B4"),
    "od_S256_B4": lambda: case_train("InstaOrderNet_od", 256, 4, 16, 1, "od_S256_B4"),
    "ws2_o_S64_B4": lambda: case_ws2(64, 4, 17, "ws2_o_S64_B4"),
    "plumbing_o": lambda: case_plumbing("InstaOrderNet_o", 18, 4, 3, 256, 6, "plumbing_o"),
    "plumbing_od": lambda: case_plumbing("InstaOrderNet_od", 19, 2, 4, 256, 6, "plumbing_od"),
    "scheduler": lambda: case_scheduler("scheduler"),
    "decisions": lambda: case_decisions("decisions"),
    "decisions_ordernet": lambda: case_decisions_ordernet("decisions_ordernet"),
    "checkpoint_od": lambda: case_checkpoint("checkpoint_od"),
    "backward_o_S64_B16": lambda: case_backward("backward_o_S64_B16"),
    "backward_od_S128_B16": lambda: case_backward("backward_od_S128_B16", "InstaOrderNet_od", 128, 16, 62, 10, 1e-3),
    "depthnet_od_S64_B2": lambda: case_depthnet("InstaDepthNet_od", 64, 2, 31, "depthnet_od_S64_B2"),
    "depthnet_d_S64_B2": lambda: case_depthnet("InstaDepthNet_d", 64, 2, 32, "depthnet_d_S64_B2"),
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    install_shims()
    if any(c != "ws2_o_S64_B4" for c in which):
        init_dist(0, 1, 29531)
    torch.manual_seed(0)
    for c in which:
        if c == "ws2_o_S64_B4":
            continue
        CASES[c]()
    if "ws2_o_S64_B4" in which:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        CASES["ws2_o_S64_B4"]()
