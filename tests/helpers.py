"""Shared helpers for the parity tests (test-only; may import oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from instaorder_amd import synthetic  # noqa: E402
from oracle import resnet_oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NSAMP = 64

ALGO_CLASSES = {"InstaOrderNet_o": 2, "InstaOrderNet_od": [2, 3], "InstaOrderNet_d": 3, "OrderNet": 3}
ALGO_LR = {"InstaOrderNet_o": 1e-3, "InstaOrderNet_od": 1e-4, "InstaOrderNet_d": 1e-4, "OrderNet": 1e-3}


def sample_idx(n):
    return (np.arange(NSAMP, dtype=np.int64) * 2654435761) % max(n, 1)


def load_golden(tag):
    return np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)


def oracle_state(seed, algo, style="xavier"):
    return orc.state_from_numpy(synthetic.make_state_dict(seed, 5, ALGO_CLASSES[algo], style=style))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def norms_and_samples(tensors):
    """tensors: iterable of torch tensors in OIHW logical order."""
    norms, samp = [], []
    for t in tensors:
        a = t.detach().contiguous().reshape(-1).double().cpu().numpy()
        norms.append(np.sqrt((a * a).sum()))
        samp.append(a[sample_idx(a.size)].astype(np.float32))
    return np.array(norms), np.stack(samp)


def bn_vectors(state):
    rm = np.concatenate([v.reshape(-1).cpu().numpy() for k, v in state.items() if k.endswith("running_mean")])
    rv = np.concatenate([v.reshape(-1).cpu().numpy() for k, v in state.items() if k.endswith("running_var")])
    nb = np.array([int(v) for k, v in state.items() if k.endswith("num_batches_tracked")])
    return rm, rv, nb


def eval_logits_oracle(state, batch):
    tb = {k: torch.as_tensor(v) for k, v in batch.items()}
    with torch.no_grad():
        o = orc.resnet_forward(state, torch.cat([tb["modal1"], tb["modal2"], tb["rgb"]], 1), False)
    if isinstance(o, tuple):
        return torch.cat(o, 1).numpy()
    return o.numpy()


def checkpoint_digest(model):
    """sha256 of the package model's flat buffers (parameters in its KRSC storage layout, BN running statistics, step
    counters, momentum) + per-tensor norms of parameters and momentum in logical OIHW order."""
    import hashlib
    net, opt = model.net, model.optim
    h = lambda t: hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()   # noqa: E731
    norm = lambda ts: np.array([float(t.detach().double().norm()) for t in ts])                        # noqa: E731
    opt._ensure_buf()
    return dict(sha_params=h(net.flat_params), sha_running=h(net.flat_running), sha_nbt=h(net._nbt),
                sha_momentum=h(opt._buf), param_norms=norm([p for _, p in net._param_list]),
                momentum_norms=norm(opt._views), lr=np.float64(opt.param_groups[0]["lr"]))


def write_reference_layout_checkpoint(path, golden, sd, mom, lr, step):
    """A checkpoint file with EXACTLY the structure the reference's save_state produced when
    tests/golden/checkpoint_od.npz was made (state_dict key order / shapes / dtypes, torch.optim.SGD param-group keys,
    one momentum_buffer per parameter), filled with the seeded values."""
    from collections import OrderedDict
    state = OrderedDict()
    for k, shp, dt in zip(golden["keys"], golden["shapes"], golden["dtypes"]):
        v = torch.from_numpy(np.array(sd[str(k)]))
        assert ",".join(str(x) for x in v.shape) == str(shp) and str(v.dtype) == str(dt), k
        state[str(k)] = v
    defaults = dict(lr=lr, momentum=0.9, dampening=0, weight_decay=1e-4, nesterov=False, maximize=False, foreach=None,
                    differentiable=False, fused=None, initial_lr=1e-4)
    group = {str(k): defaults[str(k)] for k in golden["group_keys"] if str(k) in defaults}
    group["params"] = list(range(len(mom)))
    opt = {"state": {i: {"momentum_buffer": torch.from_numpy(b.copy())} for i, b in enumerate(mom)},
           "param_groups": [group]}
    torch.save({"step": step, "state_dict": state, "optimizer": opt}, path)


def prestepped_oracle_state(g, algo):
    """The well-conditioned state of tests/golden/backward_*.npz rebuilt with the CPU oracle: reference initialisation
    statistics + `pre` SGD steps of the reference recipe on the seeded batches (make_golden.py::case_backward ran the
    same steps with the real reference).  Returns (state, batch, exact): `exact` says whether the rebuild landed on the
    stored weights -- it does on the machine the golden was made on (the oracle is bit-identical to the reference
    there); another CPU takes a slightly different path through the ill-conditioned first steps and ends a few 1e-3
    away, in the same well-conditioned regime."""
    S, B, seed, pre = (int(v) for v in g["meta"])
    state = orc.state_from_numpy(synthetic.make_state_dict(seed, 5, ALGO_CLASSES[algo], style="xavier"))
    mom = {}
    for it in range(pre):
        orc.train_step(state, mom, synthetic.make_pair_batch(seed + 300 + it, B, S), algo, float(g["lr"]),
                       float(g["weight_decay"]))
    names = orc.param_names(state)
    pn, ps = norms_and_samples([state[n] for n in names])
    exact = rel_err(pn, g["pre_param_norms"]) < 1e-5 and rel_err(ps, g["pre_param_samples"]) < 1e-4
    assert rel_err(pn, g["pre_param_norms"]) < 5e-2, "the rebuilt state is nowhere near the reference's"
    return state, synthetic.make_pair_batch(seed + 100, B, S), exact


def oracle_gradients_fp32_fp64(state, batch, algo):
    """Gradients of one step from `state`: PyTorch-CPU fp32 (the reference's arithmetic) and fp64 (the anchor)."""
    st32 = {k: v.clone() for k, v in state.items()}
    st64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in state.items()}
    b64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in batch.items()}
    l32, g32 = orc.train_step(st32, {}, batch, algo, 0.0, 0.0)
    l64, g64 = orc.train_step(st64, {}, b64, algo, 0.0, 0.0)
    return l32, g32, l64, g64


def check_against_anchor(grads, g32, g64, label, factor=3.0, floor=5e-3):
    """Every parameter tensor, element-wise: ||g - g64|| / ||g64|| <= factor x (the same for PyTorch-CPU fp32) + floor.
    The floor is the price of ONE knife-edge ReLU decision: even in this well-conditioned regime (typical distance
    2e-6) a single pre-activation that two fp32 evaluations round to different sides of zero moves every gradient
    upstream of it by 1e-4 .. 2e-3 -- measured on both sides: PyTorch-CPU fp32 shows such an event on one host, the
    HIP path on another state (tools/bwd_probe.py).  A systematic 1 % error in any kernel is still twice the bound."""
    worst, bad = (0.0, 0.0, 0.0, ""), []
    for n in g64:
        ref = g64[n].double()
        den = float(ref.norm().clamp_min(1e-300))
        e = float((grads[n].double().cpu() - ref).norm()) / den
        ec = float((g32[n].double() - ref).norm()) / den
        if e > factor * ec + floor:
            bad.append((n, e, ec))
        if e / (ec + floor) > worst[0]:
            worst = (e / (ec + floor), e, ec, n)
    assert not bad, (label, sorted(bad, key=lambda t: -t[1])[:12])
    return worst


def bwd_subset(n, cap=1 << 16):
    return np.arange(n) if n <= cap else (np.arange(cap, dtype=np.int64) * n) // cap


def check_backward_golden(g, grads, label, factor=3.0, floor=5e-3):
    """grads: name -> gradient (any float dtype, OIHW).  Every tensor's norm against the fp64 anchor, the stored
    tensors element-wise (L2 over the stored subset), each held to `factor` x the REFERENCE's own fp32-vs-fp64 distance
    for that tensor (+ a floor of a few fp32 ulps of accumulated rounding).  Returns the worst ratios for printing."""
    names = [str(n) for n in g["names"]]
    worst = (0.0, 0.0, "")
    for n, n64, rd in zip(names, g["norms64"], g["ref_dist"]):
        e = abs(float(grads[n].double().norm()) - n64) / max(n64, 1e-300)
        assert e <= factor * rd + floor, (label, "norm", n, e, rd)
    for n in (str(t) for t in g["tensors"]):
        a = grads[n].detach().double().reshape(-1).cpu().numpy()
        ref = g["g64/" + n].astype(np.float64)
        sub = a[bwd_subset(a.size)]
        e = float(np.sqrt(((sub - ref) ** 2).sum()) / np.sqrt((ref ** 2).sum()))
        rd = float(g["refsub/" + n])
        assert e <= factor * rd + floor, (label, "elements", n, e, rd)
        if e / (rd + floor) > worst[0]:
            worst = (e / (rd + floor), e, n)
    return worst


from instaorder_amd import inference as _inference  # noqa: E402


# ---- the Tester's model-free methods in the CALLABLE form of evaluate.evaluate (the string form 'area' / 'yaxis' is
# dispatched inside the package; both are held to tests/golden/tester.npz)
infer_occ_order_area = _inference.infer_occ_order_area
infer_occ_order_yaxis = _inference.infer_occ_order_yaxis
infer_depth_order_area = _inference.infer_depth_order_area
infer_depth_order_yaxis = _inference.infer_depth_order_yaxis


def baseline_rule(kind, method):
    """The Tester's model-free methods as ``evaluate.evaluate`` callables (tools/test.py:306-318, 421-433)."""
    def rule(modal, dataset):
        lower = "lower" if dataset in ("COCOA", "InstaOrder") else "higher"
        if kind == "SupDepthOrderDataset":
            return infer_depth_order_area(modal, closer="larger") if method == "area" else \
                infer_depth_order_yaxis(modal, closer=lower)
        return infer_occ_order_area(modal, occluder="larger") if method == "area" else \
            infer_occ_order_yaxis(modal, occluder=lower)
    return rule
