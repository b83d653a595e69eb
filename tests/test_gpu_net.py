"""Whole-path parity on the MI355X: the HIP ResNet-50 + wrappers (through the C ABI) against
 (1) the CPU oracle on the same seeded inputs, (2) golden vectors produced by the real reference,
 (3) size-independent properties at larger batch.

Tolerances.  Forward quantities (logits, probabilities, losses, BN statistics): 1e-3 relative, the
north-star bar (measured ~1e-5..1e-4).  Gradients: fp32 ReLU networks are ill-conditioned in the
backward direction -- a pre-activation within 1e-6 of zero lands on different sides in two fp32
implementations and flips a whole gradient path; PyTorch-CPU fp32 itself differs from an fp64
evaluation of the same graph by ~2 % in gradient L2 norm (measured; see DESIGN.md "Backward
conditioning").  So backward parity is asserted two ways: tight (1e-3) on a network whose ReLUs never
switch (BN biases shifted positive), and statistically on the real network: distance to the fp64
oracle no larger than a small multiple of PyTorch-CPU-fp32's own distance to it."""
import copy

import numpy as np
import pytest
import torch

from helpers import (ALGO_CLASSES, ALGO_LR, bn_vectors, eval_logits_oracle, load_golden, norms_and_samples, orc,
                     oracle_state, rel_err, synthetic)

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-3


def cfg_for(algo):
    nc = ALGO_CLASSES[algo]
    cfg = dict(algo=algo, lr=ALGO_LR[algo], weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
               backbone_param=dict(in_channels=5, num_classes=nc), use_rgb=True, overlap_weight=0.1,
               distinct_weight=0.9, lr_steps=[32000, 48000], lr_mults=[0.1, 0.1], warmup_lr=[], warmup_steps=[])
    return cfg


def build(algo, seed, style="xavier"):
    import instaorder_amd as ia
    m = getattr(ia, algo)(cfg_for(algo), dist_model=False)
    sd = synthetic.make_state_dict(seed, 5, ALGO_CLASSES[algo], prefix="module.", style=style)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m


def set_input(m, algo, batch):
    t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
    if algo == "InstaOrderNet_od":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
    elif algo == "InstaOrderNet_d":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"])
    elif algo == "OrderNet":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"])
    else:
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["occ_order"])


def hip_eval_logits(m, batch):
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    x = torch.cat([t["modal1"], t["modal2"], t["rgb"]], 1).cuda()
    with torch.no_grad():
        o = m.model(x)
    return (torch.cat(o, 1) if isinstance(o, tuple) else o).cpu().numpy()


def hip_state(m):
    return {k[len("module."):]: v.detach().cpu() for k, v in m.model.state_dict().items()}


def unpack(ret):
    if isinstance(ret, tuple):
        out = {k: float(v) for k, v in ret[0].items()}
        out["loss"] = float(ret[1]["loss"])
        return out
    return {"loss": float(ret["loss"])}


# ------------------------------------------------------------------------------------------------
# 1. forward parity vs oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo,style,S,B", [("InstaOrderNet_o", "kaiming", 64, 4), ("InstaOrderNet_od", "kaiming", 96, 3),
                                            ("InstaOrderNet_o", "xavier", 128, 2), ("InstaOrderNet_o", "kaiming", 256, 2),
                                            # 384: the input_size of the reference's own _od configuration
                                            # (experiments/InstaOrder/InstaOrderNet_od/config.yaml:35); 12 x 12 at layer 4
                                            ("InstaOrderNet_od", "kaiming", 384, 2)])
def test_forward_train_and_eval_vs_oracle(algo, style, S, B):
    m = build(algo, 31, style)
    state = oracle_state(31, algo, style)
    batch = synthetic.make_pair_batch(500, B, S)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    x1 = torch.cat([tb["modal1"], tb["modal2"], tb["rgb"]], 1)
    # train-mode forward (batch statistics, running stats updated) -- several passes to warm the stats
    m.switch_to("train")
    for it in range(3):
        with torch.no_grad():
            zo = orc.resnet_forward(state, x1 * (1 + 0.1 * it), True)
            zh = m.model(x1.cuda() * (1 + 0.1 * it))
        zo = torch.cat(zo, 1) if isinstance(zo, tuple) else zo
        zh = torch.cat(zh, 1) if isinstance(zh, tuple) else zh
        assert rel_err(zh.cpu().numpy(), zo.numpy()) < FWD_TOL
    rm, rv, nb = bn_vectors(state)
    hrm, hrv, hnb = bn_vectors(hip_state(m))
    assert rel_err(hrm, rm) < FWD_TOL and rel_err(hrv, rv) < FWD_TOL and (hnb == nb).all()
    # eval-mode forward (running statistics)
    m.switch_to("eval")
    assert rel_err(hip_eval_logits(m, batch), eval_logits_oracle(state, batch)) < FWD_TOL


# ------------------------------------------------------------------------------------------------
# 2. backward parity
# ------------------------------------------------------------------------------------------------
def _shift_bn_bias(sd, shift):
    for k in sd:
        if (".bn" in k or k.startswith("bn1") or "downsample.1" in k) and k.endswith(".bias"):
            sd[k] = sd[k] + np.float32(shift)
    return sd


@pytest.mark.parametrize("algo,S,B", [("InstaOrderNet_o", 64, 4), ("InstaOrderNet_od", 64, 6)])
def test_backward_tight_on_relu_free_network(algo, S, B):
    """BN biases shifted to +8: every pre-activation is positive, no ReLU ever switches, so the whole
    backward wiring (dgrad/wgrad of all 53 convs, 53 BN backwards, residual sums, pooling, heads, loss)
    must agree with autograd to fp32 rounding."""
    import instaorder_amd as ia
    sd = _shift_bn_bias(synthetic.make_state_dict(41, 5, ALGO_CLASSES[algo], style="kaiming"), 8.0)
    m = getattr(ia, algo)(cfg_for(algo), dist_model=False)
    m.model.load_state_dict({"module." + k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    state = orc.state_from_numpy(sd)
    for k in state:
        if state[k].dtype == torch.float32:
            state[k] = state[k].double()
    batch = synthetic.make_pair_batch(600, B, S)
    b64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in batch.items()}
    logs, grads = orc.train_step(state, {}, b64, algo, 0.0, 0.0)
    _, g32 = orc.train_step(orc.state_from_numpy(sd), {}, batch, algo, 0.0, 0.0)   # PyTorch-CPU fp32, same graph
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    set_input(m, algo, batch)
    out = unpack(m.step())
    assert abs(out["loss"] - float(logs["loss"])) < 1e-4 * abs(float(logs["loss"]))
    names = orc.param_names(state)
    worst = worst_cpu = 0.0
    # BN biases that feed (through a linear conv) straight into another BN have an exactly-zero true
    # gradient here; measure those against the typical tensor scale instead of their own ~1e-17 norm
    floor = 1e-4 * float(np.median([float(grads[n].norm()) for n in names]))
    for n, p in zip(names, m.net.parameters()):
        g = p.grad.detach().cpu().double()
        ref = grads[n]
        den = max(float(ref.norm()), floor)
        e = float((g - ref).norm()) / den
        ec = float((g32[n].double() - ref).norm()) / den
        worst, worst_cpu = max(worst, e), max(worst_cpu, ec)
        # 1e-3 is the bar; the deepest tensors accumulate rounding through ~50 layers of large-mean
        # activations, where PyTorch-CPU fp32 itself sits near 1e-3, hence the relative clause
        assert e < max(1e-3, 3 * ec), (n, e, ec)
    print("relu-free backward: worst per-tensor rel L2 err vs fp64: HIP %.2e, torch-CPU-fp32 %.2e" % (worst, worst_cpu))


@pytest.mark.parametrize("algo,style,S,B", [("InstaOrderNet_o", "kaiming", 64, 8), ("InstaOrderNet_od", "xavier", 128, 4),
                                            ("InstaOrderNet_od", "kaiming", 384, 2),     # the reference _od input_size
                                            # the bench's network at the bench's input size with a mid-size batch: 64
                                            # samples per step, every layer on the fused whole-tile paths (32 k rows in
                                            # layer 1 .. 2 k in layer 4), ~10 s of oracle on the host in fp32 + fp64
                                            ("InstaOrderNet_o", "kaiming", 256, 32)])
def test_backward_statistical_vs_fp64_anchor(algo, style, S, B):
    sd = synthetic.make_state_dict(43, 5, ALGO_CLASSES[algo], style=style)
    st32 = orc.state_from_numpy(sd)
    st64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in orc.state_from_numpy(sd).items()}
    batch = synthetic.make_pair_batch(700, B, S)
    b64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in batch.items()}
    _, g32 = orc.train_step(st32, {}, batch, algo, 0.0, 0.0)
    _, g64 = orc.train_step(st64, {}, b64, algo, 0.0, 0.0)
    m = build(algo, 43, style)
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    set_input(m, algo, batch)
    m.step()
    names = orc.param_names(st32)
    num_h = num_c = den = 0.0
    for n, p in zip(names, m.net.parameters()):
        ref = g64[n]
        gh = p.grad.detach().cpu().double()
        eh = float((gh - ref).norm() / ref.norm().clamp_min(1e-30))
        ec = float((g32[n].double() - ref).norm() / ref.norm().clamp_min(1e-30))
        num_h += float((gh - ref).norm() ** 2)
        num_c += float((g32[n].double() - ref).norm() ** 2)
        den += float(ref.norm() ** 2)
        # no tensor may be grossly off: well inside what ReLU flips explain
        assert eh < max(6 * ec, 0.08), (n, eh, ec)
    eh, ec = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    print("global grad rel err vs fp64: HIP %.3e, torch-CPU-fp32 %.3e" % (eh, ec))
    assert eh < 3 * ec + 1e-3


# ------------------------------------------------------------------------------------------------
# 3. golden vectors from the real reference
# ------------------------------------------------------------------------------------------------
GOLD = [("o_S64_B4", "InstaOrderNet_o", "xavier"), ("od_S64_B6", "InstaOrderNet_od", "xavier"),
        ("d_S64_B6", "InstaOrderNet_d", "xavier"), ("ordernet_S64_B4", "OrderNet", "xavier"),
        ("o_S64_B4_k", "InstaOrderNet_o", "kaiming"), ("od_S64_B6_k", "InstaOrderNet_od", "kaiming"),
        ("o_S256_B4", "InstaOrderNet_o", "xavier"), ("od_S256_B4", "InstaOrderNet_od", "xavier"),
        # well-scaled states (eval logits O(0.1 .. 1)): the assertions below are informative on every quantity
        ("o_S256_B4_k", "InstaOrderNet_o", "kaiming"), ("od_S256_B4_k", "InstaOrderNet_od", "kaiming"),
        ("od_S384_B2_k", "InstaOrderNet_od", "kaiming")]


@pytest.mark.parametrize("tag,algo,style", GOLD)
def test_golden_first_step(tag, algo, style):
    """Everything the reference produced for the FIRST optimisation step from identical weights:
    eval logits / loss, step loss (+ components), gradient norms, post-step weights, BN running stats.
    (Later steps of these tiny-batch cases are chaotic -- a 1e-7 weight perturbation changes step-1
    gradients by tens of percent in the reference itself -- so only step 0 is a parity target.)"""
    g = load_golden(tag)
    S, B, seed, steps = [int(v) for v in g["meta"]]
    m = build(algo, seed, style)
    b0 = synthetic.make_pair_batch(seed + 100, B, S)
    m.switch_to("eval")
    scale = max(np.abs(g["eval0_logits"]).max(), 1e-6)
    # (absolute slack only for the xavier-gain-0.02 states, whose eval logits are ~1e-12; none where they are O(0.1))
    slack = 1e-7 if scale < 1e-3 else 0.0
    assert np.abs(hip_eval_logits(m, b0) - g["eval0_logits"]).max() < FWD_TOL * scale + slack
    set_input(m, algo, b0)
    ev = m.forward_only()
    assert abs(float(ev[1]["loss"]) - float(g["eval0_loss"])) < FWD_TOL * float(g["eval0_loss"])
    m.switch_to("train")
    set_input(m, algo, synthetic.make_pair_batch(seed + 100, B, S))
    out = unpack(m.step())
    for k, v in out.items():
        ref = float(g["step0_" + k])
        assert abs(v - ref) < FWD_TOL * abs(ref), (k, v, ref)
    params = list(m.net.parameters())
    gn, _ = norms_and_samples([p.grad for p in params])
    gerr = np.abs(gn - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-30)
    print(tag, "grad-norm rel diff: median %.2e max %.2e" % (np.median(gerr), gerr.max()))
    assert np.median(gerr) < 0.02 and gerr.max() < 0.15
    # post-step weights: p - lr*(g + wd*p).  The weights themselves agree to ~1e-6; what can differ is
    # lr x (gradient difference), and gradients carry the ReLU-flip conditioning discussed above, so the
    # sampled elements are held to lr x 15 % of the tensor's typical gradient element (+ fp32 rounding).
    pn, ps = norms_and_samples(params)
    assert rel_err(pn, g["step0_param_norms"]) < 1e-4
    lr = float(g["lr"])
    numel = np.array([p.numel() for p in params], np.float64)
    dps = np.sqrt(((ps.astype(np.float64) - g["step0_param_samples"]) ** 2).sum(1))
    bound = lr * 0.15 * g["grad_norms"] * np.sqrt(64.0 / numel) + 1e-6 * np.abs(g["step0_param_samples"]).max(1) * 8
    assert (dps <= bound).all(), (np.argmax(dps / bound), (dps / bound).max())
    rm, rv, nb = bn_vectors(hip_state(m))
    assert rel_err(rm, g["step0_running_mean"]) < FWD_TOL and rel_err(rv, g["step0_running_var"]) < FWD_TOL
    assert (nb == g["step0_num_batches"]).all()


@pytest.mark.parametrize("tag,algo", [("backward_o_S64_B16", "InstaOrderNet_o"), ("backward_od_S128_B16", "InstaOrderNet_od")])
def test_golden_backward_real_relu_tight(tag, algo):
    """Backward parity on the real network, per tensor, tight.  Ten SGD steps of the reference recipe away from its
    initialisation the reference's own fp32 gradients agree with an fp64 evaluation to ~1e-6 (1-2 % at a random
    initialisation, whatever the batch -- which is why the first-step goldens above can only be held to percent
    level).  The golden holds the fp64 gradients the REFERENCE computed there (make_golden.py::case_backward); the HIP
    path, started from the oracle's rebuild of that state, is held to 3x the reference's own fp32-vs-fp64 distance per
    tensor + 5e-3 (one knife-edge ReLU decision moves everything upstream of it by up to ~2e-3, see
    helpers.check_against_anchor; typical distances are printed: ~2e-6 on both sides), every tensor element-wise (the fp64 anchor is evaluated by the oracle on this host from the same
    rebuilt state; the stored reference vectors are compared too when the rebuild is exact).  A 1 % systematic backward
    error in any kernel fails it.  S=64: the last stage has 64 rows per BatchNorm group -- the unfused statistics paths;
    S=128: every layer runs the fused ones."""
    from helpers import (check_against_anchor, check_backward_golden, oracle_gradients_fp32_fp64,
                         prestepped_oracle_state)
    g = load_golden(tag)
    state, batch, exact = prestepped_oracle_state(g, algo)
    l32, g32, l64, g64 = oracle_gradients_fp32_fp64(state, batch, algo)      # the anchor for THIS state, on this host
    import instaorder_amd as ia
    m = getattr(ia, algo)(cfg_for(algo), dist_model=False)
    m.model.load_state_dict({"module." + k: v.clone() for k, v in state.items()}, strict=True)
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    set_input(m, algo, batch)
    out = unpack(m.step())
    assert abs(out["loss"] - float(l64["loss"])) < 1e-5 * abs(float(l64["loss"]))
    names = orc.param_names(state)
    grads = {n: p.grad.detach().cpu() for n, p in zip(names, m.net.parameters())}
    worst = check_against_anchor(grads, g32, g64, "hip")
    dist = np.array([float((g32[n].double() - g64[n]).norm() / g64[n].norm().clamp_min(1e-300)) for n in g64])
    print(tag, "PyTorch-CPU fp32 vs fp64 here: median %.2e max %.2e (reference when the golden was made: %.2e / %.2e)"
          % (np.median(dist), dist.max(), np.median(g["ref_dist"]), g["ref_dist"].max()))
    eh = np.array([float((grads[n].double() - g64[n]).norm() / g64[n].norm().clamp_min(1e-300)) for n in g64])
    print(tag, "HIP vs fp64: median %.2e max %.2e (worst tensor %s: %.2e, PyTorch-CPU fp32 there %.2e)"
          % (np.median(eh), eh.max(), worst[3], worst[1], worst[2]))
    assert np.median(dist) < 1e-4            # the well-conditioned regime, not the 1-2 % one of a random initialisation
    if exact:                                 # same CPU arithmetic as the build container: also the stored reference vectors
        check_backward_golden(g, grads, "hip-vs-golden")


def test_trajectory_in_the_well_conditioned_regime():
    """Four consecutive optimisation steps -- forward, loss, backward, momentum SGD with weight decay, BatchNorm running
    statistics -- of the HIP path against the CPU oracle, from the well-conditioned state of the backward golden and on
    fresh batches.  Every step's loss terms within 2e-4 of the oracle's.  The weights: a trajectory amplifies rounding
    (each step's gradient depends on the previous steps' weights), so the yardstick is the fp32 arithmetic's own
    sensitivity -- the same steps by the oracle in fp64 are the anchor, and per tensor the HIP path's distance from
    it, measured against how far the tensor MOVED, is held to 3x the distance of the oracle's own fp32 run + 2 % of the
    movement (a systematic error in the optimiser, the weight decay, the momentum or a gradient kernel is of order 1)."""
    from helpers import prestepped_oracle_state
    algo = "InstaOrderNet_od"
    g = load_golden("backward_od_S128_B16")
    S, B, seed, pre = (int(v) for v in g["meta"])
    state, batch0, _ = prestepped_oracle_state(g, algo)
    import instaorder_amd as ia
    cfg = cfg_for(algo)
    cfg["lr"], cfg["weight_decay"] = float(g["lr"]), float(g["weight_decay"])
    m = getattr(ia, algo)(cfg, dist_model=False)
    m.model.load_state_dict({"module." + k: v.clone() for k, v in state.items()}, strict=True)
    m.switch_to("train")
    start = {k: v.clone() for k, v in state.items()}
    st64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in state.items()}
    mom, mom64 = {}, {}
    for it in range(4):
        batch = synthetic.make_pair_batch(seed + 500 + it, B, S)
        b64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in batch.items()}
        set_input(m, algo, batch)
        out = unpack(m.step())
        logs, _ = orc.train_step(state, mom, batch, algo, cfg["lr"], cfg["weight_decay"])
        orc.train_step(st64, mom64, b64, algo, cfg["lr"], cfg["weight_decay"])
        for k in ("loss", "loss_occ", "loss_depth"):
            assert abs(out[k] - float(logs[k])) < 2e-4 * abs(float(logs[k])), (it, k, out[k], float(logs[k]))
    hs = hip_state(m)
    bad, ratios = [], []
    for k, v in st64.items():
        if v.dtype != torch.float64:
            assert int(hs[k]) == int(v), k
            continue
        ref = v
        mv = float((ref - start[k].double()).norm())
        if mv == 0.0:
            continue
        eh = float((hs[k].double() - ref).norm()) / mv
        ec = float((state[k].double() - ref).norm()) / mv
        ratios.append((eh, ec))
        if eh > 3 * ec + 2e-2:
            bad.append((k, eh, ec))
    # (a knife-edge ReLU decision inside the steps moves single tensors: over five steps layer3.3.conv2.weight sat at 9 % of its
    # movement where the oracle's own fp32 run is at 1.5 % -- its five siblings in the stage are inside the bound, which
    # a kernel error could not arrange; so: at most two of the 163 tensors outside, none beyond 20 %)
    assert len(bad) <= 2 and all(e < 0.2 for _, e, _ in bad), sorted(bad, key=lambda t: -t[1])[:6]
    r = np.asarray(ratios)
    print("trajectory, distance from the fp64 run / movement: HIP median %.2e max %.2e; oracle fp32 median %.2e max %.2e"
          % (np.median(r[:, 0]), r[:, 0].max(), np.median(r[:, 1]), r[:, 1].max()))


@pytest.mark.parametrize("tag,algo", [("plumbing_o", "InstaOrderNet_o"), ("plumbing_od", "InstaOrderNet_od")])
def test_golden_plumbing(tag, algo):
    """config 1: synthetic 256x256 images x instances through the batched O(n^2) pair driver; order
    matrices and metrics equal the reference's batch-1 Python loop."""
    from instaorder_amd import inference as infer
    g = load_golden(tag)
    S, n_images, n_inst, seed, warm = [int(v) for v in g["meta"]]
    m = build(algo, seed, "kaiming")
    m.switch_to("train")
    for it in range(warm):
        b = synthetic.make_pair_batch(seed + 300 + it, 8, S)
        with torch.no_grad():
            m.model(torch.cat([torch.from_numpy(b["modal1"]), torch.from_numpy(b["modal2"]),
                               torch.from_numpy(b["rgb"])], 1).cuda())
    hb = torch.from_numpy(g["head_bias"])
    with torch.no_grad():
        if algo == "InstaOrderNet_o":
            m.net.fc.bias.copy_(hb)
        else:
            m.net.fc_occ.bias.copy_(hb[:2])
            m.net.fc_depth.bias.copy_(hb[2:])
    m.switch_to("eval")
    items = synthetic.make_images(seed + 400, n_images, n_inst, S)
    for ii, item in enumerate(items):
        rgb, masks = synthetic.image_mode_inputs(item["image"], item["modal"], S)
        res = infer.infer_order_batched(m, torch.from_numpy(rgb), torch.from_numpy(masks), method=algo,
                                        return_logits=True)
        gl = g["pair_logits_%d" % ii]
        assert np.abs(res["pair_logits"] - gl).max() < FWD_TOL * np.abs(gl).max() + 1e-6
        # decisions must match wherever the reference's own margin exceeds the fp32 noise floor
        ref_occ = g["occ_%d" % ii]
        margin = infer.decision_margins(torch.from_numpy(gl), algo)
        safe = margin["occ"] > 1e-5
        got = res["occ_order"]
        pairs = res["pairs"]
        for k, (i, j) in enumerate(pairs):
            if safe[k, 0]:
                assert got[i, j] == ref_occ[i, j]
            if safe[k, 1]:
                assert got[j, i] == ref_occ[j, i]
        if bool(safe.all()):
            assert (got == ref_occ).all()
            prf = infer.eval_order_recall_precision_f1(got, item["gt_occ"], 0)
            assert np.allclose(prf, g["prf_%d" % ii], atol=1e-9)
        if algo == "InstaOrderNet_od" and bool((margin["depth"] > 1e-5).all()):
            assert (res["depth_order"] == g["depth_%d" % ii]).all()
            w = infer.eval_depth_order_whdr(res["depth_order"], (item["gt_depth"], item["gt_overlap"], item["gt_count"]))
            keys = [str(k) for k in g["whdr_keys"]]
            assert np.allclose([w[k][0] for k in keys], g["whdr_%d" % ii], atol=1e-9)


# ------------------------------------------------------------------------------------------------
# 4. size-independent properties
# ------------------------------------------------------------------------------------------------
def test_properties_larger_batch():
    algo = "InstaOrderNet_o"
    m = build(algo, 51, "kaiming")
    B, S = 32, 128
    batch = synthetic.make_pair_batch(900, B, S)
    # (a) two BN groups in one launch == two independent module calls (train mode)
    m.switch_to("train")
    st0 = copy.deepcopy({k: v.clone() for k, v in m.model.state_dict().items()})
    t = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    x1 = torch.cat([t["modal1"], t["modal2"], t["rgb"]], 1)
    x2 = torch.cat([t["modal2"], t["modal1"], t["rgb"]], 1)
    with torch.no_grad():
        z1, z2 = m.model(x1), m.model(x2)
    seq_state = {k: v.clone() for k, v in m.model.state_dict().items()}
    m.model.load_state_dict(st0)
    set_input(m, algo, batch)
    m.forward_only()
    zz = m.last_logits
    assert rel_err(zz[:B].cpu().numpy(), z1.cpu().numpy()) < 1e-4
    assert rel_err(zz[B:].cpu().numpy(), z2.cpu().numpy()) < 1e-4
    for k, v in m.model.state_dict().items():
        if v.dtype == torch.float32:
            assert rel_err(v.cpu().numpy(), seq_state[k].cpu().numpy()) < 1e-5, k
        else:
            assert torch.equal(v, seq_state[k]), k
    # (b) eval: a batch is the concatenation of its halves (no cross-sample coupling)
    m.switch_to("eval")
    with torch.no_grad():
        full = m.model(x1)
        halves = torch.cat([m.model(x1[:B // 2]), m.model(x1[B // 2:])], 0)
    assert rel_err(full.cpu().numpy(), halves.cpu().numpy()) < 1e-5
    # (c) backward is linear in the logit gradient: doubling it doubles every gradient bit-exactly
    m.switch_to("train")
    net = m.net
    from instaorder_amd import engine
    x8 = engine.pack_pair_directions(t["rgb"], t["modal1"], t["modal2"])
    logits, ws = net._run_forward(x8, 2 * B, S, 2, True)
    dl = torch.randn_like(logits) * 0.01
    net._run_backward(x8, dl, 2 * B, S, 2, ws)
    g1 = net.flat_grads.clone()
    net._run_backward(x8, 2 * dl, 2 * B, S, 2, ws)
    assert torch.equal(net.flat_grads, 2 * g1)
    # (d) determinism: same launch twice -> identical bits
    net._run_backward(x8, dl, 2 * B, S, 2, ws)
    assert torch.equal(net.flat_grads, g1)
    # (e) the backward cut into its stages (io_net_backward_stages: what the data-parallel step interleaves with the
    # bucketed gradient all-reduce) IS the backward: stage by stage, or in two halves, bit for bit -- and after stage s
    # the slice of the flat gradient buffer that grad_stage_slices() names for it is already final
    ns = net.plan.backward_stages
    sl = net.grad_stage_slices()
    assert ns == 4 and len(sl) == ns
    real = torch.zeros(net.flat_grads.numel(), dtype=torch.bool, device="cuda")      # (tensors are padded to 256 bytes)
    for t_ in net.plan.tensors:
        real[t_["offset"]:t_["offset"] + t_["numel_storage"]] = True
    mark = 12345.0
    net.flat_grads.fill_(mark)
    for s_ in range(ns):
        net._run_backward(x8, dl, 2 * B, S, 2, ws, stages=(s_, s_ + 1))
        lo, hi = sl[s_]
        assert torch.equal(net.flat_grads[lo:hi][real[lo:hi]], g1[lo:hi][real[lo:hi]]), s_
        if s_ + 1 < ns:
            later = sl[s_ + 1][1]
            assert bool((net.flat_grads[:later] == mark).all())                       # nothing of the later stages yet
    assert torch.equal(net.flat_grads[real], g1[real])
    net.flat_grads.fill_(mark)
    net._run_backward(x8, dl, 2 * B, S, 2, ws, stages=(0, 2))
    net._run_backward(x8, dl, 2 * B, S, 2, ws, stages=(2, 4))
    assert torch.equal(net.flat_grads[real], g1[real])
    net._pool.give(ws)


def test_generic_autograd_path_is_wired_like_the_fused_step():
    """The STRICT wiring check between the generic autograd path (model(x) in train mode + loss.backward() through the
    reference's own loss expression) and the fused step(): with the Winograd row forms switched off (engine.set_winograd)
    both paths run the direct kernels, whose sums do not depend on the rows per launch -- every gradient tensor, BN betas
    and the fc bias included, agrees to 1e-4 of its own norm.  (With the forms on the two paths take different product
    forms per layer: the conditioning-aware bar of the test below.)"""
    from instaorder_amd import engine
    algo = "InstaOrderNet_od"
    B, S = 4, 64
    batch = synthetic.make_pair_batch(951, B, S)
    prev = engine.set_winograd(False)
    try:
        m = build(algo, 54, "kaiming")
        m.switch_to("train")
        m.optim.param_groups[0]["lr"] = 0.0
        set_input(m, algo, batch)
        m.step()
        m2 = build(algo, 54, "kaiming")
        m2.switch_to("train")
        t = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
        F = torch.nn.functional
        tot = 0
        for ma, mb, flip in ((t["modal1"], t["modal2"], False), (t["modal2"], t["modal1"], True)):
            occ, dep = m2.model(torch.cat([ma, mb, t["rgb"]], 1))
            y = t["occ_order"][:, [1, 0]] if flip else t["occ_order"]
            d = t["depth_order"]
            if flip:
                d = torch.where(d == 2, d, 1 - d)
            pd = torch.softmax(dep, 1)
            ov = t["is_overlap"] == 1
            tot = tot + F.binary_cross_entropy(torch.sigmoid(occ), y)
            # supervised_order.py:62-73: CE on the probabilities, per overlap subset, weights 0.1 / 0.9
            if bool(ov.any()):
                tot = tot + 0.1 * F.cross_entropy(pd[ov], d[ov])
            di = t["is_overlap"] == 0
            if bool(di.any()):
                tot = tot + 0.9 * F.cross_entropy(pd[di], d[di])
        tot.backward()
    finally:
        engine.set_winograd(prev)
    worst = (0.0, None)
    for (tinfo, p), gv in zip(m2.net._param_list, m.net._grad_views):
        a, b = p.grad.double().cpu().numpy(), gv.double().cpu().numpy()
        assert np.sqrt((b * b).sum()) > 0, tinfo
        e = np.sqrt(((a - b) ** 2).sum()) / np.sqrt((b * b).sum())
        if e > worst[0]:
            worst = (e, tinfo)
    print("generic vs fused, direct kernels: worst per-tensor relative L2 %.2e" % worst[0])
    assert worst[0] < 1e-4, worst


def test_generic_autograd_path_matches_fused_step():
    """model(x) in train mode carries autograd history: loss.backward() through the reference's own
    loss expression gives the same gradients as the fused step() -- here with the Winograd forms ON (the extra check;
    the strict one is test_generic_autograd_path_is_wired_like_the_fused_step)."""
    algo = "InstaOrderNet_o"
    B, S = 4, 64
    batch = synthetic.make_pair_batch(950, B, S)
    m = build(algo, 53, "kaiming")
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    set_input(m, algo, batch)
    m.step()
    g_fused = m.net.flat_grads.clone()
    m2 = build(algo, 53, "kaiming")
    m2.switch_to("train")
    t = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    o1 = torch.sigmoid(m2.model(torch.cat([t["modal1"], t["modal2"], t["rgb"]], 1)))
    o2 = torch.sigmoid(m2.model(torch.cat([t["modal2"], t["modal1"], t["rgb"]], 1)))
    y1 = t["occ_order"]
    loss = torch.nn.functional.binary_cross_entropy(o1, y1) + torch.nn.functional.binary_cross_entropy(o2, y1[:, [1, 0]])
    loss.backward()
    # The two paths do not run the same arithmetic any more: the fused step sees 2B samples per launch, the generic path B,
    # and which product form a 3x3 layer takes (Winograd F(4,3) / F(2,3) / direct) depends on the rows per launch -- the
    # results differ at fp32 rounding level, and a random-weight ReLU net turns that into ~1 % of a gradient tensor
    # (finding 1: PyTorch-CPU fp32 itself sits 2 % from an fp64 evaluation).  A wiring error -- a
    # missing term, a wrong scale -- would be tens of percent everywhere: the bar of the first-step goldens (per-tensor norms
    # median 2 % / max 15 %) + 5e-2 in L2 over everything.  (At B = 4, S = 64 single elements of the deep 4 x 4 layers move
    # by 10 % of their tensor's maximum on one flipped ReLU.)
    num = den = 0.0
    nerr = []
    for (tinfo, p), gv in zip(m2.net._param_list, m.net._grad_views):
        a, b = p.grad.double().cpu().numpy(), gv.double().cpu().numpy()
        nerr.append(abs(np.sqrt((a * a).sum()) - np.sqrt((b * b).sum())) / max(np.sqrt((b * b).sum()), 1e-30))
        num += float(((a - b) ** 2).sum())
        den += float((b ** 2).sum())
    assert np.median(nerr) < 0.02 and max(nerr) < 0.15, (float(np.median(nerr)), float(max(nerr)))
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5
    assert g_fused.abs().sum() > 0


def test_checkpoint_roundtrip_and_reference_layout(tmp_path):
    algo = "InstaOrderNet_od"
    m = build(algo, 61, "kaiming")
    m.switch_to("train")
    set_input(m, algo, synthetic.make_pair_batch(960, 4, 64))
    m.step()
    m.save_state(str(tmp_path), 7)
    ck = torch.load(str(tmp_path / "ckpt_iter_7.pth.tar"), map_location="cpu", weights_only=False)
    assert ck["step"] == 7 and set(ck) == {"step", "state_dict", "optimizer"}
    specs = synthetic.state_specs(5, [2, 3])
    assert list(ck["state_dict"].keys()) == ["module." + n for n, _, _ in specs]
    for (n, shape, _), v in zip(specs, ck["state_dict"].values()):
        assert tuple(v.shape) == tuple(shape) and v.is_contiguous()
    assert len(ck["optimizer"]["state"]) == 163 and len(ck["optimizer"]["param_groups"]) == 1
    m2 = build(algo, 62, "xavier")
    m2.load_state(str(tmp_path), 7, resume=True)
    for (k, a), b in zip(m.model.state_dict().items(), m2.model.state_dict().values()):
        assert torch.equal(a, b), k
    # resumed optimiser continues identically
    for mm in (m, m2):
        mm.switch_to("train")
        set_input(mm, algo, synthetic.make_pair_batch(961, 4, 64))
        mm.step()
    assert torch.equal(m.net.flat_params, m2.net.flat_params)


def test_nccl_single_rank_dp_path():
    """The data-parallel wrapper path (DistModule broadcast, flat RCCL all-reduce, loss/world) on one
    rank: same numbers as the non-distributed path."""
    import os
    import torch.distributed as dist
    from instaorder_amd import distributed_utils as du
    import instaorder_amd as ia
    algo = "InstaOrderNet_o"
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    du.dist_init_("pytorch", backend="nccl")
    try:
        batch = synthetic.make_pair_batch(970, 4, 64)
        res = []
        for dist_model in (True, False):
            m = getattr(ia, algo)(cfg_for(algo), dist_model=dist_model)
            sd = synthetic.make_state_dict(71, 5, 2, prefix="module.", style="kaiming")
            m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
            assert m.world_size == 1
            m.switch_to("train")
            set_input(m, algo, batch)
            out = m.step()
            tot = du.reduce_tensors(out["loss"])
            res.append((float(tot), m.net.flat_params.clone()))
        assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    finally:
        dist.destroy_process_group()


def _nccl_one_rank():
    import os
    import socket
    from instaorder_amd import distributed_utils as du
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    du.dist_init_("pytorch", backend="nccl")


def test_nccl_staged_overlap_path_on_one_rank(monkeypatch):
    """The data-parallel step as N > 1 ranks run it -- forward + loss + backward stage 0 in one hipGraph, stages 1..3 in
    one each, ProcessGroupNCCL's (= RCCL's) asynchronous all-reduce of every stage bucket launched between the replays
    (supervised_order._step_overlapped, distributed_utils.GradientBuckets; reference: utils/distributed_utils.py:27-37,
    main.py:35) -- driven on ONE rank (IO_COMM_OVERLAP=force): eager step, capture, two replays.  A SUM over one rank is
    the identity, so losses and weights must equal the flat path (IO_COMM_OVERLAP=0: one graph, one all-reduce after the
    backward) bit for bit; what this pins is RCCL's stream hand-over around per-stage graph replays, which no gloo run
    sees.  The eager staged form (IO_NO_GRAPH=1) is held to the same."""
    import torch.distributed as dist
    import instaorder_amd as ia
    algo = "InstaOrderNet_od"
    _nccl_one_rank()
    try:
        batches = [synthetic.make_pair_batch(975 + i, 6, 64) for i in range(4)]
        res = {}
        for tag, env in (("staged", {"IO_COMM_OVERLAP": "force"}), ("flat", {"IO_COMM_OVERLAP": "0"}),
                         ("staged_eager", {"IO_COMM_OVERLAP": "force", "IO_NO_GRAPH": "1"})):
            for k in ("IO_COMM_OVERLAP", "IO_NO_GRAPH"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            m = getattr(ia, algo)(cfg_for(algo), dist_model=True)
            sd = synthetic.make_state_dict(72, 5, [2, 3], prefix="module.", style="kaiming")
            m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
            m.switch_to("train")
            losses = []
            for b in batches:           # the wrapper keys its graphs on the input buffer: same shapes -> step 2 captures
                set_input(m, algo, b)
                losses.append(float(m.step()[1]["loss"]))
            torch.cuda.synchronize()
            res[tag] = (losses, m.net.flat_params.clone(), None, bool(m._dp_graphs), bool(m._graph))
        assert res["staged"][3] and not res["staged"][4], "the staged run must have replayed per-stage graphs"
        assert res["flat"][4] and not res["flat"][3]
        assert not res["staged_eager"][3] and not res["staged_eager"][4]
        for tag in ("staged", "staged_eager"):
            assert res[tag][0] == res["flat"][0], (tag, res[tag][0], res["flat"][0])
            assert torch.equal(res[tag][1], res["flat"][1]), tag
    finally:
        dist.destroy_process_group()


def test_full_size_properties_config2():
    """BASELINE configs[1] size (256 pairs at 256x256, 512 samples per step): the oracle cannot run this in
    seconds, so parity rests on size-independent properties -- (1) the two-group step equals two independent
    module calls, (2) bit-exact determinism and linearity of the backward, (3) the first 4 pairs embedded in
    the 256-pair EVAL batch give the oracle's logits for those 4 pairs (eval has no cross-sample coupling)."""
    import instaorder_amd as ia
    from instaorder_amd import engine
    algo, B, S = "InstaOrderNet_o", 256, 256
    m = build(algo, 81, "kaiming")
    base = synthetic.make_pair_batch(990, 8, S)
    reps = B // 8
    dev = {k: torch.from_numpy(np.concatenate([v] * reps, 0)).cuda() for k, v in base.items()}
    dev["rgb"] = dev["rgb"] + 0.01 * torch.arange(B, device="cuda", dtype=torch.float32).view(B, 1, 1, 1)
    net = m.net
    m.switch_to("train")
    st0 = {k: v.clone() for k, v in m.model.state_dict().items()}
    x8 = engine.pack_pair_directions(dev["rgb"], dev["modal1"], dev["modal2"])
    assert x8.shape == (2 * B, S, S, 8)
    logits, ws = net._run_forward(x8, 2 * B, S, 2, True)
    two = {k: v.clone() for k, v in m.model.state_dict().items()}
    dl = torch.randn_like(logits) * 1e-3
    net._run_backward(x8, dl, 2 * B, S, 2, ws)
    g1 = net.flat_grads.clone()
    assert torch.isfinite(g1).all() and float(g1.abs().sum()) > 0
    net._run_backward(x8, dl, 2 * B, S, 2, ws)
    assert torch.equal(net.flat_grads, g1)                       # deterministic
    net._run_backward(x8, 4 * dl, 2 * B, S, 2, ws)
    assert torch.equal(net.flat_grads, 4 * g1)                   # linear (power-of-two scaling is exact)
    net._pool.give(ws)
    del ws
    # (1) two sequential single-group calls from the same starting state
    m.model.load_state_dict(st0)
    with torch.no_grad():
        za = net.forward_packed(x8[:B], 1)
        zb = net.forward_packed(x8[B:], 1)
    assert rel_err(logits[:B].cpu().numpy(), za.cpu().numpy()) < 1e-4
    assert rel_err(logits[B:].cpu().numpy(), zb.cpu().numpy()) < 1e-4
    for k, v in m.model.state_dict().items():
        if v.dtype == torch.float32:
            assert rel_err(v.cpu().numpy(), two[k].cpu().numpy()) < 1e-5, k
        else:
            assert torch.equal(v, two[k]), k
    # (3) eval: 4 of the 256 pairs against the CPU oracle
    m.switch_to("eval")
    state = orc.state_from_numpy({k[len("module."):]: v.cpu().numpy() for k, v in m.model.state_dict().items()})
    with torch.no_grad():
        ze = net.forward_packed(x8[:B], 1)[:4].cpu().numpy()
        xo = torch.cat([dev["modal1"][:4], dev["modal2"][:4], dev["rgb"][:4]], 1).cpu()
        zo = orc.resnet_forward(state, xo, False).numpy()
    assert rel_err(ze, zo) < FWD_TOL


def test_use_rgb_false_and_ordernet_ext():
    """The reference's mask-only form (use_rgb False, in_channels 2) and the 4-class OrderNet_ext head."""
    import instaorder_amd as ia
    B, S = 4, 64
    batch = synthetic.make_pair_batch(975, B, S)
    cfg = cfg_for("InstaOrderNet_o")
    cfg["use_rgb"] = False
    cfg["backbone_param"] = dict(in_channels=2, num_classes=2)
    m = ia.InstaOrderNet_o(cfg, dist_model=False)
    sd = synthetic.make_state_dict(73, 2, 2, prefix="module.", style="kaiming")
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    state = orc.state_from_numpy(sd, prefix="module.")
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    set_input(m, "InstaOrderNet_o", batch)
    out = m.step()
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    x1, x2 = torch.cat([tb["modal1"], tb["modal2"]], 1), torch.cat([tb["modal2"], tb["modal1"]], 1)
    with torch.no_grad():
        o1 = torch.sigmoid(orc.resnet_forward(state, x1, True))
        o2 = torch.sigmoid(orc.resnet_forward(state, x2, True))
        ref = (torch.nn.functional.binary_cross_entropy(o1, tb["occ_order"])
               + torch.nn.functional.binary_cross_entropy(o2, tb["occ_order"][:, [1, 0]]))
    assert abs(float(out["loss"]) - float(ref)) < FWD_TOL * float(ref)
    cfg = cfg_for("OrderNet")
    cfg["backbone_param"] = dict(in_channels=5, num_classes=4)
    m = ia.OrderNet(cfg, dist_model=False)
    sd = synthetic.make_state_dict(74, 5, 4, prefix="module.", style="kaiming")
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    state = orc.state_from_numpy(sd, prefix="module.")
    lab = torch.tensor([0, 3, 1, 2])
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    m.set_input(tb["rgb"], tb["modal1"], tb["modal2"], lab)
    out = m.step()
    b2 = dict(batch)
    b2["depth_order"] = lab.numpy()
    logs, _ = orc.train_step(state, {}, b2, "OrderNet", 0.0, 0.0)
    assert abs(float(out["loss"]) - float(logs["loss"])) < FWD_TOL * float(logs["loss"])


def test_synthetic_val_accuracy_within_0p1_pp():
    """'val order accuracy within 0.1 pp': occlusion recall/precision/F1 and depth WHDR of the HIP path vs the
    reference-pinned oracle on a seeded synthetic validation set (no dataset / checkpoint exists offline)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("synthetic_val", os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "synthetic_val.py"))
    sv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sv)
    # SURVEY.md 8(d) "Accuracy": >= 200 synthetic images x 3-20 instances, at the bench's input size.  (The full
    # 200 x 5-instance run -- 4000 network evaluations, 0 differing decisions -- is profiles/r03_synthetic_val.txt.)
    delta, flips = sv.run(n_images=200, n_inst=3, S=256, verbose=True)
    for k, v in delta.items():
        assert abs(v) <= 0.1, (k, v)


def test_trained_model_accuracy_fp32_and_bf16():
    """The same claim on a TRAINED network, whose decisions sit away from the thresholds (tools/trained_val.py: 400
    iterations of the synthetic training chain, then 40 unseen 6-instance scenes through the 'patch' pre-processing; the
    very same network inputs go to the HIP path in fp32, to the CPU oracle on the same weights, and to the HIP path in
    bf16).  fp32: every decision is the oracle's -- 0.000 pp.  bf16 (not a reference precision): a decision may flip only
    where the oracle's own margin is inside the bf16 noise (< 2e-3 in probability units); on this set that is ~1 decision
    in 1200, which already moves a mean-over-40-images precision by 0.17 pp -- so the 0.1 pp bar is NOT claimed for bf16
    (measured: profiles/r03_trained_val.txt); what is held is 0.5 pp and the margin rule."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("trained_val", os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "trained_val.py"))
    tv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tv)
    r = tv.run(iters=400, n_val=40, verbose=True)
    # the point of training: decisions sit away from the thresholds (the random-weight stand-in has all of them within
    # 1e-3).  (How WELL it learned varies run to run -- F1 57 .. 92 after 400 iterations at lr 0.01 -- and is not the
    # subject here.)
    assert float(np.median(r["margins"])) > 0.02 and float((r["margins"] < 1e-3).mean()) < 0.05
    # (seeded: the draw of seed 0 learns the rule -- F1 86 on unseen scenes; a draw that learns it badly, e.g. seed 2 with
    # F1 32, still has its decisions far from the thresholds, which is all this test needs)
    assert r["mean"]["oracle"][2] > 15.0
    assert not r["flips"]["fp32"], r["flips"]["fp32"]
    for k, v in r["delta"]["fp32"].items():
        assert v == 0.0, (k, v)
    assert all(mg < 2e-3 for mg in r["flips"]["bf16"]), r["flips"]["bf16"]
    assert len(r["flips"]["bf16"]) <= 0.01 * 2 * r["npairs"]
    for k, v in r["delta"]["bf16"].items():
        assert abs(v) <= 0.5, (k, v)


# ------------------------------------------------------------------------------------------------
# edge shapes: the smallest input the network accepts (layer4 is 1x1), a single pair, odd batches
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,B", [(32, 3), (32, 5), (64, 1), (96, 3), (256, 32)])
def test_edge_shapes_forward_and_step(S, B):
    """S = 32: layer4 works on 1x1 maps (3x3 convolutions that only see padding around one pixel, a stride-2
    convolution from 2x2 to 1x1, a 1-pixel average pool); B = 1: each BatchNorm group is a single sample; S = 96 /
    odd B: row counts that are not multiples of the 128-row tile anywhere (unfused statistics path, ragged tiles);
    S = 256, B = 32: the reference's own per-GPU batch (BASELINE configs[0]) -- the launches where layers 3-4 have at most
    one 128-wide tile per CU and run 64 wide instead (conv_igemm.hip, IO_NT_SMALL_TILES).
    (Batch statistics over TWO values -- S = 32 with 2 samples -- are left out: every normalised value is +-1 by the
    sign of a difference, which no two fp32 implementations agree on.)"""
    algo = "InstaOrderNet_od"
    m = build(algo, 61, "kaiming")
    sd = synthetic.make_state_dict(61, 5, ALGO_CLASSES[algo], prefix="module.", style="kaiming")
    batch = synthetic.make_pair_batch(600 + S + B, B, S)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    x1 = torch.cat([t["modal1"], t["modal2"], t["rgb"]], 1)
    for training in (False, True):
        state = orc.state_from_numpy(sd, prefix="module.")
        m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        m.switch_to("train" if training else "eval")
        with torch.no_grad():
            zo = torch.cat(orc.resnet_forward(state, x1, training), 1)
            zh = torch.cat(m.model(x1.cuda()), 1)
        # batch statistics over 3 values (S = 32, B = 3, layer4) amplify fp32 rounding through nine such layers
        tol = 2e-2 if (training and S == 32 and B < 5) else FWD_TOL
        assert rel_err(zh.cpu().numpy(), zo.numpy()) < tol, (S, B, training)
    # one training step: loss against the oracle's
    state = orc.state_from_numpy(sd, prefix="module.")
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m.switch_to("train")
    set_input(m, algo, batch)
    logs_h, out = m.step()
    logs, _ = orc.train_step(state, {}, batch, algo, 0.0, 0.0)
    tol = 2e-2 if (S == 32 and B < 5) else FWD_TOL
    assert abs(float(out["loss"]) - float(logs["loss"])) < tol * max(1.0, abs(float(logs["loss"])))
    assert np.isfinite(float(m.net.flat_grads.norm()))
