"""Per-kernel parity: every HIP entry point of include/instaorder_hip.h, called through the C ABI
(ctypes), against the plain torch op the reference uses at that site, evaluated in fp64 on the CPU
on the same seeded inputs.  Tolerance for fp32 GEMM-like kernels: 2e-5 of the output scale (the
north-star bar is 1e-3; single kernels are far inside it)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from instaorder_amd import _lib, engine

pytestmark = pytest.mark.gpu
DEV = "cuda"


_KEEP = []


def P(t):
    """Device pointer of t.  The tensor is kept alive (launches are asynchronous: a temporary freed
    right after taking its pointer would be recycled by the caching allocator under the kernel)."""
    if t is None:
        return C.c_void_p(0)
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:-64]
    return C.c_void_p(t.data_ptr())


def ST():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def L():
    return _lib.lib()


def nhwc(x):     # NCHW cpu -> NHWC cuda fp32
    return x.permute(0, 2, 3, 1).contiguous().float().to(DEV)


def krsc(w, cin_store=None):   # OIHW -> [O][R*S][C] cuda
    O, I, R, S = w.shape
    k = w.permute(0, 2, 3, 1).contiguous().float()
    if cin_store and cin_store != I:
        pad = torch.zeros(O, R, S, cin_store - I)
        k = torch.cat([k, pad], 3).contiguous()
    return k.to(DEV)


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 1, 1, 0),
    (3, 10, 14, 64, 256, 1, 1, 0),      # M not a multiple of the 128-row tile
    (2, 16, 16, 256, 64, 1, 1, 0),
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 12, 20, 128, 128, 3, 2, 1),
    (1, 9, 9, 128, 128, 3, 1, 1),       # odd spatial size, M < tile
    (2, 16, 16, 256, 512, 1, 2, 0),     # strided 1x1 downsample
    (2, 8, 8, 512, 2048, 1, 1, 0),
    (1, 8, 8, 512, 512, 3, 1, 1),
    (1, 32, 32, 64, 64, 3, 1, 1),       # 32 | Wo: scalar row decode of the 64-wide filter-gradient tiles
    (1, 64, 64, 64, 128, 3, 2, 1),      # the same through a strided 3x3 with a 128 x 64 tile
    (1, 12, 12, 256, 128, 3, 1, 1),     # 4 | Wo on 128 x 128 tiles with rows that wrap inside a k-tile (Wo = 12)
    (8, 64, 64, 128, 512, 1, 1, 0),     # 1024 tiles of a 1x1 GEMM: the grid that keeps the two-block NT build
    (2, 32, 32, 128, 128, 3, 2, 1),     # LDS-DMA filter gradient: strided 3x3, 128 x 128 tiles
    (3, 16, 32, 128, 256, 3, 1, 1),     # ... non-square map, three samples, several k-tiles per split
]


def _conv_inputs(case, seed=0):
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)
    return x, w


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case):
    N, H, W, Cin, Cout, k, s, p = case
    x, w = _conv_inputs(case)
    ref = F.conv2d(x, w, stride=s, padding=p)
    y = torch.full((N, ref.shape[2], ref.shape[3], Cout), float("nan"), device=DEV)
    _lib.check(L().io_conv2d_fwd(P(nhwc(x)), P(krsc(w)), P(y), N, H, W, Cin, Cout, k, k, s, p, ST()), "conv")
    assert relerr(y.permute(0, 3, 1, 2), ref) < 2e-5


XF_CASES = [((4, 16, 16, 64, 64, 3, 1, 1), 2), ((2, 16, 16, 64, 256, 1, 1, 0), 1), ((8, 16, 16, 128, 128, 3, 2, 1), 2),
            ((2, 16, 16, 256, 128, 3, 1, 1), 1), ((4, 8, 8, 512, 2048, 1, 1, 0), 2), ((2, 12, 32, 128, 128, 3, 1, 1), 2)]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case,G", XF_CASES)
def test_conv_fwd_input_transform(case, G, dtype):
    """io_conv2d_fwd_xf_dt: conv(relu(bn(x)), w) with the BatchNorm scale / shift + ReLU applied to the A operand while
    it is staged == F.conv2d(F.relu(x * scale + shift), w): per-group tables, zero padding AFTER the transform (a zero
    read from the padding must not become relu(shift)), fp32 and bf16 storage."""
    N, H, W, Cin, Cout, k, s, p = case
    x, w = _conv_inputs(case, 7)
    g = torch.Generator().manual_seed(3)
    scale = (torch.randn(G, Cin, generator=g, dtype=torch.float64) * 0.7 + 0.3)        # both signs
    shift = torch.randn(G, Cin, generator=g, dtype=torch.float64) * 0.5 + 0.4           # relu(shift) > 0 mostly
    mean = torch.randn(G, Cin, generator=g, dtype=torch.float64) * 0.3
    bf = dtype == "bf16"
    if bf:
        x, w = x.bfloat16().double(), w.bfloat16().double()
    per = N // G
    v4 = lambda t, gi: t[gi].view(1, -1, 1, 1)      # noqa: E731
    xa = torch.cat([F.relu((x[gi * per:(gi + 1) * per] - v4(mean, gi)) * v4(scale, gi) + v4(shift, gi)) for gi in range(G)])
    if bf:
        xa = xa.bfloat16().double()              # the kernel rounds the transformed operand to bf16 for the MFMA
    ref = F.conv2d(xa, w, stride=s, padding=p)
    Ho, Wo = ref.shape[2:]
    td = torch.bfloat16 if bf else torch.float32
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV, dtype=td)
    _lib.check(L().io_conv2d_fwd_xf_dt(P(nhwc(x).to(td)), P(krsc(w).to(td)), P(y), N, H, W, Cin, Cout, k, k, s, p, G,
                                       P(mean.float().to(DEV)), P(scale.float().to(DEV)), P(shift.float().to(DEV)),
                                       None, None, None, None, 0.1, 1e-5, None, None, None, None, None, 0,
                                       1 if bf else 0, ST()), "conv_xf")
    assert relerr(y.float().permute(0, 3, 1, 2), ref) < (6e-3 if bf else 2e-5)
    # with statistics: the tables of the NEXT BatchNorm from the same launch
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean2, rstd, sc2, sh2 = (torch.empty(G * Cout, device=DEV) for _ in range(4))
    nws = L().io_conv2d_bnstats_workspace_floats(N, H, W, Cout, k, k, s, p, G)
    ws = torch.empty(nws, device=DEV)
    y2 = torch.empty_like(y)
    _lib.check(L().io_conv2d_fwd_xf_dt(P(nhwc(x).to(td)), P(krsc(w).to(td)), P(y2), N, H, W, Cin, Cout, k, k, s, p, G,
                                       P(mean.float().to(DEV)), P(scale.float().to(DEV)), P(shift.float().to(DEV)),
                                       P(gamma.to(DEV)), P(beta.to(DEV)), P(rm), P(rv), 0.1, 1e-5, P(mean2), P(rstd),
                                       P(sc2), P(sh2), P(ws), nws, 1 if bf else 0, ST()), "conv_xf+stats")
    assert torch.equal(y2, y)
    mref = torch.stack([ref[gi * per:(gi + 1) * per].mean((0, 2, 3)) for gi in range(G)])
    vref = torch.stack([ref[gi * per:(gi + 1) * per].var((0, 2, 3), unbiased=False) for gi in range(G)])
    assert relerr(mean2.view(G, Cout), mref) < (6e-3 if bf else 2e-5)
    assert relerr(rstd.view(G, Cout), 1.0 / torch.sqrt(vref + 1e-5)) < (6e-3 if bf else 2e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("N,H,W,C,G", [(4, 16, 16, 64, 2), (2, 9, 13, 64, 1), (6, 32, 32, 64, 2)])
def test_maxpool_input_transform(N, H, W, C, G, dtype):
    """io_maxpool_fwd_xf_dt == F.max_pool2d(relu(x * scale + shift), 3, 2, 1) incl. indices that route the backward."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    bf = dtype == "bf16"
    if bf:
        x = x.bfloat16().double()
    scale = torch.randn(G, C, generator=g, dtype=torch.float64).float().double()
    shift = (torch.randn(G, C, generator=g, dtype=torch.float64) * 0.5).float().double()
    mean = (torch.randn(G, C, generator=g, dtype=torch.float64) * 0.3).float().double()
    per = N // G
    v4 = lambda t, gi: t[gi].view(1, -1, 1, 1)      # noqa: E731
    xa = torch.cat([F.relu((x[gi * per:(gi + 1) * per] - v4(mean, gi)) * v4(scale, gi) + v4(shift, gi))
                    for gi in range(G)]).requires_grad_(True)
    ref = F.max_pool2d(xa, 3, 2, 1)
    Ho, Wo = ref.shape[2:]
    td = torch.bfloat16 if bf else torch.float32
    out = torch.empty(N, Ho, Wo, C, device=DEV, dtype=td)
    idx = torch.empty(N * Ho * Wo * C // 4, device=DEV, dtype=torch.int32)
    _lib.check(L().io_maxpool_fwd_xf_dt(P(nhwc(x).to(td)), N, H, W, C, P(out), P(idx), G, P(mean.float().to(DEV)),
                                        P(scale.float().to(DEV)), P(shift.float().to(DEV)), 1 if bf else 0, ST()),
               "maxpool_xf")
    assert relerr(out.float().permute(0, 3, 1, 2), ref.detach()) < (4e-3 if bf else 1e-6)
    if not bf:      # the stored arg-max indices scatter a gradient exactly as autograd does (ties aside: relu zeros)
        dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
        ref.backward(dy)
        dx = torch.empty(N, H, W, C, device=DEV)
        _lib.check(L().io_maxpool_bwd(P(nhwc(dy)), P(idx), N, H, W, C, P(dx), ST()), "maxpool_bwd")
        # where several window entries are clipped to the same 0 the winner is a convention; compare the pooled sums
        got = dx.permute(0, 3, 1, 2).double().cpu()
        assert abs(float(got.sum()) - float(xa.grad.sum())) < 1e-3 * float(xa.grad.abs().sum())
        nz = (ref.detach() > 0)                                  # windows with a positive maximum have a unique winner
        want = torch.zeros_like(x)
        # recompute autograd's routing restricted to those windows
        xa2 = xa.detach().clone().requires_grad_(True)
        (F.max_pool2d(xa2, 3, 2, 1) * dy * nz).sum().backward()
        dy_nz = (dy * nz)
        dx2 = torch.empty(N, H, W, C, device=DEV)
        _lib.check(L().io_maxpool_bwd(P(nhwc(dy_nz)), P(idx), N, H, W, C, P(dx2), ST()), "maxpool_bwd")
        assert relerr(dx2.permute(0, 3, 1, 2), xa2.grad) < 1e-6


@pytest.mark.parametrize("case,G", [((4, 16, 16, 64, 64, 1, 1, 0), 2), ((2, 16, 16, 64, 256, 3, 1, 1), 1),
                                    ((8, 16, 16, 256, 128, 3, 2, 1), 2), ((64, 8, 8, 512, 2048, 1, 1, 0), 2)])
def test_conv_fwd_with_fused_bn_statistics(case, G):
    """conv + BN statistics in the GEMM epilogue == conv, then nn.BatchNorm2d's batch statistics per group
    (incl. |mean| >> sigma outputs: the per-tile mean / M2 merge must not cancel)."""
    N, H, W, Cin, Cout, k, s, p = case
    x, w = _conv_inputs(case, 4)
    w = w + 0.5 / np.sqrt(Cin * k * k)              # biased filters -> conv outputs with a large mean
    ref = F.conv2d(x, w, stride=s, padding=p)
    Ho, Wo = ref.shape[2:]
    g = torch.Generator().manual_seed(11)
    gamma = torch.rand(Cout, generator=g) + 0.5
    beta = torch.randn(Cout, generator=g)
    rm, rv = torch.zeros(Cout, dtype=torch.float64), torch.ones(Cout, dtype=torch.float64)
    means, rstds = [], []
    for gi in range(G):
        yg = ref[gi * N // G:(gi + 1) * N // G]
        mu, var = yg.mean((0, 2, 3)), yg.var((0, 2, 3), unbiased=False)
        means.append(mu)
        rstds.append(1.0 / torch.sqrt(var + 1e-5))
        n = yg.numel() / Cout
        rm = 0.9 * rm + 0.1 * mu
        rv = 0.9 * rv + 0.1 * var * n / (n - 1)
    f = lambda t: t.float().to(DEV).contiguous()
    y = torch.empty(N, Ho, Wo, Cout, device=DEV)
    d_rm, d_rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean, rstd, scale, shift = (torch.empty(G * Cout, device=DEV) for _ in range(4))
    nws = L().io_conv2d_bnstats_workspace_floats(N, H, W, Cout, k, k, s, p, G)
    ws = torch.empty(nws, device=DEV)
    _lib.check(L().io_conv2d_fwd_bnstats(P(nhwc(x)), P(krsc(w)), P(y), N, H, W, Cin, Cout, k, k, s, p, G, P(f(gamma)),
                                         P(f(beta)), P(d_rm), P(d_rv), 0.1, 1e-5, P(mean), P(rstd), P(scale), P(shift),
                                         P(ws), nws, ST()), "conv+stats")
    assert relerr(y.permute(0, 3, 1, 2), ref) < 2e-5
    assert relerr(mean.view(G, Cout), torch.stack(means)) < 2e-5
    assert relerr(rstd.view(G, Cout), torch.stack(rstds)) < 1e-4
    assert relerr(d_rm, rm) < 2e-5 and relerr(d_rv, rv) < 1e-4
    assert relerr(scale.view(G, Cout), torch.stack(rstds) * gamma.double()) < 1e-4
    assert torch.equal(shift.view(G, Cout).cpu(), beta.expand(G, Cout))


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(case):
    N, H, W, Cin, Cout, k, s, p = case
    x, w = _conv_inputs(case, 1)
    x.requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    ref = torch.autograd.grad(y, x, dy)[0]
    wk = krsc(w)
    wt = torch.empty(Cin, k * k, Cout, device=DEV)
    _lib.check(L().io_filter_transpose(P(wk), Cout, k * k, Cin, P(wt), ST()), "transpose")
    assert torch.equal(wt.cpu(), wk.cpu().view(Cout, k * k, Cin).permute(2, 1, 0).contiguous())
    dx = torch.full((N, H, W, Cin), float("nan"), device=DEV)
    _lib.check(L().io_conv2d_dgrad(P(nhwc(dy)), P(wt), P(dx), None, None, N, H, W, Cin, Cout, k, k, s, p, ST()), "dgrad")
    assert relerr(dx.permute(0, 3, 1, 2), ref) < 2e-5
    # with accumulation into an existing tensor (residual / downsample sum), in place
    base = torch.randn(N, H, W, Cin, generator=torch.Generator().manual_seed(9)).to(DEV)
    acc = base.clone()
    _lib.check(L().io_conv2d_dgrad(P(nhwc(dy)), P(wt), P(acc), P(acc), None, N, H, W, Cin, Cout, k, k, s, p, ST()),
               "dgrad+add")
    assert relerr(acc.permute(0, 3, 1, 2), ref + base.cpu().double().permute(0, 3, 1, 2)) < 2e-5
    # ... and with the ReLU mask of the tensor the gradient belongs to applied in the epilogue
    msk = torch.randn(N, H, W, Cin, generator=torch.Generator().manual_seed(10)).to(DEV)
    acc2 = base.clone()
    _lib.check(L().io_conv2d_dgrad(P(nhwc(dy)), P(wt), P(acc2), P(acc2), P(msk), N, H, W, Cin, Cout, k, k, s, p, ST()),
               "dgrad+add+mask")
    want = (ref + base.cpu().double().permute(0, 3, 1, 2)) * (msk.cpu().permute(0, 3, 1, 2) > 0)
    assert relerr(acc2.permute(0, 3, 1, 2), want) < 2e-5


@pytest.mark.parametrize("N,H,Cin,Cout,k,G", [(4, 16, 64, 256, 1, 2), (2, 16, 128, 128, 3, 1), (8, 8, 512, 64, 1, 2)])
def test_conv_dgrad_with_fused_bn_backward(N, H, Cin, Cout, k, G):
    """autograd through conv(relu(bn(y))) w.r.t. y / gamma / beta in ONE fused call (dgrad + ReLU mask from y +
    BN-backward reductions in the epilogue + apply) against torch in fp64."""
    g = torch.Generator().manual_seed(N * H + Cin)
    y = (torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64) * 0.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(Cin, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(Cin, generator=g, dtype=torch.float64)).requires_grad_(True)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)
    outs = []
    for gi in range(G):
        sl = slice(gi * N // G, (gi + 1) * N // G)
        outs.append(F.batch_norm(y[sl], None, None, gamma, beta, True, 0.1, 1e-5))
    a = F.relu(torch.cat(outs, 0))
    o = F.conv2d(a, w, padding=k // 2)
    do = torch.randn(o.shape, generator=g, dtype=torch.float64)
    gy, gg, gb = torch.autograd.grad(o, [y, gamma, beta], do)
    f = lambda t: t.detach().float().to(DEV).contiguous()
    M = N * H * H
    yd = nhwc(y.detach())
    mean, rstd, scale, shift = (torch.empty(G * Cin, device=DEV) for _ in range(4))
    npart = L().io_bn_partial_floats(M, Cin, G)
    part = torch.empty(npart, device=DEV)
    _lib.check(L().io_bn_stats_finalize(P(yd), M, Cin, G, P(f(gamma)), P(f(beta)), None, None, 0.1, 1e-5, P(mean), P(rstd),
                                        P(scale), P(shift), P(part), npart, ST()), "stats")
    wt = krsc(w).view(Cout, k * k, Cin).permute(2, 1, 0).contiguous()
    tiles = M // 128
    nws = 2 * ((tiles + tiles // 64 + G + 2) * Cin) + 2 * G * Cin
    ws = torch.empty(nws, device=DEV)
    dz, dyb = torch.empty_like(yd), torch.empty_like(yd)
    dgam, dbet = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    _lib.check(L().io_conv2d_dgrad_bnbwd(P(nhwc(do)), P(wt), P(dz), N, H, H, Cin, Cout, k, k, k // 2, P(yd), G, P(f(gamma)),
                                         P(mean), P(rstd), P(scale), P(shift), P(dgam), P(dbet), P(dyb), P(ws), nws, ST()),
               "dgrad+bnbwd")
    assert relerr(dyb.permute(0, 3, 1, 2), gy) < 3e-5
    assert relerr(dgam, gg) < 3e-5 and relerr(dbet, gb) < 3e-5
    da = torch.autograd.grad(F.conv2d(a.detach().requires_grad_(True), w, padding=k // 2), [], do, allow_unused=True) \
        if False else None
    ref_dz = torch.autograd.grad(o, a, do, retain_graph=False) if False else None
    # dz = (conv data gradient) * [a > 0]
    a2 = a.detach().requires_grad_(True)
    dz_ref = torch.autograd.grad(F.conv2d(a2, w, padding=k // 2), a2, do)[0] * (a.detach() > 0)
    assert relerr(dz.permute(0, 3, 1, 2), dz_ref) < 3e-5


@pytest.mark.parametrize("N,H,Cin,Cout,G", [(4, 16, 256, 64, 2), (8, 8, 512, 128, 2), (2, 16, 1024, 256, 1), (16, 8, 2048, 512, 2)])
def test_conv_fwd_residual_operand(N, H, Cin, Cout, G):
    """io_conv2d_fwd_resid: the block output relu(bn3(y3) + identity) built on the staged operand of the next block's
    conv1 and written out on the way == io_bn_apply (bit for bit) followed by the plain convolution (bit for bit: the
    same operand values go through the same GEMM), incl. the statistics of the result."""
    g = torch.Generator().manual_seed(N + Cin)
    y3 = torch.randn(N, H, H, Cin, generator=g).to(DEV)
    idt = torch.randn(N, H, H, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, 1, Cin, generator=g) / np.sqrt(Cin)).to(DEV)
    mean = (torch.randn(G * Cin, generator=g) * 0.3).to(DEV)
    scale = (torch.randn(G * Cin, generator=g) * 0.7 + 0.3).to(DEV)
    shift = (torch.randn(G * Cin, generator=g) * 0.5).to(DEV)
    M = N * H * H
    out_ref = torch.empty_like(y3)
    _lib.check(L().io_bn_apply(P(y3), M, Cin, G, 1, P(mean), P(scale), P(shift), P(idt), None, None, None, 1, P(out_ref), ST()),
               "bn_apply")
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    nws = L().io_conv2d_bnstats_workspace_floats(N, H, H, Cout, 1, 1, 1, 0, G)

    def stats():
        return [torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)] + [torch.empty(G * Cout, device=DEV) for _ in range(4)]
    sa, sb = stats(), stats()
    y_ref = torch.empty(N, H, H, Cout, device=DEV)
    ws = torch.empty(nws, device=DEV)
    _lib.check(L().io_conv2d_fwd_bnstats(P(out_ref), P(w), P(y_ref), N, H, H, Cin, Cout, 1, 1, 1, 0, G, P(gamma), P(beta),
                                         P(sa[0]), P(sa[1]), 0.1, 1e-5, P(sa[2]), P(sa[3]), P(sa[4]), P(sa[5]), P(ws), nws, ST()),
               "conv+stats")
    y = torch.full_like(y_ref, float("nan"))
    out = torch.full_like(y3, float("nan"))
    ws2 = torch.empty(nws, device=DEV)
    _lib.check(L().io_conv2d_fwd_resid(P(y3), P(idt), P(w), P(y), P(out), N, H, H, Cin, Cout, G, P(mean), P(scale), P(shift),
                                       P(gamma), P(beta), P(sb[0]), P(sb[1]), 0.1, 1e-5, P(sb[2]), P(sb[3]), P(sb[4]), P(sb[5]),
                                       P(ws2), nws, ST()), "conv_resid")
    assert torch.equal(out, out_ref)
    assert torch.equal(y, y_ref)
    for a, b in zip(sa, sb):
        assert torch.equal(a, b)
    # and against fp64 torch, end to end
    per = N // G
    o64 = torch.cat([F.relu((y3[i * per:(i + 1) * per].double().cpu() - mean.view(G, Cin)[i].double().cpu())
                            * scale.view(G, Cin)[i].double().cpu() + shift.view(G, Cin)[i].double().cpu()
                            + idt[i * per:(i + 1) * per].double().cpu()) for i in range(G)])
    ref = F.conv2d(o64.permute(0, 3, 1, 2), w.double().cpu().view(Cout, Cin, 1, 1))
    assert relerr(y.permute(0, 3, 1, 2), ref) < 2e-5


XB_CASES = [(4, 16, 64, 256, 1, 2, False), (4, 16, 128, 128, 3, 2, False), (8, 8, 512, 64, 1, 2, False),
            (2, 16, 64, 64, 3, 1, False), (4, 16, 256, 64, 1, 2, True), (2, 32, 64, 128, 3, 2, True),
            (16, 8, 128, 2048, 1, 2, False)]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("N,H,Cin,Cout,k,G,resid", XB_CASES)
def test_conv_dgrad_fused_operand_transform(N, H, Cin, Cout, k, G, resid, dtype):
    """The data-gradient launch of the training step, everything on: autograd through
    z = bn_b(conv(relu(bn_a(y_a)))) given the (masked) gradient dz_b of z.  BatchNorm b's backward has no apply pass --
    io_bn_bwd_coefs_dt turns its reductions into per-channel tables and io_conv2d_dgrad_fused_dt evaluates
    dy_b = a*dz_b + b*y_b + c on its staged operand (side output: dy_b for the filter gradient); the epilogue masks with
    relu(bn_a(y_a)) > 0 recomputed from y_a (or, `resid`: adds a residual gradient and masks with a stored activation, as
    conv1's data gradient does), rebuilds that activation, and leaves the tile partials from which
    io_bn_bwd_coefs_from_tile_partials makes BatchNorm a's tables.  Against torch autograd in fp64."""
    bf = dtype == "bf16"
    td = torch.bfloat16 if bf else torch.float32
    rt = (lambda t: t.bfloat16().double()) if bf else (lambda t: t.float().double())
    g = torch.Generator().manual_seed(N * H + Cin + k)
    y_a = rt(torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64) * 0.7 + 0.2).requires_grad_(True)
    gam_a = (1 + 0.2 * torch.randn(Cin, generator=g, dtype=torch.float64)).float().double().requires_grad_(True)
    bet_a = (0.2 * torch.randn(Cin, generator=g, dtype=torch.float64)).float().double().requires_grad_(True)
    gam_b = (1 + 0.2 * torch.randn(Cout, generator=g, dtype=torch.float64)).float().double().requires_grad_(True)
    bet_b = (0.2 * torch.randn(Cout, generator=g, dtype=torch.float64)).float().double().requires_grad_(True)
    w = rt(torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k))
    per = N // G
    bn = lambda t, ga, be: torch.cat([F.batch_norm(t[i * per:(i + 1) * per], None, None, ga, be, True, 0.1, 1e-5)   # noqa: E731
                                      for i in range(G)])
    t_a = bn(y_a, gam_a, bet_a)
    t_a.retain_grad()
    if resid:      # conv1's form: the tensor the gradient belongs to is a block input with its own stored activation
        act = rt(torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64))
        a = t_a
    else:
        a = F.relu(t_a)
    y_b = rt(F.conv2d(a, w, padding=k // 2).detach()).requires_grad_(True)     # what the forward stored (rounded once)
    z = bn(y_b, gam_b, bet_b)
    dz_b = rt(torch.randn(z.shape, generator=g, dtype=torch.float64) *
              (torch.rand(z.shape, generator=g, dtype=torch.float64) > 0.4))
    dyb_ref, dgb_ref, dbb_ref = torch.autograd.grad(z, [y_b, gam_b, bet_b], dz_b)
    f = lambda t: t.detach().float().to(DEV).contiguous()    # noqa: E731
    M = N * H * H
    lib = L()
    dt = 1 if bf else 0
    # forward tables of both BatchNorms
    def tables(yt, C, ga, be):
        mean, rstd, scale, shift = (torch.empty(G * C, device=DEV) for _ in range(4))
        npart = lib.io_bn_partial_floats(M, C, G)
        part = torch.empty(npart, device=DEV)
        _lib.check(lib.io_bn_stats_finalize_dt(P(yt), M, C, G, P(f(ga)), P(f(be)), None, None, 0.1, 1e-5, P(mean), P(rstd),
                                               P(scale), P(shift), P(part), npart, dt, ST()), "stats")
        return mean, rstd, scale, shift
    ya_d, yb_d = nhwc(y_a.detach()).to(td), nhwc(y_b.detach()).to(td)
    mean_a, rstd_a, scale_a, shift_a = tables(ya_d, Cin, gam_a, bet_a)
    mean_b, rstd_b, scale_b, shift_b = tables(yb_d, Cout, gam_b, bet_b)
    # BatchNorm b: reductions -> tables (no apply pass)
    dzb_d = nhwc(dz_b).to(td)
    coef_b = torch.empty(3 * G * Cout, device=DEV)
    dgb, dbb = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    npart = lib.io_bn_partial_floats(M, Cout, G)
    part = torch.empty(npart, device=DEV)
    _lib.check(lib.io_bn_bwd_coefs_dt(P(dzb_d), P(yb_d), M, Cout, G, P(f(gam_b)), P(mean_b), P(rstd_b), P(dgb), P(dbb),
                                      P(coef_b), P(part), npart, dt, ST()), "bn_bwd_coefs")
    tol = 2e-2 if bf else 3e-5
    assert relerr(dgb, dgb_ref) < tol and relerr(dbb, dbb_ref) < tol
    # the fused launch
    wt = krsc(w).view(Cout, k * k, Cin).permute(2, 1, 0).contiguous().to(td)
    nt = lib.io_bn_tile_partial_floats(M, Cin, G)
    p1, p2 = torch.empty(nt, device=DEV), torch.empty(nt, device=DEV)
    dx = torch.full((N, H, H, Cin), float("nan"), device=DEV, dtype=td)
    dyb_out = torch.full((N, H, H, Cout), float("nan"), device=DEV, dtype=td)
    opt = _lib.DgradFused()
    opt.xb_y, opt.xb_coef, opt.xb_dy_out = yb_d.data_ptr(), coef_b.data_ptr(), dyb_out.data_ptr()
    opt.ep_y, opt.ep_mean, opt.ep_rstd, opt.ep_p1, opt.ep_p2 = (ya_d.data_ptr(), mean_a.data_ptr(), rstd_a.data_ptr(),
                                                                p1.data_ptr(), p2.data_ptr())
    keep = [ya_d, yb_d, coef_b, mean_a, rstd_a, scale_a, shift_a]
    if resid:
        base = rt(torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64))
        base_d, act_d = nhwc(base).to(td), nhwc(act).to(td)
        opt.add, opt.relu_mask = base_d.data_ptr(), act_d.data_ptr()
        keep += [base_d, act_d]
        aout = None
    else:
        aout = torch.full((N, H, H, Cin), float("nan"), device=DEV, dtype=td)
        opt.ep_scale, opt.ep_shift, opt.ep_act_out = scale_a.data_ptr(), shift_a.data_ptr(), aout.data_ptr()
    _lib.check(lib.io_conv2d_dgrad_fused_dt(P(dzb_d), P(wt), P(dx), N, H, H, Cin, Cout, k, k, k // 2, G, C.byref(opt), dt,
                                            ST()), "dgrad_fused")
    assert relerr(dyb_out.float().permute(0, 3, 1, 2), dyb_ref) < (1e-2 if bf else 3e-5)
    # gradient of bn_a's output: conv data gradient of dy_b (the kernel's own, rounded dy_b in bf16), masked
    dyb_used = dyb_out.float().permute(0, 3, 1, 2).double().cpu() if bf else dyb_ref
    a2 = a.detach().requires_grad_(True)
    da = torch.autograd.grad(F.conv2d(a2, w, padding=k // 2), a2, dyb_used)[0]
    dza_ref = (da + base) * (act > 0) if resid else da * (t_a.detach() > 0)
    if bf and not resid:     # knife-edge mask decisions of a bf16-rounded bn(y): compare where the sign is certain
        sure = (t_a.detach().abs() > 2e-2)
        assert relerr(dx.float().permute(0, 3, 1, 2).cpu() * sure, dza_ref * sure) < 1e-2
    else:
        assert relerr(dx.float().permute(0, 3, 1, 2), dza_ref) < (1e-2 if bf else 3e-5)
    if aout is not None:
        assert relerr(aout.float().permute(0, 3, 1, 2), F.relu(t_a.detach())) < (1e-2 if bf else 3e-5)
    # BatchNorm a from the tile partials: dgamma / dbeta and the tables that give dy_a
    coef_a = torch.empty(3 * G * Cin, device=DEV)
    dga, dba = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    _lib.check(lib.io_bn_bwd_coefs_from_tile_partials(P(p1), P(p2), M, Cin, G, P(f(gam_a)), P(mean_a), P(rstd_a), P(dga),
                                                      P(dba), P(coef_a), ST()), "coefs_from_tiles")
    dza_k = dx.float().permute(0, 3, 1, 2).double().cpu()        # the kernel's own dz_a (fp64 reference chain from it)
    dya_ref, dga_ref, dba_ref = torch.autograd.grad(t_a, [y_a, gam_a, bet_a], dza_k)
    assert relerr(dga, dga_ref) < (2e-2 if bf else 5e-5) and relerr(dba, dba_ref) < (2e-2 if bf else 5e-5)
    ca = coef_a.view(3, G, Cin).double().cpu()
    grp = torch.arange(N) // per
    v = lambda t: t[grp].view(N, Cin, 1, 1)          # noqa: E731
    dya = v(ca[0]) * dza_k + v(ca[1]) * y_a.detach() + v(ca[2])
    assert relerr(dya, dya_ref) < (5e-3 if bf else 5e-5)
    del keep


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(case):
    N, H, W, Cin, Cout, k, s, p = case
    x, w = _conv_inputs(case, 2)
    w.requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    ref = torch.autograd.grad(y, w, dy)[0]
    nb = L().io_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, k, k, s, p)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.full((Cout, k * k, Cin), float("nan"), device=DEV)
    _lib.check(L().io_conv2d_wgrad(P(nhwc(x)), P(nhwc(dy)), P(dw), N, H, W, Cin, Cout, k, k, s, p, P(ws), nb, ST()),
               "wgrad")
    got = dw.view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert relerr(got, ref) < 2e-5


@pytest.mark.parametrize("N,S", [(2, 64), (1, 40)])
def test_stem(N, S):
    """7x7/2 conv on the 5-channel (mask_a, mask_b, R, G, B) input stored with 8 channels."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 5, S, S, generator=g, dtype=torch.float64)
    w = (torch.randn(64, 5, 7, 7, generator=g, dtype=torch.float64) / 15.0).requires_grad_(True)
    ref = F.conv2d(x, w, stride=2, padding=3)
    x8 = engine.pack_nchw(x.float().to(DEV))
    assert torch.equal(x8[..., :5].cpu(), x.float().permute(0, 2, 3, 1)) and float(x8[..., 5:].abs().max()) == 0
    Ho = ref.shape[2]
    y = torch.full((N, Ho, Ho, 64), float("nan"), device=DEV)
    _lib.check(L().io_conv2d_fwd(P(x8), P(krsc(w.detach(), 8)), P(y), N, S, S, 8, 64, 7, 7, 2, 3, ST()), "stem")
    assert relerr(y.permute(0, 3, 1, 2), ref) < 2e-5
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gref = torch.autograd.grad(ref, w, dy)[0]
    nb = L().io_conv2d_wgrad_workspace_bytes(N, S, S, 8, 64, 7, 7, 2, 3)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.full((64, 49, 8), float("nan"), device=DEV)
    _lib.check(L().io_conv2d_wgrad(P(x8), P(nhwc(dy)), P(dw), N, S, S, 8, 64, 7, 7, 2, 3, P(ws), nb, ST()), "stem wg")
    got = dw.view(64, 7, 7, 8).permute(0, 3, 1, 2)
    assert relerr(got[:, :5], gref) < 2e-5
    assert float(got[:, 5:].abs().max()) == 0.0      # padded channels never receive gradient


@pytest.mark.parametrize("N,S,G", [(2, 256, 2), (3, 256, 1), (20, 256, 2), (2, 512, 1), (2, 128, 2)])
def test_stem_exact_k_rows(N, S, G):
    """The fp32 step's stem: reduction over the 5 REAL channels, BatchNorm statistics from the convolution's epilogue.
    128 | Wo runs the row-persistent kernel of csrc/stem.hip -- N = 2: one output row per block (no steady state),
    N = 3: two rows per block, N = 20: ten (the branch-free steady-state body), S = 512: two column blocks per output row;
    S = 128 (Wo = 64) is the generic exact-K kernel.  Every output element against an fp64 convolution, the statistics
    against the fp64 mean / variance of the output per sample group."""
    g = torch.Generator().manual_seed(11 + N + S)
    x = torch.randn(N, 5, S, S, generator=g, dtype=torch.float64)
    x[:, :2] = (x[:, :2] > 0).double()                      # the two mask planes
    w = torch.randn(64, 5, 7, 7, generator=g, dtype=torch.float64) / 15.0
    ref = F.conv2d(x, w, stride=2, padding=3)
    x8 = engine.pack_nchw(x.float().to(DEV))
    Ho = S // 2
    y = torch.full((N, Ho, Ho, 64), float("nan"), device=DEV)
    gamma, beta = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    mean, rstd, scale, shift = (torch.full((G * 64,), float("nan"), device=DEV) for _ in range(4))
    nws = L().io_conv2d_bnstats_workspace_floats(N, S, S, 64, 7, 7, 2, 3, G)
    ws = torch.empty(nws, device=DEV)
    packed = torch.empty(L().io_stem_packed_floats(5), device=DEV)
    _lib.check(L().io_stem_fwd_bnstats_exact(P(x8), P(krsc(w, 8)), P(y), N, S, S, 5, G, P(gamma), P(beta), P(rm), P(rv), 0.1,
                                             1e-5, P(mean), P(rstd), P(scale), P(shift), P(ws), nws, P(packed), ST()),
               "stem exact")
    assert relerr(y.permute(0, 3, 1, 2), ref) < 2e-5
    rg = ref.view(G, N // G, 64, Ho, Ho) if N % G == 0 else None
    assert rg is not None
    mref = rg.mean(dim=(1, 3, 4))
    vref = rg.var(dim=(1, 3, 4), unbiased=False)
    assert float((mean.view(G, 64).cpu().double() - mref).abs().max()) < 2e-5 * float(vref.sqrt().max())
    assert relerr(rstd.view(G, 64), 1.0 / torch.sqrt(vref + 1e-5)) < 2e-5
    # filter gradient over the same 5 real channels (row-persistent where the forward is; one partial per block)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    wq = w.clone().requires_grad_(True)
    gref = torch.autograd.grad(F.conv2d(x, wq, stride=2, padding=3), wq, dy)[0]
    nb = L().io_stem_wgrad_exact_workspace_bytes(N, S, S, 5)
    wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.full((64, 49, 8), float("nan"), device=DEV)
    _lib.check(L().io_stem_wgrad_exact(P(x8), P(nhwc(dy)), P(dw), N, S, S, 5, P(wsb), nb, P(packed), ST()), "stem wgrad exact")
    got = dw.view(64, 7, 7, 8).permute(0, 3, 1, 2)
    assert relerr(got[:, :5], gref) < 2e-5
    assert float(got[:, 5:].abs().max()) == 0.0
    if S % 256:
        return
    # ... and with bn1's backward folded into the gradient kernel's staging (resnet_cls.py:157-158: relu(bn1(conv1(x)))):
    # da = gradient of the ReLU output; reference = autograd through batch_norm + relu + conv in fp64, per sample group
    gamma = (torch.rand(64, generator=g, dtype=torch.float64) + 0.5)
    beta = torch.randn(64, generator=g, dtype=torch.float64) * 0.3
    da = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    wq = w.clone().requires_grad_(True)
    gq, bq = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    _lib.check(L().io_stem_fwd_bnstats_exact(P(x8), P(krsc(w, 8)), P(y), N, S, S, 5, G, P(gamma.float().to(DEV)),
                                             P(beta.float().to(DEV)), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(scale),
                                             P(shift), P(ws), nws, P(packed), ST()), "stem exact (tables)")
    # the ReLU mask as the kernels see it: sign of fma(y - mean, scale, shift) on the fp32 output and tables (the product of
    # two floats is exact in fp64, so this reproduces the bit; a mask from the fp64 forward flips a knife-edge element or
    # two in 21 M, each worth 1e-3 of the largest filter-gradient entry)
    yv = y.view(G, N // G, Ho, Ho, 64)
    tabs = [t.view(G, 1, 1, 1, 64) for t in (mean, scale, shift)]
    mask = ((yv - tabs[0]).double() * tabs[1].double() + tabs[2].double()) > 0
    mask = mask.view(N, Ho, Ho, 64).permute(0, 3, 1, 2).cpu()
    yq = F.conv2d(x, wq, stride=2, padding=3)
    outs = [F.batch_norm(yq[k * (N // G):(k + 1) * (N // G)], None, None, gq, bq, True, 0.1, 1e-5) for k in range(G)]
    gw, gg, gb = torch.autograd.grad(torch.cat(outs), (wq, gq, bq), da * mask)
    M = N * Ho * Ho
    npart = L().io_bn_partial_floats(M, 64, G)
    part = torch.empty(npart, device=DEV)
    coef = torch.full((3 * G * 64,), float("nan"), device=DEV)
    dgam, dbet = torch.full((64,), float("nan"), device=DEV), torch.full((64,), float("nan"), device=DEV)
    dw2 = torch.full((64, 49, 8), float("nan"), device=DEV)
    _lib.check(L().io_stem_wgrad_exact_bn(P(x8), P(nhwc(da)), P(y), P(dw2), N, S, S, 5, G, P(gamma.float().to(DEV)), P(mean),
                                          P(rstd), P(scale), P(shift), P(dgam), P(dbet), P(coef), P(part), npart, P(wsb), nb,
                                          P(packed), ST()), "stem wgrad exact + bn")
    got2 = dw2.view(64, 7, 7, 8).permute(0, 3, 1, 2)
    # (BatchNorm backward subtracts the two projections: the result is small against its terms, hence the looser bound)
    assert relerr(got2[:, :5], gw) < 2e-4
    assert float(got2[:, 5:].abs().max()) == 0.0
    assert relerr(dgam, gg) < 1e-4 and relerr(dbet, gb) < 1e-4


@pytest.mark.parametrize("N,H,C,G", [(4, 8, 64, 1), (4, 8, 256, 2), (6, 4, 2048, 2), (2, 16, 128, 1), (8, 1, 512, 2)])
def test_batchnorm(N, H, C, G):
    """train-mode BN (+ReLU, + residual) forward, running statistics, backward; G groups == G
    independent nn.BatchNorm2d calls on consecutive sample groups."""
    g = torch.Generator().manual_seed(C + G)
    x = (torch.randn(N, C, H, H, generator=g, dtype=torch.float64) * 0.3 + 0.2)
    if C == 128:
        x = x + 300.0       # |mean| = 1000 sigma: E[x^2]-E[x]^2 in fp32 would lose the variance entirely
    idt = torch.randn(N, C, H, H, generator=g, dtype=torch.float64)
    gamma = (1 + 0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm0 = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    rv0 = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    xr = x.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    outs = []
    for gi in range(G):
        sl = slice(gi * N // G, (gi + 1) * N // G)
        outs.append(F.batch_norm(xr[sl], rm, rv, gamma, beta, True, 0.1, 1e-5))
    pre = torch.cat(outs, 0)
    ref = F.relu(pre + idt)
    dout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gx, gg, gb = torch.autograd.grad(ref, [xr, gamma, beta], dout, retain_graph=True)

    M = N * H * H
    y = nhwc(x)
    f = lambda t: t.detach().float().to(DEV).contiguous()
    d_rm, d_rv = f(rm0), f(rv0)
    mean, rstd, scale, shift = (torch.empty(G * C, device=DEV) for _ in range(4))
    npart = L().io_bn_partial_floats(M, C, G)
    part = torch.empty(npart, device=DEV)
    _lib.check(L().io_bn_stats_finalize(P(y), M, C, G, P(f(gamma)), P(f(beta)), P(d_rm), P(d_rv), 0.1, 1e-5, P(mean),
                                        P(rstd), P(scale), P(shift), P(part), npart, ST()), "bn_stats")
    assert relerr(d_rm, rm) < 1e-5 and relerr(d_rv, rv) < 1e-5
    out = torch.empty_like(y)
    _lib.check(L().io_bn_apply(P(y), M, C, G, 1, P(mean), P(scale), P(shift), P(nhwc(idt)), None, None, None, 1,
                               P(out), ST()), "bn_apply")
    assert relerr(out.permute(0, 3, 1, 2), ref) < (1e-5 if C != 128 else 2e-3)   # x itself is only 24-bit at 300
    # backward (ReLU mask taken from the stored output)
    dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dy, dz = torch.empty_like(y), torch.empty_like(y)
    coef = torch.empty(2 * G * C, device=DEV)
    _lib.check(L().io_bn_bwd(P(nhwc(dout)), P(out), None, None, P(y), M, C, G, P(f(gamma)), P(mean), P(rstd), P(dgam),
                             P(dbet), P(dy), P(dz), P(part), npart, P(coef), ST()), "bn_bwd")
    tol = 2e-5 if C != 128 else 1e-3
    assert relerr(dy.permute(0, 3, 1, 2), gx) < tol
    assert relerr(dgam, gg) < tol and relerr(dbet, gb) < 2e-5
    assert relerr(dz.permute(0, 3, 1, 2), dout * (ref > 0)) < 1e-6
    # no residual: the ReLU mask is recomputed from y with the forward's tables -- must reproduce the stored
    # activation's mask bit for bit
    out2 = torch.empty_like(y)
    _lib.check(L().io_bn_apply(P(y), M, C, G, 1, P(mean), P(scale), P(shift), None, None, None, None, 1, P(out2), ST()),
               "bn_apply")
    ref2 = F.relu(pre)
    gx2, gg2, gb2 = torch.autograd.grad(ref2, [xr, gamma, beta], dout)
    dy_a, dy_b = torch.empty_like(y), torch.empty_like(y)
    dg_a, db_a, dg_b, db_b = (torch.empty(C, device=DEV) for _ in range(4))
    _lib.check(L().io_bn_bwd(P(nhwc(dout)), P(out2), None, None, P(y), M, C, G, P(f(gamma)), P(mean), P(rstd), P(dg_a),
                             P(db_a), P(dy_a), None, P(part), npart, P(coef), ST()), "bn_bwd act")
    _lib.check(L().io_bn_bwd(P(nhwc(dout)), None, P(scale), P(shift), P(y), M, C, G, P(f(gamma)), P(mean), P(rstd),
                             P(dg_b), P(db_b), P(dy_b), None, P(part), npart, P(coef), ST()), "bn_bwd mask-from-y")
    assert torch.equal(dy_a, dy_b) and torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b)
    assert relerr(dy_b.permute(0, 3, 1, 2), gx2) < tol and relerr(dg_b, gg2) < tol


def test_batchnorm_eval_and_downsample_mode():
    g = torch.Generator().manual_seed(1)
    N, C, H = 3, 256, 4
    x = torch.randn(N, C, H, H, generator=g, dtype=torch.float64)
    xd = torch.randn(N, C, H, H, generator=g, dtype=torch.float64)
    par = [torch.rand(C, generator=g, dtype=torch.float64) + 0.5 for _ in range(8)]
    ga, be, rm, rv, ga2, be2, rm2, rv2 = par
    ref = F.relu(F.batch_norm(x, rm, rv, ga, be, False, 0.1, 1e-5) + F.batch_norm(xd, rm2, rv2, ga2, be2, False, 0.1, 1e-5))
    f = lambda t: t.float().to(DEV).contiguous()
    mu, sc, sh, mu2, sc2, sh2 = (torch.empty(C, device=DEV) for _ in range(6))
    _lib.check(L().io_bn_eval_prepare(C, P(f(ga)), P(f(be)), P(f(rm)), P(f(rv)), 1e-5, P(mu), P(sc), P(sh), ST()), "prep")
    _lib.check(L().io_bn_eval_prepare(C, P(f(ga2)), P(f(be2)), P(f(rm2)), P(f(rv2)), 1e-5, P(mu2), P(sc2), P(sh2), ST()),
               "prep")
    out = torch.empty(N, H, H, C, device=DEV)
    _lib.check(L().io_bn_apply(P(nhwc(x)), N * H * H, C, 1, 0, P(mu), P(sc), P(sh), P(nhwc(xd)), P(mu2), P(sc2), P(sh2),
                               1, P(out), ST()), "apply2")
    assert relerr(out.permute(0, 3, 1, 2), ref) < 1e-5


@pytest.mark.parametrize("N,H,W", [(2, 16, 16), (1, 10, 14)])
def test_maxpool(N, H, W):
    g = torch.Generator().manual_seed(2)
    x = F.relu(torch.randn(N, 64, H, W, generator=g, dtype=torch.float64)).requires_grad_(True)   # many exact ties at 0
    ref = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gref = torch.autograd.grad(ref, x, dy)[0]
    Ho, Wo = ref.shape[2:]
    out = torch.empty(N, Ho, Wo, 64, device=DEV)
    idx = torch.empty(N * Ho * Wo * 16, dtype=torch.int32, device=DEV)
    _lib.check(L().io_maxpool_fwd(P(nhwc(x.detach())), N, H, W, 64, P(out), P(idx), ST()), "maxpool")
    assert torch.equal(out.permute(0, 3, 1, 2).cpu(), ref.detach().float())
    dx = torch.empty(N, H, W, 64, device=DEV)
    _lib.check(L().io_maxpool_bwd(P(nhwc(dy)), P(idx), N, H, W, 64, P(dx), ST()), "maxpool_bwd")
    # where the window maximum is an exact tie, only positions with x > 0 matter downstream (ReLU)
    m = (x.detach() > 0)
    assert relerr(dx.permute(0, 3, 1, 2).cpu() * m, gref * m) < 1e-6
    assert abs(float(dx.sum()) - float(dy.sum())) < 1e-3 * float(dy.abs().sum())


@pytest.mark.parametrize("heads", [[2], [2, 3], [4]])
def test_avgpool_fc(heads):
    g = torch.Generator().manual_seed(4)
    N, HW, Cc = 5, 4, 2048
    x = torch.randn(N, Cc, 2, 2, generator=g, dtype=torch.float64).requires_grad_(True)
    ws = [torch.randn(k, Cc, generator=g, dtype=torch.float64).requires_grad_(True) for k in heads]
    bs = [torch.randn(k, generator=g, dtype=torch.float64).requires_grad_(True) for k in heads]
    pooled = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    ref = torch.cat([F.linear(pooled, w, b) for w, b in zip(ws, bs)], 1)
    dl = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    grads = torch.autograd.grad(ref, [x] + ws + bs, dl)
    f = lambda t: t.detach().float().to(DEV).contiguous()
    K0, K1 = heads[0], (heads[1] if len(heads) > 1 else 0)
    dw = [f(w) for w in ws]
    db = [f(b) for b in bs]
    pl = torch.empty(N, Cc, device=DEV)
    lg = torch.empty(N, K0 + K1, device=DEV)
    _lib.check(L().io_avgpool_fc_fwd(P(nhwc(x.detach())), N, HW, Cc, P(dw[0]), P(db[0]), K0, P(dw[1]) if K1 else None,
                                     P(db[1]) if K1 else None, K1, P(pl), P(lg), ST()), "head")
    assert relerr(lg, ref) < 1e-5 and relerr(pl, pooled) < 1e-6
    dx = torch.empty(N, 2, 2, Cc, device=DEV)
    gw = [torch.empty_like(w) for w in dw]
    gb = [torch.empty_like(b) for b in db]
    _lib.check(L().io_avgpool_fc_bwd(P(f(dl)), P(pl), N, HW, Cc, P(dw[0]), K0, P(dw[1]) if K1 else None, K1, None,
                                     P(dx), P(gw[0]), P(gb[0]), P(gw[1]) if K1 else None, P(gb[1]) if K1 else None,
                                     ST()), "head bwd")
    assert relerr(dx.permute(0, 3, 1, 2), grads[0]) < 1e-5
    msk = torch.randn(N, 2, 2, Cc, generator=g).to(DEV)
    dxm = torch.empty_like(dx)
    _lib.check(L().io_avgpool_fc_bwd(P(f(dl)), P(pl), N, HW, Cc, P(dw[0]), K0, P(dw[1]) if K1 else None, K1, P(msk),
                                     P(dxm), P(gw[0]), P(gb[0]), P(gw[1]) if K1 else None, P(gb[1]) if K1 else None,
                                     ST()), "head bwd mask")
    assert torch.equal(dxm, dx * (msk > 0))
    for i in range(len(heads)):
        assert relerr(gw[i], grads[1 + i]) < 1e-5
        assert relerr(gb[i], grads[1 + len(heads) + i]) < 1e-5


def _ref_losses(z, B, Kocc, Kdep, occ_t, dep_t, ov, w_ov, w_di, inv_world):
    z = z.clone().requires_grad_(True)
    l_occ = torch.zeros((), dtype=torch.float64)
    l_dep = torch.zeros((), dtype=torch.float64)
    for d in range(z.shape[0] // B):
        sl = slice(d * B, (d + 1) * B)
        if Kocc:
            l_occ = l_occ + F.binary_cross_entropy(torch.sigmoid(z[sl, :Kocc]), occ_t[sl])
        if Kdep:
            q = F.softmax(z[sl, Kocc:], 1)
            if ov is None:
                l_dep = l_dep + F.cross_entropy(q, dep_t[sl])
            else:
                for msk, wgt in ((ov == 1, w_ov), (ov == 0, w_di)):
                    if int(msk.sum()) > 0:
                        l_dep = l_dep + wgt * F.cross_entropy(q[msk], dep_t[sl][msk])
    tot = (l_occ + l_dep) * inv_world
    return tot, l_occ, l_dep, torch.autograd.grad(tot, z)[0]


@pytest.mark.parametrize("Kocc,Kdep,weighted,B", [(2, 0, False, 7), (2, 3, True, 9), (0, 3, True, 6), (0, 3, False, 6),
                                                  (0, 4, False, 5), (2, 3, True, 300)])
def test_order_loss(Kocc, Kdep, weighted, B):
    g = torch.Generator().manual_seed(B)
    N = 2 * B
    z = torch.randn(N, Kocc + Kdep, generator=g, dtype=torch.float64) * 2
    occ_t = (torch.rand(N, 2, generator=g) < 0.4).double()
    dep_t = torch.randint(0, max(Kdep, 1), (N,), generator=g)
    ov = (torch.rand(B, generator=g) < 0.5).long() if weighted else None
    tot, lo, ld, dz = _ref_losses(z, B, Kocc, Kdep, occ_t, dep_t, ov, 0.1, 0.9, 0.5)
    losses, dl = engine.order_loss(z.float().to(DEV), B, Kocc, Kdep, occ_t.float().to(DEV), dep_t.to(DEV),
                                   ov.to(DEV) if weighted else None, 0.1, 0.9, 0.5, True)
    got = losses.cpu().double()
    assert abs(got[0] - tot) < 2e-6 * max(1, abs(float(tot)))
    assert abs(got[1] - lo) < 2e-6 * max(1, abs(float(lo))) and abs(got[2] - ld) < 2e-6 * max(1, abs(float(ld)))
    assert relerr(dl, dz) < 1e-5


def test_order_loss_subset_empty_and_saturated():
    """all pairs 'distinct' (overlap subset empty -> skipped), and saturated logits (BCE log clamp)."""
    B = 4
    z = torch.tensor([[60.0, -60.0, 1.0, 0.0, -1.0]] * (2 * B), dtype=torch.float64)
    occ_t = torch.tensor([[0.0, 1.0]] * (2 * B), dtype=torch.float64)
    dep_t = torch.zeros(2 * B, dtype=torch.long)
    ov = torch.zeros(B, dtype=torch.long)
    tot, lo, ld, dz = _ref_losses(z.float().double(), B, 2, 3, occ_t, dep_t, ov, 0.1, 0.9, 1.0)
    losses, dl = engine.order_loss(z.float().to(DEV), B, 2, 3, occ_t.float().to(DEV), dep_t.to(DEV), ov.to(DEV), 0.1,
                                   0.9, 1.0, True)
    # fp32 sigmoid saturates to exactly 0/1 at +-60 -> the reference's fp32 path clamps log at -100
    ref32 = F.binary_cross_entropy(torch.sigmoid(z[:B, :2].float()), occ_t[:B].float()) * 2
    assert abs(float(losses[1]) - float(ref32)) < 1e-3
    assert abs(float(losses[2]) - float(ld)) < 1e-5
    assert torch.isfinite(dl).all()


def test_sgd_momentum():
    g = torch.Generator().manual_seed(8)
    n = 4096 + 64
    p0 = torch.randn(n, generator=g)
    par = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([par], lr=1e-2, momentum=0.9, weight_decay=1e-3)
    dp, buf = p0.clone().to(DEV), torch.zeros(n, device=DEV)
    for it in range(3):
        gr = torch.randn(n, generator=g)
        par.grad = gr.clone()
        opt.step()
        engine.sgd_momentum(dp, gr.to(DEV), buf, 1e-2, 0.9, 1e-3)
    assert relerr(dp, par.detach()) < 1e-6


def test_c_abi_from_plain_c_program(tmp_path):
    """The library used the way a non-Python host would: a C program with the HIP runtime only (hipMalloc'd buffers,
    NULL stream) runs a convolution forward, data gradient and filter gradient and checks them against host loops."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers on this box")
    lib = os.path.join(root, "instaorder_amd", "libinstaorder_hip.so")
    exe = str(tmp_path / "abi_gpu")
    cmd = ["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(root, "include"),
           os.path.join(root, "tests", "c", "abi_gpu.c"), "-o", exe, lib, "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-1000:])


def test_unsupported_channel_counts_are_errors_not_crashes():
    """Filter gradient with Cin = 32 (the kernel needs multiples of 64): the planning call and the launch must return
    an error code with a message (this used to divide by zero on the host)."""
    N, H, Cin, Cout = 2, 8, 32, 64
    nb = L().io_conv2d_wgrad_workspace_bytes(N, H, H, Cin, Cout, 3, 3, 1, 1)
    x = torch.zeros(N, H, H, Cin, device=DEV)
    dy = torch.zeros(N, H, H, Cout, device=DEV)
    dw = torch.zeros(Cout, 9, Cin, device=DEV)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=DEV)
    rc = L().io_conv2d_wgrad(P(x), P(dy), P(dw), N, H, H, Cin, Cout, 3, 3, 1, 1, P(ws), int(nb), ST())
    assert rc == -1 and b"multiple of 64" in L().io_last_error_string()
