"""The Winograd F(2, 3) row form of the 3x3 stride-1 convolutions (conv_nt_kernel<..., WINO>, csrc/conv_igemm.hip) against
fp64 torch on the same seeded inputs -- resnet_cls.py:23-26 ``conv3x3`` as conv2 of a Bottleneck uses it (:88, stride 1):
forward (plain, with the input transform relu(bn(x)) on the staged operand, with the statistics epilogue) and data
gradient with the fused BatchNorm-backward epilogue (mask recomputed from y, tile partial sums, activation side output).

Tolerance: the form re-associates the arithmetic (4 products of sums instead of 6 products per filter row and output
pair), so it is held to 4e-5 of the output scale against fp64 -- twice the 2e-5 of the direct kernels, far inside the
north-star 1e-3 -- and to the DIRECT HIP kernel's result at the same bar."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from instaorder_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 4e-5

_KEEP = []


def P(t):
    if t is None:
        return C.c_void_p(0)
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:-64]
    return C.c_void_p(t.data_ptr())


def ST():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().float().to(DEV)


def krsc(w):
    return w.permute(0, 2, 3, 1).contiguous().float().to(DEV)


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# N, H, W, Cin, Cout, G     (N*H*W/G a multiple of 128, W even)
CASES = [(2, 8, 8, 64, 64, 1), (4, 16, 16, 64, 128, 2), (2, 32, 32, 128, 64, 2), (2, 8, 16, 256, 256, 1),
         (8, 8, 8, 512, 512, 2), (2, 64, 64, 64, 64, 1), (4, 4, 8, 128, 192, 1)]
FWD_ONLY = [(2, 64, 64, 32, 64, 1), (2, 16, 32, 96, 64, 2)]      # input channel counts the data gradient does not take
# widths that are not powers of two -- what 'orig'-mode inference and the 384 x 384 MiDaS decoder (96 / 48 / 24 / 12 wide maps)
# send here: F(4,3) WITHOUT the HALO staging (256 % W != 0), tiles that straddle image rows and samples, the F(2,3)
# fallback where 4 does not divide W or 256 does not divide M, the filter gradient at 8 | W but 16 !| W
NONPOW2 = [(4, 8, 24, 64, 64, 1), (2, 24, 48, 128, 64, 1), (4, 8, 40, 64, 128, 1), (16, 6, 20, 64, 64, 1),
           (8, 8, 12, 64, 64, 2), (4, 16, 6, 64, 64, 1), (2, 12, 96, 64, 64, 1)]


def _scratch(Cin, Cout):
    n = _lib.lib().io_conv2d_wino_scratch_floats(Cin, Cout)
    return torch.empty(n, device=DEV), n


@pytest.mark.parametrize("case", CASES + FWD_ONLY + NONPOW2)
def test_wino_forward_plain_xf_stats(case):
    N, H, W, Cin, Cout, G = case
    g = torch.Generator().manual_seed(11 + Cin + W)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64).float().double()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) / (3.0 * Cin ** 0.5)).float().double()
    lib = _lib.lib()
    sc, nsc = _scratch(Cin, Cout)
    xd, wd = nhwc(x), krsc(w).view(Cout, 9, Cin).contiguous()
    # plain
    ref = F.conv2d(x, w, padding=1)
    y = torch.full((N, H, W, Cout), float("nan"), device=DEV)
    _lib.check(lib.io_conv2d_fwd_wino(P(xd), P(wd), P(y), N, H, W, Cin, Cout, G, None, None, None, None, None, None, None,
                                      0.1, 1e-5, None, None, None, None, None, 0, P(sc), nsc, ST()), "wino plain")
    assert relerr(y.permute(0, 3, 1, 2), ref) < TOL
    yd = torch.empty_like(y)
    _lib.check(lib.io_conv2d_fwd_dt(P(xd), P(wd), P(yd), N, H, W, Cin, Cout, 3, 3, 1, 1, 0, 0, ST()), "direct")
    assert relerr(y.permute(0, 3, 1, 2), yd.permute(0, 3, 1, 2).double().cpu()) < TOL
    # input transform + statistics
    scale = torch.randn(G, Cin, generator=g, dtype=torch.float64) * 0.7 + 0.3
    shift = torch.randn(G, Cin, generator=g, dtype=torch.float64) * 0.5 + 0.4
    mean = torch.randn(G, Cin, generator=g, dtype=torch.float64) * 0.3
    per = N // G
    v4 = lambda t, gi: t[gi].view(1, -1, 1, 1)      # noqa: E731
    xa = torch.cat([F.relu((x[gi * per:(gi + 1) * per] - v4(mean.float().double(), gi)) * v4(scale.float().double(), gi)
                           + v4(shift.float().double(), gi)) for gi in range(G)])
    ref2 = F.conv2d(xa, w, padding=1)
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean2, rstd, sc2, sh2 = (torch.empty(G * Cout, device=DEV) for _ in range(4))
    nws = lib.io_conv2d_bnstats_workspace_floats(N, H, W, Cout, 3, 3, 1, 1, G)
    ws = torch.empty(nws, device=DEV)
    y2 = torch.full((N, H, W, Cout), float("nan"), device=DEV)
    _lib.check(lib.io_conv2d_fwd_wino(P(xd), P(wd), P(y2), N, H, W, Cin, Cout, G, P(mean.float().to(DEV)),
                                      P(scale.float().to(DEV)), P(shift.float().to(DEV)), P(gamma.to(DEV)),
                                      P(beta.to(DEV)), P(rm), P(rv), 0.1, 1e-5, P(mean2), P(rstd), P(sc2), P(sh2), P(ws),
                                      nws, P(sc), nsc, ST()), "wino xf+stats")
    assert relerr(y2.permute(0, 3, 1, 2), ref2) < TOL
    yk = y2.permute(0, 3, 1, 2).double().cpu()          # statistics of the kernel's own output
    mref = torch.stack([yk[gi * per:(gi + 1) * per].mean((0, 2, 3)) for gi in range(G)])
    vref = torch.stack([yk[gi * per:(gi + 1) * per].var((0, 2, 3), unbiased=False) for gi in range(G)])
    assert relerr(mean2.view(G, Cout), mref) < 2e-5
    assert relerr(rstd.view(G, Cout), 1.0 / torch.sqrt(vref + 1e-5)) < 2e-5


@pytest.mark.parametrize("case", CASES + NONPOW2)
def test_wino_dgrad_with_bn_backward_epilogue(case):
    """dz = dgrad(conv3x3)(dy) * [relu(bn_a(y_a)) > 0] with the per-tile sums of dz and dz * xhat and relu(bn_a(y_a)) as a
    side output: the launch the executor makes for conv2 of a Bottleneck (net.hip dgrad_then_bn), Winograd form against the
    direct form and fp64."""
    N, H, W, Cin, Cout, G = case
    g = torch.Generator().manual_seed(5 + Cout + H)
    lib = _lib.lib()
    M = N * H * W
    per = N // G
    rt = lambda t: t.float().double()      # noqa: E731
    dy = rt(torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64))
    w = rt(torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) / (3.0 * Cout ** 0.5))
    y_a = rt(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64))
    gam = rt(torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5)
    bet = rt(torch.randn(Cin, generator=g, dtype=torch.float64) * 0.3)
    mean = torch.stack([y_a[gi * per:(gi + 1) * per].mean((0, 2, 3)) for gi in range(G)])
    var = torch.stack([y_a[gi * per:(gi + 1) * per].var((0, 2, 3), unbiased=False) for gi in range(G)])
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    mean_f, rstd_f = mean.float(), rstd.float()
    scale_f = (gam.float() * rstd_f)
    shift_f = bet.float().expand(G, Cin).contiguous()
    grp = torch.arange(N) // per
    v = lambda t: t[grp].view(N, Cin, 1, 1)          # noqa: E731
    # the mask exactly as the kernel forms it: sign of fma(y - mean, scale, shift) in fp32
    t32 = torch.addcmul(v(shift_f), (y_a.float() - v(mean_f)), v(scale_f))
    mask = (t32 > 0)
    a_in = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    da = torch.autograd.grad(F.conv2d(a_in, w, padding=1), a_in, dy)[0]
    dz_ref = da * mask
    wt = krsc(w).view(Cout, 9, Cin).permute(2, 1, 0).contiguous()
    nt = lib.io_bn_tile_partial_floats(M, Cin, G)
    outs = []
    for use_wino in (True, False):
        p1, p2 = torch.zeros(nt, device=DEV), torch.zeros(nt, device=DEV)
        dx = torch.full((N, H, W, Cin), float("nan"), device=DEV)
        aout = torch.full((N, H, W, Cin), float("nan"), device=DEV)
        opt = _lib.DgradFused()
        ya_d = nhwc(y_a)
        tabs = [t.to(DEV).contiguous() for t in (mean_f, rstd_f, scale_f, shift_f)]
        opt.ep_y, opt.ep_mean, opt.ep_rstd, opt.ep_scale, opt.ep_shift = (ya_d.data_ptr(), tabs[0].data_ptr(),
                                                                          tabs[1].data_ptr(), tabs[2].data_ptr(),
                                                                          tabs[3].data_ptr())
        opt.ep_p1, opt.ep_p2, opt.ep_act_out = p1.data_ptr(), p2.data_ptr(), aout.data_ptr()
        sc, nsc = _scratch(Cout, Cin)
        if use_wino:
            opt.wino_scratch, opt.wino_scratch_floats = sc.data_ptr(), nsc
        _lib.check(lib.io_conv2d_dgrad_fused_dt(P(nhwc(dy)), P(wt), P(dx), N, H, W, Cin, Cout, 3, 3, 1, G, C.byref(opt), 0,
                                                ST()), "dgrad_fused wino=%s" % use_wino)
        torch.cuda.synchronize()
        outs.append((dx, aout, p1, p2))
        assert relerr(dx.permute(0, 3, 1, 2), dz_ref) < TOL, use_wino
        assert relerr(aout.permute(0, 3, 1, 2), F.relu(t32.double())) < 2e-6
        # tile partials against fp64 sums of the kernel's own dz
        dzk = dx.double().cpu().view(M // 128, 128, Cin)
        xhat = ((y_a - v(mean)) * v(rstd)).permute(0, 2, 3, 1).reshape(M // 128, 128, Cin)
        assert relerr(p1[:M // 128 * Cin].view(M // 128, Cin), dzk.sum(1)) < 3e-5
        assert relerr(p2[:M // 128 * Cin].view(M // 128, Cin), (dzk * xhat).sum(1)) < 3e-5
        del ya_d, tabs, sc
    assert relerr(outs[0][0], outs[1][0].double().cpu()) < TOL
    assert torch.equal(outs[0][1], outs[1][1])          # the side output does not depend on the product form


@pytest.mark.parametrize("case", CASES + NONPOW2 + [(16, 32, 32, 64, 64, 1), (64, 16, 16, 128, 128, 1)])
def test_wino_wgrad(case):
    """Filter gradient of the 3x3 stride-1 convolution in the Winograd row form (conv_wgrad_wino_kernel: the launcher takes it
    for fp32, 8 | W, 64 | N*H*W) against fp64 autograd; the last cases split the reduction over many blocks."""
    N, H, W, Cin, Cout, G = case
    g = torch.Generator().manual_seed(3 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64).float().double()
    dy = torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64).float().double()
    w0 = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(x, w0, padding=1), w0, dy)[0]
    lib = _lib.lib()
    nb = lib.io_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, 3, 3, 1, 1)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.full((Cout, 9, Cin), float("nan"), device=DEV)
    _lib.check(lib.io_conv2d_wgrad_dt(P(nhwc(x)), P(nhwc(dy)), P(dw), N, H, W, Cin, Cout, 3, 3, 1, 1, P(ws), nb, 0, 0, ST()),
               "wgrad")
    got = dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    assert relerr(got, ref) < TOL
