"""The persistent LDS-DMA bf16 kernels of round 5 behind the storage-typed C entry points, each against fp64 torch on the same
bf16-rounded inputs AND against the kernel it replaces (io_set_bf16_p256: 3 = the new kernels for every eligible shape, 0 = none),
with the route asserted (io_debug_last_nt_route / io_debug_last_wgrad_route):
  csrc/conv_p256.hip   forward convolutions (resnet_cls.py:23-31: 1x1 / 3x3, stride 1 / 2) plain, with the statistics epilogue and
                       with the folded-BatchNorm inference epilogue; data gradients with the fused BatchNorm-backward epilogue
                       (residual gradient, ReLU mask read -- as a tensor or as bits -- and recomputed);
  csrc/conv_halo3.hip  conv_halo3_kernel (3x3 stride 1 -> 64 channels from a halo image, filters in registers), stem_halo_kernel (the
                       7x7 / 2 stem), conv_wgrad_halo3_kernel (3x3 filter gradients, 64 .. 256 channels), stem_wgrad_halo_kernel
                       (the stem's filter gradient, plain and with bn1's backward folded in).
Bar: bf16 output rounding (2^-8 relative per element) on top of fp32 accumulation -- 1e-2 of the output scale; filter gradients
(fp32 outputs of exact bf16 products) 2e-5."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from instaorder_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = 1
TOL = 1e-2
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


def bf(x):
    """fp64 tensor -> (bf16 device tensor NHWC / as is, the fp64 values it holds)"""
    d = x.to(torch.bfloat16)
    return d.to(DEV).contiguous(), d.double()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _restore():
    prev = _lib.lib().io_get_bf16_p256()
    yield
    _lib.lib().io_set_bf16_p256(prev)


# N, H, W, Cin, Cout, R, stride     (N * Ho * Wo a multiple of 256)
# The cases of csrc/conv_halo3.hip (3x3 stride 1, 64 output channels from 64 -- filters in registers -- or 128, 64- / 32-wide maps).  Mode 3 takes the persistent kernels
# for every eligible shape (mode 1 leaves launches that fill < 80 % of a round of 256 blocks to the 128-row kernel); the
# route is asserted (io_debug_last_nt_route: 1 = conv_p256, 2 = conv_halo3).  One tile .. several rounds, ragged last round.
HALO_CASES = [(1, 64, 64, 64, 64, 3, 1), (3, 32, 32, 128, 64, 3, 1), (16, 64, 64, 128, 64, 3, 1), (67, 32, 32, 64, 64, 3, 1),
              (40, 64, 64, 64, 64, 3, 1), (2, 128, 64, 64, 64, 3, 1), (5, 64, 32, 64, 64, 3, 1)]


def _halo_shape(N, H, W, Cin, Cout, R, stride=1):
    return R == 3 and stride == 1 and Cin in (64, 128) and Cout == 64 and W in (32, 64) and (H * W) % 256 == 0


CASES = [(2, 16, 16, 64, 256, 1, 1), (4, 8, 8, 128, 128, 3, 1), (2, 32, 32, 64, 128, 3, 1), (8, 16, 16, 128, 256, 1, 2),
         (8, 16, 16, 64, 128, 3, 2), (4, 8, 8, 512, 512, 1, 1), (32, 64, 64, 64, 256, 1, 1), (1, 16, 16, 256, 384, 3, 1)]


def _inputs(case, seed):
    N, H, W, Cin, Cout, R, stride = case
    g = torch.Generator().manual_seed(seed)
    x, xr = bf(nhwc(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)))
    w, wr = bf((torch.randn(Cout, Cin, R, R, generator=g, dtype=torch.float64) / (R * Cin ** 0.5)).permute(0, 2, 3, 1).contiguous())
    return x, xr.permute(0, 3, 1, 2), w.view(Cout, R * R, Cin), wr.permute(0, 3, 1, 2)


@pytest.mark.parametrize("case", CASES + HALO_CASES)
def test_p256_forward_plain_stats_bias(case):
    N, H, W, Cin, Cout, R, stride = case
    pad = R // 2
    lib = _lib.lib()
    halo = _halo_shape(*case)
    MODES = (3, 2, 0) if halo else (3, 0)
    x, xr, w, wr = _inputs(case, 3 + Cin + H)
    ref = F.conv2d(xr, wr, stride=stride, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    outs = []
    for on in MODES:
        lib.io_set_bf16_p256(on)
        y = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.io_conv2d_fwd_dt(P(x), P(w), P(y), N, H, W, Cin, Cout, R, R, stride, pad, BF, BF, ST()), "fwd")
        if on != 2:
            assert lib.io_debug_last_nt_route() == ((2 if halo else 1) if on == 3 else 0), on
        assert relerr(y.permute(0, 3, 1, 2), ref) < TOL, on
        outs.append(y)
    assert relerr(outs[0], outs[-1].double().cpu()) < TOL
    # statistics epilogue (G = 2 when the rows per group stay whole 128-row tiles)
    M = N * Ho * Wo
    G = 2 if (N % 2 == 0 and (M // 2) % 128 == 0) else 1
    gen = torch.Generator().manual_seed(9)
    gamma, beta = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    for on in MODES:
        lib.io_set_bf16_p256(on)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        mean, rstd, sc, sh = (torch.empty(G * Cout, device=DEV) for _ in range(4))
        nws = lib.io_conv2d_bnstats_workspace_floats(N, H, W, Cout, R, R, stride, pad, G)
        ws = torch.empty(nws, device=DEV)
        y2 = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.io_conv2d_fwd_bnstats_dt(P(x), P(w), P(y2), N, H, W, Cin, Cout, R, R, stride, pad, G, P(gamma.to(DEV)),
                                                P(beta.to(DEV)), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(sc), P(sh),
                                                P(ws), nws, BF, 0, ST()), "fwd+stats")
        assert relerr(y2.permute(0, 3, 1, 2), ref) < TOL
        # the statistics are those of the fp32 accumulators (before the bf16 rounding of y): against the fp64 convolution
        per = N // G
        mref = torch.stack([ref[gi * per:(gi + 1) * per].mean((0, 2, 3)) for gi in range(G)])
        vref = torch.stack([ref[gi * per:(gi + 1) * per].var((0, 2, 3), unbiased=False) for gi in range(G)])
        assert float((mean.view(G, Cout).double().cpu() - mref).abs().max()) < 2e-4 * float(ref.abs().max())
        assert relerr(rstd.view(G, Cout), 1.0 / torch.sqrt(vref + 1e-5)) < 2e-4
    # inference epilogue: relu(conv + bias + add)
    bias = torch.randn(Cout, generator=gen)
    addd, addr = bf(torch.randn(N, Ho, Wo, Cout, generator=gen, dtype=torch.float64))
    ref3 = F.relu(ref + bias.double().view(1, -1, 1, 1) + addr.permute(0, 3, 1, 2))
    for on in MODES:
        lib.io_set_bf16_p256(on)
        y3 = torch.full((N, Ho, Wo, Cout), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.io_conv2d_fwd_bias_dt(P(x), P(w), P(y3), N, H, W, Cin, Cout, R, R, stride, pad, P(bias.to(DEV)), P(addd), 1,
                                             BF, 0, ST()), "fwd+bias")
        assert relerr(y3.permute(0, 3, 1, 2), ref3) < TOL, on


# (the halo kernel's data gradients: conv2 of layers 1-2, mask recomputed from y)
HALO_DG = [(2, 64, 64, 64, 64, 3), (6, 32, 32, 64, 64, 3), (40, 64, 64, 64, 64, 3), (34, 32, 32, 64, 128, 3)]
DG_CASES = [(2, 16, 16, 256, 64, 1), (4, 8, 8, 128, 128, 3), (2, 32, 32, 128, 64, 3), (4, 16, 16, 512, 128, 1), (32, 32, 32, 256, 64, 1)]


@pytest.mark.parametrize("case", DG_CASES + HALO_DG)
@pytest.mark.parametrize("form", ["recompute_mask", "read_mask_add"])
def test_p256_dgrad_with_bn_backward_epilogue(case, form):
    """dz = (dgrad(conv)(dy) [+ add]) * mask with the per-tile sums of dz and dz * xhat -- the launches the executor makes for
    conv3 / conv2 (mask recomputed from y) and conv1 (identity-path gradient added, mask read) of a Bottleneck."""
    N, H, W, Cin, Cout, R = case       # the data gradient has Cin output channels (>= 128) and reduces over Cout
    pad = R // 2
    lib = _lib.lib()
    halo = _halo_shape(N, H, W, Cout, Cin, R) and form == "recompute_mask"     # (output channels of the gradient: Cin)
    if case in HALO_DG and form != "recompute_mask":
        pytest.skip("the halo kernel recomputes the mask (conv2's data gradient); a mask tensor goes to the other kernels")
    g = torch.Generator().manual_seed(7 + Cin + H)
    M = N * H * W
    G = 2 if (N % 2 == 0 and (M // 2) % 256 == 0) else 1
    per = N // G
    dy, dyr = bf(torch.randn(N, H, W, Cout, generator=g, dtype=torch.float64))
    w64 = torch.randn(Cout, Cin, R, R, generator=g, dtype=torch.float64) / (R * Cout ** 0.5)
    wt, wtr = bf(w64.permute(1, 2, 3, 0).reshape(Cin, R * R, Cout).contiguous())          # W^T [Cin][taps][Cout]
    wr = wtr.view(Cin, R, R, Cout).permute(3, 0, 1, 2)                                   # back to OIHW, as rounded
    ya, yar = bf(torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64))
    yn = yar.permute(0, 3, 1, 2)
    mean = torch.stack([yn[gi * per:(gi + 1) * per].mean((0, 2, 3)) for gi in range(G)])
    var = torch.stack([yn[gi * per:(gi + 1) * per].var((0, 2, 3), unbiased=False) for gi in range(G)])
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    gam = torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5
    bet = torch.randn(Cin, generator=g, dtype=torch.float64) * 0.3
    mean_f, rstd_f = mean.float(), rstd.float()
    scale_f = gam.float() * rstd_f
    shift_f = bet.float().expand(G, Cin).contiguous()
    grp = torch.arange(N) // per
    v = lambda t: t[grp].view(N, Cin, 1, 1)          # noqa: E731
    a_in = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    da = torch.autograd.grad(F.conv2d(a_in, wr, padding=pad), a_in, dyr.permute(0, 3, 1, 2))[0]
    addd = maskd = None
    if form == "recompute_mask":
        t32 = torch.addcmul(v(shift_f), (yn.float() - v(mean_f)), v(scale_f))
        dz_ref = da * (t32 > 0)
    else:
        addd, addr = bf(torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64))
        maskd, maskr = bf(F.relu(torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64)))
        dz_ref = (da + addr.permute(0, 3, 1, 2)) * (maskr.permute(0, 3, 1, 2) > 0)
    nt = lib.io_bn_tile_partial_floats(M, Cin, G)
    outs = []
    for on in ((3, 2, 0) if halo else (3, 0)):
        lib.io_set_bf16_p256(on)
        p1, p2 = torch.zeros(nt, device=DEV), torch.zeros(nt, device=DEV)
        dx = torch.full((N, H, W, Cin), float("nan"), device=DEV, dtype=torch.bfloat16)
        opt = _lib.DgradFused()
        tabs = [t.to(DEV).contiguous() for t in (mean_f, rstd_f, scale_f, shift_f)]
        opt.ep_y, opt.ep_mean, opt.ep_rstd = ya.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr()
        if form == "recompute_mask":
            opt.ep_scale, opt.ep_shift = tabs[2].data_ptr(), tabs[3].data_ptr()
        else:
            opt.add, opt.relu_mask = addd.data_ptr(), maskd.data_ptr()
        opt.ep_p1, opt.ep_p2 = p1.data_ptr(), p2.data_ptr()
        _lib.check(lib.io_conv2d_dgrad_fused_dt(P(dy), P(wt), P(dx), N, H, W, Cin, Cout, R, R, pad, G, C.byref(opt), BF, ST()),
                   "dgrad_fused p256=%d" % on)
        torch.cuda.synchronize()
        if not halo:
            assert lib.io_debug_last_nt_route() == (1 if on == 3 else 0), on
        elif on != 2:
            assert lib.io_debug_last_nt_route() == {3: 2, 0: 0}[on], on
        assert relerr(dx.permute(0, 3, 1, 2), dz_ref) < TOL, on
        # tile partials: sums of the kernel's fp32 dz (before rounding) -- against fp64 sums of the reference dz
        dzk = dz_ref.permute(0, 2, 3, 1).reshape(M // 128, 128, Cin)
        xhat = ((yn - v(mean)) * v(rstd)).permute(0, 2, 3, 1).reshape(M // 128, 128, Cin)
        s1, s2 = dzk.sum(1), (dzk * xhat).sum(1)
        assert float((p1[:M // 128 * Cin].view(M // 128, Cin).double().cpu() - s1).abs().max()) < 2e-3 * float(s1.abs().max())
        assert float((p2[:M // 128 * Cin].view(M // 128, Cin).double().cpu() - s2).abs().max()) < 2e-3 * float(s2.abs().max())
        outs.append(dx)
        del tabs
    assert relerr(outs[0], outs[-1].double().cpu()) < TOL


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("p256", [3, 0])
def test_relu_mask_as_bits(dtype, p256):
    """The ReLU mask of a block output kept as one bit per element: io_bn_apply_bits_dt writes bit c % 32 of word (m Cc + c) / 32
    = (relu(bn(y) + identity) > 0) next to the activation, and the 1x1 data gradient that completes d(out) (conv1 of the next
    Bottleneck, resnet_cls.py:99, 114) on the 256-row kernel masks with it -- bit-identical to the same launch reading the
    activation tensor, in both epilogue forms (plain add + mask, and with the BatchNorm-backward sums of the previous block's
    bn3); the 128-row kernel takes the tensor either way."""
    if dtype == "fp32" and p256 == 3:
        pytest.skip("the 256-row kernel is a bf16 kernel")
    lib = _lib.lib()
    lib.io_set_bf16_p256(p256)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dt = BF if dtype == "bf16" else 0
    N, H, W, Cc, Cr = 4, 16, 16, 256, 64         # out / d(out): Cc channels; conv1: Cc -> Cr
    M = N * H * W
    g = torch.Generator().manual_seed(21)
    y3 = torch.randn(N, H, W, Cc, generator=g).to(tdt).to(DEV)
    idt = torch.randn(N, H, W, Cc, generator=g).to(tdt).to(DEV)
    mean, scale, shift = (torch.randn(Cc, generator=g) * 0.2).to(DEV), (torch.rand(Cc, generator=g) + 0.5).to(DEV), \
        (torch.randn(Cc, generator=g) * 0.3).to(DEV)
    out = torch.empty_like(y3)
    bits = torch.zeros(M * Cc // 32, dtype=torch.int32, device=DEV)
    _lib.check(lib.io_bn_apply_bits_dt(P(y3), M, Cc, 1, 0, P(mean), P(scale), P(shift), P(idt), None, None, None, P(out), P(bits),
                                       dt, ST()), "bn_apply_bits")
    out2 = torch.empty_like(y3)
    _lib.check(lib.io_bn_apply_dt(P(y3), M, Cc, 1, 0, P(mean), P(scale), P(shift), P(idt), None, None, None, 1, P(out2), dt, ST()),
               "bn_apply")
    assert torch.equal(out, out2)
    want = (out.float() > 0).view(M, Cc // 32, 32).to(torch.int64)
    words = (want << torch.arange(32, device=DEV)).sum(-1)
    got = bits.to(torch.int64) & 0xffffffff
    assert torch.equal(got.view(M, Cc // 32), words)
    assert 0.2 < float(want.float().mean()) < 0.8
    # the data gradient of conv1 (Cr -> Cc in the gradient direction) with the identity-path gradient added
    dy = torch.randn(N, H, W, Cr, generator=g).to(tdt).to(DEV)
    wt = (torch.randn(Cc, 1, Cr, generator=g) / Cr ** 0.5).to(tdt).to(DEV)
    addt = torch.randn(N, H, W, Cc, generator=g).to(tdt).to(DEV)
    yprev = torch.randn(N, H, W, Cc, generator=g).to(tdt).to(DEV)
    mu, rs = (torch.randn(Cc, generator=g) * 0.1).to(DEV), (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    nt = lib.io_bn_tile_partial_floats(M, Cc, 1)
    for with_sums in (False, True):
        res = []
        for use_bits in (True, False):
            dx = torch.full((N, H, W, Cc), float("nan"), device=DEV, dtype=tdt)
            p1, p2 = torch.zeros(nt, device=DEV), torch.zeros(nt, device=DEV)
            opt = _lib.DgradFused()
            opt.add = addt.data_ptr()
            opt.relu_mask = out.data_ptr()
            if use_bits:                       # the bit form rides along; the 256-row kernel then reads it INSTEAD of the tensor
                opt.relu_maskbits = bits.data_ptr()
            if with_sums:
                opt.ep_y, opt.ep_mean, opt.ep_rstd = yprev.data_ptr(), mu.data_ptr(), rs.data_ptr()
                opt.ep_p1, opt.ep_p2 = p1.data_ptr(), p2.data_ptr()
            _lib.check(lib.io_conv2d_dgrad_fused_dt(P(dy), P(wt), P(dx), N, H, W, Cc, Cr, 1, 1, 0, 1, C.byref(opt), dt, ST()),
                       "dgrad bits=%s sums=%s" % (use_bits, with_sums))
            torch.cuda.synchronize()
            res.append((dx, p1, p2))
        assert torch.isfinite(res[0][0].float()).all()
        assert torch.equal(res[0][0], res[1][0]), (with_sums, "dx")
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]), (with_sums, "sums")
        zero = (out.float() <= 0)
        assert float(res[0][0].float()[zero].abs().max()) == 0.0


@pytest.mark.parametrize("N,H", [(2, 256), (3, 256), (1, 512)])
def test_stem_halo_forward_plain_stats_bias(N, H):
    """The bf16 stem (7x7 stride 2 on the packed 8-channel input, resnet_cls.py:155) on 128-wide output rows: stem_halo_kernel of
    csrc/conv_halo3.hip (patch of two output rows in LDS by pixel parity, filters in registers) against fp64 torch on the same
    bf16-rounded operands and against the 64-wide implicit-GEMM kernel it replaces -- plain, with the statistics epilogue, and
    with the folded-BatchNorm inference epilogue.  (H = 512: Wo = 256, not this kernel's shape -- both modes take the old one.)"""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(40 + N)
    x5 = torch.randn(N, 5, H, H, generator=g, dtype=torch.float64)
    x8, x8r = bf(nhwc(torch.cat([x5, torch.zeros(N, 3, H, H, dtype=torch.float64)], 1)))
    w, wr = bf((torch.randn(64, 8, 7, 7, generator=g, dtype=torch.float64) * 0.05).permute(0, 2, 3, 1).contiguous())   # [64][7][7][8]
    ref = F.conv2d(x8r.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), stride=2, padding=3)
    Ho = ref.shape[2]
    want = 3 if Ho == 128 else 0
    wd = w.view(64, 49, 8)
    outs = []
    for on in (3, 0):
        lib.io_set_bf16_p256(on)
        y = torch.full((N, Ho, Ho, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.io_conv2d_fwd_dt(P(x8), P(wd), P(y), N, H, H, 8, 64, 7, 7, 2, 3, BF, BF, ST()), "stem fwd")
        assert lib.io_debug_last_nt_route() == (want if on == 3 else 0), on
        assert relerr(y.permute(0, 3, 1, 2), ref) < TOL, on
        outs.append(y)
    assert relerr(outs[0], outs[1].double().cpu()) < TOL
    M = N * Ho * Ho
    G = 2 if N % 2 == 0 else 1
    gen = torch.Generator().manual_seed(9)
    gamma, beta = torch.rand(64, generator=gen) + 0.5, torch.randn(64, generator=gen)
    for on in (3, 0):
        lib.io_set_bf16_p256(on)
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        mean, rstd, sc, sh = (torch.empty(G * 64, device=DEV) for _ in range(4))
        nws = lib.io_conv2d_bnstats_workspace_floats(N, H, H, 64, 7, 7, 2, 3, G)
        ws = torch.empty(nws, device=DEV)
        y2 = torch.full((N, Ho, Ho, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.io_conv2d_fwd_bnstats_dt(P(x8), P(wd), P(y2), N, H, H, 8, 64, 7, 7, 2, 3, G, P(gamma.to(DEV)),
                                                P(beta.to(DEV)), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(sc), P(sh),
                                                P(ws), nws, BF, 0, ST()), "stem fwd+stats")
        assert relerr(y2.permute(0, 3, 1, 2), ref) < TOL
        per = N // G
        mref = torch.stack([ref[gi * per:(gi + 1) * per].mean((0, 2, 3)) for gi in range(G)])
        vref = torch.stack([ref[gi * per:(gi + 1) * per].var((0, 2, 3), unbiased=False) for gi in range(G)])
        assert float((mean.view(G, 64).double().cpu() - mref).abs().max()) < 2e-4 * float(ref.abs().max())
        assert relerr(rstd.view(G, 64), 1.0 / torch.sqrt(vref + 1e-5)) < 2e-4
    bias = torch.randn(64, generator=gen)
    ref3 = F.relu(ref + bias.double().view(1, -1, 1, 1))
    for on in (3, 0):
        lib.io_set_bf16_p256(on)
        y3 = torch.full((N, Ho, Ho, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.io_conv2d_fwd_bias_dt(P(x8), P(wd), P(y3), N, H, H, 8, 64, 7, 7, 2, 3, P(bias.to(DEV)), None, 1,
                                             BF, 0, ST()), "stem fwd+bias")
        assert relerr(y3.permute(0, 3, 1, 2), ref3) < TOL, on


@pytest.mark.parametrize("N,G", [(2, 2), (3, 1), (4, 2)])
def test_stem_wgrad_halo_plain_and_with_bn1_backward(N, G):
    """The bf16 stem's filter gradient on stem_wgrad_halo_kernel (csrc/conv_halo3.hip: patch of two output rows in LDS, dy rows
    through registers, transposing fragment reads, one partial per persistent block): (1) the plain gradient through
    io_conv2d_wgrad_dt against fp64 on the same bf16 operands and against the kernel it replaces; (2) with bn1's backward folded
    into the staging of dy (io_stem_wgrad_bn_bf16; resnet_cls.py:155-158: relu(bn1(conv1(x)))) against the BatchNorm-backward
    formula in fp64 on the tables and the ReLU mask the kernels see, and against the unfused pair io_bn_bwd_dt +
    io_conv2d_wgrad_dt."""
    lib = _lib.lib()
    H = 256
    g = torch.Generator().manual_seed(70 + N)
    x5 = torch.randn(N, 5, H, H, generator=g, dtype=torch.float64)
    x8, x8r = bf(nhwc(torch.cat([x5, torch.zeros(N, 3, H, H, dtype=torch.float64)], 1)))
    w, wr = bf((torch.randn(64, 8, 7, 7, generator=g, dtype=torch.float64) * 0.05).permute(0, 2, 3, 1).contiguous())
    xr = x8r.permute(0, 3, 1, 2)
    Ho = H // 2
    M = N * Ho * Ho
    dy, dyr = bf(torch.randn(N, Ho, Ho, 64, generator=g, dtype=torch.float64))

    def wgrad_ref(dy_nchw):
        wq = wr.permute(0, 3, 1, 2).clone().requires_grad_(True)
        return torch.autograd.grad(F.conv2d(xr, wq, stride=2, padding=3), wq, dy_nchw)[0]        # [64][8][7][7]

    gref = wgrad_ref(dyr.permute(0, 3, 1, 2))
    nb = lib.io_conv2d_wgrad_workspace_bytes(N, H, H, 8, 64, 7, 7, 2, 3)
    assert nb >= lib.io_stem_wgrad_bf16_workspace_bytes()
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    res = []
    for on in (3, 0):
        lib.io_set_bf16_p256(on)
        dw = torch.full((64, 49, 8), float("nan"), device=DEV)
        _lib.check(lib.io_conv2d_wgrad_dt(P(x8), P(dy), P(dw), N, H, H, 8, 64, 7, 7, 2, 3, P(ws), nb, BF, BF, ST()), "stem wgrad")
        assert lib.io_debug_last_wgrad_route() == (1 if on == 3 else 0)
        assert relerr(dw.view(64, 7, 7, 8).permute(0, 3, 1, 2), gref) < 2e-5, on     # fp32 accumulation of exact bf16 products
        res.append(dw)
    # ---- with bn1's backward: y and its tables from the forward kernel, da = gradient of relu(bn1(y))
    lib.io_set_bf16_p256(3)
    gen = torch.Generator().manual_seed(5)
    gamma, beta = torch.rand(64, generator=gen) + 0.5, torch.randn(64, generator=gen) * 0.3
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    mean, rstd, sc, sh = (torch.empty(G * 64, device=DEV) for _ in range(4))
    nws = lib.io_conv2d_bnstats_workspace_floats(N, H, H, 64, 7, 7, 2, 3, G)
    wsf = torch.empty(nws, device=DEV)
    y = torch.empty(N, Ho, Ho, 64, device=DEV, dtype=torch.bfloat16)
    _lib.check(lib.io_conv2d_fwd_bnstats_dt(P(x8), P(w.view(64, 49, 8)), P(y), N, H, H, 8, 64, 7, 7, 2, 3, G, P(gamma.to(DEV)),
                                            P(beta.to(DEV)), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(sc), P(sh), P(wsf), nws,
                                            BF, 0, ST()), "stem fwd + stats")
    da, dar = bf(torch.randn(N, Ho, Ho, 64, generator=g, dtype=torch.float64))
    per = N // G
    grp = (torch.arange(N) // per)
    tv = lambda t: t.view(G, 64).cpu()[grp].view(N, 1, 1, 64)          # noqa: E731
    yf = y.float().cpu()
    act = torch.addcmul(tv(sh), yf - tv(mean), tv(sc))                  # fma(y - mean, scale, shift) in fp32, as bn_apply
    dz = dar * (act > 0)
    xhat = (yf.double() - tv(mean).double()) * tv(rstd).double()
    dy_ref = torch.empty_like(dz)
    dgam_ref, dbet_ref = torch.zeros(64, dtype=torch.float64), torch.zeros(64, dtype=torch.float64)
    for gi in range(G):
        sl = slice(gi * per, (gi + 1) * per)
        s1 = dz[sl].mean((0, 1, 2))
        s2 = (dz[sl] * xhat[sl]).mean((0, 1, 2))
        dy_ref[sl] = gamma.double() * rstd.view(G, 64)[gi].double().cpu() * (dz[sl] - s1 - xhat[sl] * s2)
        dgam_ref += (dz[sl] * xhat[sl]).sum((0, 1, 2))
        dbet_ref += dz[sl].sum((0, 1, 2))
    gw = wgrad_ref(dy_ref.permute(0, 3, 1, 2))
    npart = lib.io_bn_partial_floats(M, 64, G)
    part = torch.empty(npart, device=DEV)
    coef = torch.full((3 * G * 64,), float("nan"), device=DEV)
    dgam, dbet = torch.full((64,), float("nan"), device=DEV), torch.full((64,), float("nan"), device=DEV)
    dw2 = torch.full((64, 49, 8), float("nan"), device=DEV)
    _lib.check(lib.io_stem_wgrad_bn_bf16(P(x8), P(da), P(y), P(dw2), N, H, H, G, P(gamma.to(DEV)), P(mean), P(rstd), P(sc), P(sh),
                                         P(dgam), P(dbet), P(coef), P(part), npart, P(ws), nb, ST()), "stem wgrad + bn1")
    got2 = dw2.view(64, 7, 7, 8).permute(0, 3, 1, 2)
    # (dy is rounded to bf16 on its way into the matrix pipe, as the unfused pair rounds it when it writes the tensor; BatchNorm
    # backward subtracts two projections, so the result is small against its terms)
    assert relerr(got2, gw) < 2e-3
    assert relerr(dgam, dgam_ref) < 1e-3 and relerr(dbet, dbet_ref) < 1e-3
    # the unfused pair on the same operands
    dyt = torch.empty(N, Ho, Ho, 64, device=DEV, dtype=torch.bfloat16)
    dg3, db3 = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    _lib.check(lib.io_bn_bwd_dt(P(da), None, P(sc), P(sh), P(y), M, 64, G, P(gamma.to(DEV)), P(mean), P(rstd), P(dg3), P(db3),
                                P(dyt), None, P(part), npart, P(torch.empty(3 * G * 64, device=DEV)), BF, ST()), "bn_bwd")
    dw3 = torch.full((64, 49, 8), float("nan"), device=DEV)
    lib.io_set_bf16_p256(0)
    _lib.check(lib.io_conv2d_wgrad_dt(P(x8), P(dyt), P(dw3), N, H, H, 8, 64, 7, 7, 2, 3, P(ws), nb, BF, BF, ST()), "stem wgrad (unfused)")
    assert relerr(dw2, dw3.double().cpu()) < 2e-3
    assert relerr(dgam, dg3.double().cpu()) < 1e-5 and relerr(dbet, db3.double().cpu()) < 1e-5


@pytest.mark.parametrize("N,H,W,Cc", [(1, 64, 64, 64), (5, 64, 64, 64), (3, 128, 64, 64), (70, 64, 64, 64), (2, 32, 32, 128),
                                       (37, 32, 32, 128), (4, 16, 16, 256), (50, 16, 16, 256), (3, 64, 32, 128), (2, 8, 16, 256)])
def test_wgrad_halo3(N, H, W, Cc):
    """Filter gradient of the bf16 3x3 stride-1 convolutions with 64 / 128 / 256 channels on 64- / 32- / 16-wide maps (conv2 of
    the layer-1 .. layer-3 Bottlenecks, resnet_cls.py:88) on conv_wgrad_halo3_kernel (csrc/conv_halo3.hip: halo image of a
    64-channel slice of x + the dy rows in LDS once per 128-pixel tile, all nine taps from them through transposing fragment
    reads, one partial per persistent block, (Co / 64) x (Ci / 64) slice pairs as sub-problems) against fp64 on the same bf16
    operands and against conv_wgrad_bf16_tr_kernel.  One tile per block .. several rounds, ragged last round."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(90 + N + H + Cc)
    x, xr = bf(torch.randn(N, H, W, Cc, generator=g, dtype=torch.float64))
    dy, dyr = bf(torch.randn(N, H, W, Cc, generator=g, dtype=torch.float64))
    wq = torch.zeros(Cc, Cc, 3, 3, dtype=torch.float64, requires_grad=True)
    gref = torch.autograd.grad(F.conv2d(xr.permute(0, 3, 1, 2), wq, padding=1), wq, dyr.permute(0, 3, 1, 2))[0]
    nb = lib.io_conv2d_wgrad_workspace_bytes(N, H, W, Cc, Cc, 3, 3, 1, 1)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    res = []
    for on in (3, 0):
        lib.io_set_bf16_p256(on)
        dw = torch.full((Cc, 9, Cc), float("nan"), device=DEV)
        _lib.check(lib.io_conv2d_wgrad_dt(P(x), P(dy), P(dw), N, H, W, Cc, Cc, 3, 3, 1, 1, P(ws), nb, BF, BF, ST()), "wgrad")
        assert lib.io_debug_last_wgrad_route() == (2 if on == 3 else 0)
        assert relerr(dw.view(Cc, 3, 3, Cc).permute(0, 3, 1, 2), gref) < 2e-5, on
        res.append(dw)
    assert relerr(res[0], res[1].double().cpu()) < 2e-5
