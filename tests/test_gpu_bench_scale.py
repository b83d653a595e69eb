"""Bench-scale parity: every convolution launch that ``bench.py`` times, at ITS OWN size.

BASELINE.json configs[1] runs InstaOrderNet_o on 256 pairs = 512 samples at 256x256 (the reference's step,
models/supervised_order.py:535-548, through models/backbone/resnet_cls.py:96-116).  The per-kernel tests of
test_gpu_ops.py stop at M = 32 k rows; at the bench batch layer 1 has M = 2.1 M rows and the grid logic is different:
768 / 2304 split-K slices in the filter gradient, the two-block build for 769..1024-tile grids, the XCD remap over many
rounds, BatchNorm tile partials over 16 k tiles, descriptors rebased per tile on multi-GB tensors.  So each distinct
convolution shape of ResNet-50 at N = 512, S = 256 is run here through the C ABI in the very configuration the executor
(csrc/net.hip) launches it in -- forward with the operand transform + statistics epilogue (the fp32 3x3 stride-1 layers
in the Winograd F(4,3) row form: forward, data gradient and filter gradient), data gradient with the fused
BatchNorm-backward epilogue (and, in fp32, the backward operand transform + side output), filter gradient at its real
split count -- in fp32 and bf16, and checked against fp64:

* SAMPLED entries (256 random outputs / filter taps per launch): the operands of each sampled dot product are gathered
  from the very tensors the kernel read (a pure indexing op on the device), moved to the HOST and reduced there in fp64
  with torch over the FULL reduction (K = taps x Cin for the NT launches, M = all 0.03 .. 2.1 M rows for the filter
  gradient);
* the per-channel BatchNorm sums the epilogues produce (mean / rstd of the output; sum dz, sum dz * xhat -> dgamma, dbeta
  and the coefficient tables) against fp64 reductions of the kernel's own output tensor (torch reductions in fp64 on the
  device -- an independent implementation; moving 2 GB per launch to the host would only add minutes).

The streaming kernels the step still runs (BatchNorm backward of bn2 / the stem, the remaining BatchNorm apply forms,
the stem's pooling and its backward, average pool + heads, the fused SGD update on the 23.5 M-float buffer, pair packing,
the order loss) are run at the same batch at the end of the file.

A wrong partition of a reduction at 2304 splits, a tile that reads its neighbour's rows, a partial that lands in the
wrong slot: each shows up here as an O(1) error in some sample or channel.  Tolerances: fp32 2e-5 of the output scale
(the per-kernel bar), bf16 one output rounding (6e-3)."""
import ctypes as C

import numpy as np
import pytest
import torch

from instaorder_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, G, NS = 512, 2, 256          # samples per step (256 pairs, both mask orders), BatchNorm groups, sampled entries

# (name, H_in, Cin, Cout, k, stride, role).  role: which launch configuration the executor uses for this layer --
# c1: plain forward + statistics; data gradient = operand transform (fp32) + residual add + stored-activation mask + the
#     reductions of the previous block's bn3
# c2 / c3: forward through the producer's BatchNorm + ReLU (fp32) + statistics; data gradient masks with relu(bn(y)) > 0
#     recomputed from y, reduces for that BatchNorm and rebuilds its activation; c3 also takes the operand transform
# c2s: strided 3x3 on a stored activation; data gradient as parity classes
# cd: downsample 1x1 (strided except in layer 1); plain data gradient
SHAPES = [
    ("l1.c1a", 64, 64, 64, 1, 1, "c1"), ("l1.c1b", 64, 256, 64, 1, 1, "c1"), ("l1.c2", 64, 64, 64, 3, 1, "c2"),
    ("l1.c3", 64, 64, 256, 1, 1, "c3"), ("l1.cd", 64, 64, 256, 1, 1, "cd"),
    ("l2.c1a", 64, 256, 128, 1, 1, "c1"), ("l2.c2s", 64, 128, 128, 3, 2, "c2s"), ("l2.c3", 32, 128, 512, 1, 1, "c3"),
    ("l2.cd", 64, 256, 512, 1, 2, "cd"), ("l2.c1b", 32, 512, 128, 1, 1, "c1"), ("l2.c2", 32, 128, 128, 3, 1, "c2"),
    ("l3.c1a", 32, 512, 256, 1, 1, "c1"), ("l3.c2s", 32, 256, 256, 3, 2, "c2s"), ("l3.c3", 16, 256, 1024, 1, 1, "c3"),
    ("l3.cd", 32, 512, 1024, 1, 2, "cd"), ("l3.c1b", 16, 1024, 256, 1, 1, "c1"), ("l3.c2", 16, 256, 256, 3, 1, "c2"),
    ("l4.c1a", 16, 1024, 512, 1, 1, "c1"), ("l4.c2s", 16, 512, 512, 3, 2, "c2s"), ("l4.c3", 8, 512, 2048, 1, 1, "c3"),
    ("l4.cd", 16, 1024, 2048, 1, 2, "cd"), ("l4.c1b", 8, 2048, 512, 1, 1, "c1"), ("l4.c2", 8, 512, 512, 3, 1, "c2"),
]
IDS = [s[0] for s in SHAPES]


@pytest.fixture(autouse=True)
def _free():
    yield
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def P(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def ST():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def td_of(dtype):
    return torch.bfloat16 if dtype == "bf16" else torch.float32       # ("fp32", "fp32-exactK": float32)


def gen(seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return g


def randn(shape, g, td, scale=1.0, shift=0.0):
    return (torch.randn(shape, generator=g, device=DEV) * scale + shift).to(td)


def tables(g, Cn):
    """[G][C] fp32 tables in a BatchNorm-like range: mean, rstd, scale (= gamma * rstd), shift"""
    mean = torch.randn(G * Cn, generator=g, device=DEV) * 0.3
    rstd = torch.rand(G * Cn, generator=g, device=DEV) * 1.5 + 0.5
    gamma = torch.rand(Cn, generator=g, device=DEV) + 0.5
    scale = (gamma.repeat(G) * rstd).contiguous()
    shift = torch.randn(G * Cn, generator=g, device=DEV) * 0.3
    return mean, rstd, gamma, scale, shift


def relerr(got, ref):
    got, ref = got.double().cpu().reshape(-1), ref.double().cpu().reshape(-1)
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def out_size(H, k, s):
    pad = k // 2
    return (H + 2 * pad - k) // s + 1, pad


def decode(m, Ho):
    n = m // (Ho * Ho)
    r = m - n * (Ho * Ho)
    ho = r // Ho
    return n, ho, r - ho * Ho


def grp_of(n):
    return n // (N // G)


def host(t):
    return t.double().cpu()


# ---------------------------------------------------------------------------------------------------------------------
# sampled fp64 references (operands gathered on the device, reduced on the host)
# ---------------------------------------------------------------------------------------------------------------------
def ref_forward(x, w, H, Cin, k, s, m, o, xf, bf):
    """sum_{tap, c} X[pix(m, tap)][c] * W[o][tap][c] for the sampled (m, o); X = relu((x - mean) * scale + shift) when the
    operand transform is on (zero in the padding), rounded to the operand type as the kernel rounds it"""
    Ho, pad = out_size(H, k, s)
    n, ho, wo = decode(m, Ho)
    acc = torch.zeros(m.numel(), dtype=torch.float64)
    xv = x.view(-1, Cin)
    for tap in range(k * k):
        r_, s_ = divmod(tap, k)
        hi, wi = ho * s - pad + r_, wo * s - pad + s_
        ok = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < H)
        pix = (n * H + hi.clamp(0, H - 1)) * H + wi.clamp(0, H - 1)
        xa = host(xv[pix])
        if xf is not None:
            mean, scale, shift = (host(t.view(G, Cin)[grp_of(n)]) for t in xf)
            xa = torch.relu((xa - mean) * scale + shift)
            if bf:
                xa = xa.bfloat16().double()
        xa = xa * host(ok)[:, None]
        acc += (xa * host(w[o, tap])).sum(1)
    return acc


def ref_dgrad(src, wt, H, Cin, Cout, k, s, m, c):
    """data gradient of conv(Cin -> Cout, k, stride s, pad k // 2) at the sampled input pixels m / channels c:
    sum_{(r, s'), o} src[n, (h + pad - r) / s, (w + pad - s') / s, o] * W[o][r, s'][c] over the taps that divide.
    src: callable(pixel index tensor of the OUTPUT grid) -> host fp64 [ns, Cout] (so that the operand transform can be
    applied to exactly the gathered rows); wt: device [Cin][k * k][Cout]"""
    Ho, pad = out_size(H, k, s)
    n, h, w_ = decode(m, H)
    acc = torch.zeros(m.numel(), dtype=torch.float64)
    for tap in range(k * k):
        r_, s_ = divmod(tap, k)
        th, tw = h + pad - r_, w_ + pad - s_
        ok = (th % s == 0) & (tw % s == 0)
        yo, xo = th // s, tw // s
        ok = ok & (yo >= 0) & (yo < Ho) & (xo >= 0) & (xo < Ho)
        pix = (n * Ho + yo.clamp(0, Ho - 1)) * Ho + xo.clamp(0, Ho - 1)
        a = src(pix, n) * host(ok)[:, None]
        acc += (a * host(wt[c, tap])).sum(1)
    return acc


def ref_wgrad(x, dy, H, Cin, Cout, k, s, osel, csel, taps=None):
    """dW[o][tap][c] = sum_m dY[m][o] * X[pix(m, tap)][c] for o in osel, c in csel, every tap: the two column subsets are
    gathered on the device, the reduction over ALL m runs on the host in fp64"""
    Ho, pad = out_size(H, k, s)
    dys = host(dy.view(-1, Cout)[:, osel])                                  # [M, no]
    xs = host(x[..., csel])                                                 # [N, H, H, nc]
    xp = torch.zeros(N, H + 2 * pad, H + 2 * pad, csel.numel(), dtype=torch.float64)
    xp[:, pad:pad + H, pad:pad + H] = xs
    out = torch.zeros(osel.numel(), k * k, csel.numel(), dtype=torch.float64)
    for tap in (range(k * k) if taps is None else taps):
        r_, s_ = divmod(tap, k)
        win = xp[:, r_:r_ + (Ho - 1) * s + 1:s, s_:s_ + (Ho - 1) * s + 1:s].reshape(-1, csel.numel())
        out[:, tap] = dys.t() @ win
    return out


def group_sums(dz, y, mean, rstd, Cn):
    """fp64 on the device: per group sum(dz), sum(dz * xhat) over the rows of the kernel's own tensors"""
    M = dz.numel() // Cn
    Mg = M // G
    s1 = torch.zeros(G, Cn, dtype=torch.float64, device=DEV)
    s2 = torch.zeros(G, Cn, dtype=torch.float64, device=DEV)
    step = 1 << 17
    for gi in range(G):
        mu, rs = mean.view(G, Cn)[gi].double(), rstd.view(G, Cn)[gi].double()
        for r0 in range(gi * Mg, (gi + 1) * Mg, step):
            r1 = min(r0 + step, (gi + 1) * Mg)
            d = dz.view(M, Cn)[r0:r1].double()
            s1[gi] += d.sum(0)
            s2[gi] += (d * ((y.view(M, Cn)[r0:r1].double() - mu) * rs)).sum(0)
    return s1, s2


def sample(g, hi, n=NS):
    return torch.randint(0, hi, (n,), generator=g, device=DEV)


def tol(bf, fp32=2e-5, bf16=6e-3):
    return bf16 if bf else fp32


# ---------------------------------------------------------------------------------------------------------------------
# forward: [operand transform +] convolution + statistics epilogue
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_forward_launch_at_bench_size(shape, dtype):
    name, H, Cin, Cout, k, s, role = shape
    bf = dtype == "bf16"
    td = td_of(dtype)
    lib = _lib.lib()
    g = gen(1000 + IDS.index(name))
    Ho, pad = out_size(H, k, s)
    M = N * Ho * Ho
    x = randn((N, H, H, Cin), g, td, 0.8, 0.2)
    w = randn((Cout, k * k, Cin), g, td, 1.0 / np.sqrt(Cin * k * k))
    xf = None
    if role in ("c2", "c3") and not bf:          # fp32: conv2 (stride 1) / conv3 read relu(bn(y)) through the transform
        mean_i, _, _, scale_i, shift_i = tables(g, Cin)
        xf = (mean_i, scale_i, shift_i)
    gamma, beta = torch.rand(Cout, generator=g, device=DEV) + 0.5, torch.randn(Cout, generator=g, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean, rstd, scale, shift = (torch.empty(G * Cout, device=DEV) for _ in range(4))
    nws = lib.io_conv2d_bnstats_workspace_floats(N, H, H, Cout, k, k, s, pad, G)
    ws = torch.empty(nws, device=DEV)
    y = torch.full((N, Ho, Ho, Cout), float("nan"), device=DEV, dtype=td)
    wino = role == "c2" and not bf           # fp32 3x3 stride-1: the executor launches the Winograd F(4,3) row form
    if wino:
        nsc = lib.io_conv2d_wino_scratch_floats(Cin, Cout)
        sc = torch.empty(nsc, device=DEV)
        _lib.check(lib.io_conv2d_fwd_wino(P(x), P(w), P(y), N, H, H, Cin, Cout, G, P(xf[0]), P(xf[1]), P(xf[2]), P(gamma),
                                          P(beta), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(scale), P(shift), P(ws), nws,
                                          P(sc), nsc, ST()), "fwd_wino")
    elif xf is not None:
        _lib.check(lib.io_conv2d_fwd_xf_dt(P(x), P(w), P(y), N, H, H, Cin, Cout, k, k, s, pad, G, P(xf[0]), P(xf[1]), P(xf[2]),
                                           P(gamma), P(beta), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(scale), P(shift),
                                           P(ws), nws, int(bf), ST()), "fwd_xf")
    else:
        _lib.check(lib.io_conv2d_fwd_bnstats_dt(P(x), P(w), P(y), N, H, H, Cin, Cout, k, k, s, pad, G, P(gamma), P(beta),
                                                P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(scale), P(shift), P(ws), nws,
                                                int(bf), 0, ST()), "fwd_stats")
    m, o = sample(g, M), sample(g, Cout)
    ref = ref_forward(x, w, H, Cin, k, s, m, o, xf, bf)
    got = y.view(M, Cout)[m, o]
    assert relerr(got, ref) < (4e-5 if wino else tol(bf)), name          # (the Winograd form re-associates: twice the direct bar)
    # statistics of the output from the epilogue's 128-row tile partials, against fp64 over the kernel's own y
    Mg = M // G
    yv = y.view(G, Mg, Cout)
    mref = torch.stack([yv[gi].double().mean(0) for gi in range(G)])
    vref = torch.stack([yv[gi].double().var(0, unbiased=False) for gi in range(G)])
    # (bf16: the epilogue sums the fp32 accumulators, y holds their bf16 roundings: noise of 4e-3 / sqrt(rows))
    assert relerr(mean.view(G, Cout), mref) < tol(bf, 2e-5, 2e-4), name
    assert relerr(rstd.view(G, Cout), 1.0 / torch.sqrt(vref + 1e-5)) < 1e-4, name
    assert relerr(scale.view(G, Cout), gamma.double() / torch.sqrt(vref + 1e-5)) < 1e-4, name
    # running estimates advanced once per group, in group order, with the unbiased variance
    rme, rve = torch.zeros(Cout, dtype=torch.float64, device=DEV), torch.ones(Cout, dtype=torch.float64, device=DEV)
    for gi in range(G):
        rme = 0.9 * rme + 0.1 * mref[gi]
        rve = 0.9 * rve + 0.1 * vref[gi] * Mg / (Mg - 1)
    assert relerr(rm, rme) < tol(bf, 2e-5, 2e-4) and relerr(rv, rve) < 1e-4, name


XR_SHAPES = [s for s in SHAPES if s[6] == "c1" and s[0] != "l1.c1a"]


@pytest.mark.parametrize("shape", XR_SHAPES, ids=[s[0] for s in XR_SHAPES])
def test_forward_residual_operand_at_bench_size(shape):
    """fp32: conv1 of a block whose predecessor has no downsample branch reads (y3, identity) of that predecessor and
    builds its output relu(bn3(y3) + identity) on the staged operand, writing it out on the way (io_conv2d_fwd_resid).
    Sampled outputs against fp64 over the full reduction, the side output against fp64 element-wise on a sample and
    for completeness (no element left unwritten), statistics against fp64 over the kernel's own y."""
    name, H, Cin, Cout, k, s, role = shape
    lib = _lib.lib()
    g = gen(5000 + IDS.index(name))
    M = N * H * H
    y3 = randn((N, H, H, Cin), g, torch.float32, 0.8, 0.1)
    idt = torch.relu(randn((N, H, H, Cin), g, torch.float32))
    w = randn((Cout, 1, Cin), g, torch.float32, 1.0 / np.sqrt(Cin))
    mean_i, _, _, scale_i, shift_i = tables(g, Cin)
    gamma, beta = torch.rand(Cout, generator=g, device=DEV) + 0.5, torch.randn(Cout, generator=g, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean, rstd, scale, shift = (torch.empty(G * Cout, device=DEV) for _ in range(4))
    nws = lib.io_conv2d_bnstats_workspace_floats(N, H, H, Cout, 1, 1, 1, 0, G)
    ws = torch.empty(nws, device=DEV)
    y = torch.full((N, H, H, Cout), float("nan"), device=DEV)
    out = torch.full((N, H, H, Cin), float("nan"), device=DEV)
    _lib.check(lib.io_conv2d_fwd_resid(P(y3), P(idt), P(w), P(y), P(out), N, H, H, Cin, Cout, G, P(mean_i), P(scale_i),
                                       P(shift_i), P(gamma), P(beta), P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(scale),
                                       P(shift), P(ws), nws, ST()), "fwd_resid")
    m, o = sample(g, M), sample(g, Cout)
    n_s = decode(m, H)[0]
    gi = grp_of(n_s)
    op = torch.relu((host(y3.view(M, Cin)[m]) - host(mean_i.view(G, Cin)[gi])) * host(scale_i.view(G, Cin)[gi])
                    + host(shift_i.view(G, Cin)[gi]) + host(idt.view(M, Cin)[m]))                      # [NS, Cin] fp64
    ref = (op * host(w[o, 0])).sum(1)
    assert relerr(y.view(M, Cout)[m, o], ref) < 2e-5, name
    assert relerr(out.view(M, Cin)[m], op) < 1e-6, name
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0, name
    yv = y.view(G, M // G, Cout)
    mref = torch.stack([yv[i].double().mean(0) for i in range(G)])
    vref = torch.stack([yv[i].double().var(0, unbiased=False) for i in range(G)])
    assert relerr(mean.view(G, Cout), mref) < 2e-5, name
    assert relerr(rstd.view(G, Cout), 1.0 / torch.sqrt(vref + 1e-5)) < 1e-4, name


# ---------------------------------------------------------------------------------------------------------------------
# data gradient in the executor's configuration for this layer
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_data_gradient_launch_at_bench_size(shape, dtype):
    name, H, Cin, Cout, k, s, role = shape
    bf = dtype == "bf16"
    td = td_of(dtype)
    lib = _lib.lib()
    g = gen(2000 + IDS.index(name))
    Ho, pad = out_size(H, k, s)
    Mo, Mi = N * Ho * Ho, N * H * H
    wt = randn((Cin, k * k, Cout), g, td, 1.0 / np.sqrt(Cout * k * k))
    dz_in = randn((N, Ho, Ho, Cout), g, td)               # the gradient arriving at the conv output (or dz of its BN)
    dx = torch.full((N, H, H, Cin), float("nan"), device=DEV, dtype=td)
    m, c = sample(g, Mi), sample(g, Cin)
    n_s = decode(m, H)[0]
    if role in ("c2s", "cd"):
        # plain (strided) data gradient: every lattice class of dx is written
        _lib.check(lib.io_conv2d_dgrad_dt(P(dz_in), P(wt), P(dx), None, None, N, H, H, Cin, Cout, k, k, s, pad, int(bf),
                                          ST()), "dgrad")
        src = lambda pix, n: host(dz_in.view(Mo, Cout)[pix])       # noqa: E731
        ref = ref_dgrad(src, wt, H, Cin, Cout, k, s, m, c)
        assert relerr(dx.view(Mi, Cin)[m, c], ref) < tol(bf), name
        assert bool(torch.isfinite(dx).all()), name
        return
    xb = (role in ("c1", "c3")) and not bf      # fp32: BatchNorm backward's apply pass lives in the operand load
    opt = _lib.DgradFused()
    keep = []
    if xb:
        y_b = randn((N, Ho, Ho, Cout), g, td, 0.7, 0.1)
        coef = (torch.randn(3 * G * Cout, generator=g, device=DEV) * 0.5).contiguous()
        dy_out = torch.full((N, Ho, Ho, Cout), float("nan"), device=DEV, dtype=td)
        opt.xb_y, opt.xb_coef, opt.xb_dy_out = y_b.data_ptr(), coef.data_ptr(), dy_out.data_ptr()
        keep += [y_b, coef, dy_out]
        cf = coef.view(3, G, Cout)

        def src(pix, n):
            gi = grp_of(n)
            return host(cf[0][gi]) * host(dz_in.view(Mo, Cout)[pix]) + host(cf[1][gi]) * host(y_b.view(Mo, Cout)[pix]) + \
                host(cf[2][gi])
    else:
        src = lambda pix, n: host(dz_in.view(Mo, Cout)[pix])       # noqa: E731
    # epilogue: the BatchNorm whose output gradient dx is
    y_a = randn((N, H, H, Cin), g, td, 0.7, 0.1)
    mean_a, rstd_a, gamma_a, scale_a, shift_a = tables(g, Cin)
    nt = lib.io_bn_tile_partial_floats(Mi, Cin, G)
    p1, p2 = torch.empty(nt, device=DEV), torch.empty(nt, device=DEV)
    opt.ep_y, opt.ep_mean, opt.ep_rstd, opt.ep_p1, opt.ep_p2 = (y_a.data_ptr(), mean_a.data_ptr(), rstd_a.data_ptr(),
                                                                p1.data_ptr(), p2.data_ptr())
    aout = base = act = None
    if role == "c1":
        base = randn((N, H, H, Cin), g, td)
        act = randn((N, H, H, Cin), g, td)
        opt.add, opt.relu_mask = base.data_ptr(), act.data_ptr()
    else:
        opt.ep_scale, opt.ep_shift = scale_a.data_ptr(), shift_a.data_ptr()
        if not bf:                               # fp32 never stored relu(bn(y)): the epilogue rebuilds it
            aout = torch.full((N, H, H, Cin), float("nan"), device=DEV, dtype=td)
            opt.ep_act_out = aout.data_ptr()
    wino = role == "c2" and not bf           # fp32 3x3 stride-1: the executor hands the launch its Winograd scratch
    if wino:
        nsc = lib.io_conv2d_wino_scratch_floats(Cout, Cin)
        sc = torch.empty(nsc, device=DEV)
        opt.wino_scratch, opt.wino_scratch_floats = sc.data_ptr(), nsc
        keep.append(sc)
    _lib.check(lib.io_conv2d_dgrad_fused_dt(P(dz_in), P(wt), P(dx), N, H, H, Cin, Cout, k, k, pad, G, C.byref(opt), int(bf),
                                            ST()), "dgrad_fused")
    ref = ref_dgrad(src, wt, H, Cin, Cout, k, 1, m, c)
    gi = grp_of(n_s)
    if role == "c1":
        ref = (ref + host(base.view(Mi, Cin)[m, c])) * host(act.view(Mi, Cin)[m, c] > 0)
        sure = torch.ones(NS, dtype=torch.bool)
    else:
        t = (host(y_a.view(Mi, Cin)[m, c]) - host(mean_a.view(G, Cin)[gi, c])) * host(scale_a.view(G, Cin)[gi, c]) + \
            host(shift_a.view(G, Cin)[gi, c])
        sure = t.abs() > (2e-2 if bf else 1e-5)          # a mask decision within rounding of zero proves nothing
        ref = ref * (t > 0)
        if aout is not None:
            assert relerr(aout.view(Mi, Cin)[m, c], torch.relu(t)) < 2e-5, name
    got = host(dx.view(Mi, Cin)[m, c])
    assert int(sure.sum()) > NS // 2
    assert relerr(got * sure, ref * sure) < (4e-5 if wino else tol(bf)), name
    if xb:      # the side output: dy itself, at sampled entries, and nothing left unwritten
        mo, co = sample(g, Mo), sample(g, Cout)
        want = src(mo, decode(mo, Ho)[0])[torch.arange(NS), co.cpu()]
        assert relerr(dy_out.view(Mo, Cout)[mo, co], want) < 2e-5, name
        assert bool(torch.isfinite(dy_out).all()), name
    # the epilogue's tile partials -> dgamma / dbeta / coefficient tables, against fp64 sums over the kernel's own dx
    coef_a = torch.empty(3 * G * Cin, device=DEV)
    dga, dba = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    _lib.check(lib.io_bn_bwd_coefs_from_tile_partials(P(p1), P(p2), Mi, Cin, G, P(gamma_a), P(mean_a), P(rstd_a), P(dga),
                                                      P(dba), P(coef_a), ST()), "coefs_from_tiles")
    s1, s2 = group_sums(dx, y_a, mean_a, rstd_a, Cin)
    # (bf16: the sums are taken before dx is rounded to bf16, and sums of zero-mean data are themselves of the size of
    # sqrt(rows) roundings: one bf16 rounding is the bar)
    assert relerr(dba, s1.sum(0)) < tol(bf, 5e-5, 6e-3), name
    assert relerr(dga, s2.sum(0)) < tol(bf, 5e-5, 6e-3), name
    Mg = Mi // G
    A = gamma_a.double() * rstd_a.view(G, Cin).double()
    Bc = -A * rstd_a.view(G, Cin).double() * (s2 / Mg)
    Cc = -A * (s1 / Mg) - Bc * mean_a.view(G, Cin).double()
    ca = coef_a.view(3, G, Cin)
    assert relerr(ca[0], A) < 1e-6, name
    # b and c are differences of O(1) quantities scaled by means that are themselves sums with cancellation: hold them
    # to the scale of a (what matters is dy = a dz + b y + c, of the size of a * dz)
    assert float((ca[1].double() - Bc).abs().max()) < tol(bf, 5e-5, 6e-3) * float(A.abs().max()), name
    assert float((ca[2].double() - Cc).abs().max()) < tol(bf, 5e-5, 6e-3) * float(A.abs().max()), name
    del keep


# ---------------------------------------------------------------------------------------------------------------------
# filter gradient at its real split count
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_filter_gradient_launch_at_bench_size(shape, dtype):
    name, H, Cin, Cout, k, s, role = shape
    bf = dtype == "bf16"
    td = td_of(dtype)
    lib = _lib.lib()
    g = gen(3000 + IDS.index(name))
    Ho, pad = out_size(H, k, s)
    x = randn((N, H, H, Cin), g, td, 0.8, 0.2)
    dy = randn((N, Ho, Ho, Cout), g, td)
    nb = lib.io_conv2d_wgrad_workspace_bytes(N, H, H, Cin, Cout, k, k, s, pad)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.full((Cout, k * k, Cin), float("nan"), device=DEV)
    _lib.check(lib.io_conv2d_wgrad_dt(P(x), P(dy), P(dw), N, H, H, Cin, Cout, k, k, s, pad, P(ws), nb, int(bf), int(bf),
                                      ST()), "wgrad")
    assert bool(torch.isfinite(dw).all()), name
    osel = torch.randperm(Cout, generator=g, device=DEV)[:16]
    csel = torch.randperm(Cin, generator=g, device=DEV)[:16]
    ref = ref_wgrad(x, dy, H, Cin, Cout, k, s, osel, csel)                  # [16][taps][16]
    got = dw[osel][:, :, csel]
    # fp32 accumulation over up to 2.1 M rows in both modes (bf16 operands are exact in fp32 products)
    assert relerr(got, ref) < 2e-5, name
    # the whole gradient, cheaply: column sums over (o) of dW equal the filter gradient of the channel-summed dY --
    # one more full-reduction identity that catches a slice added twice or dropped anywhere in the tensor
    dys = dy.view(-1, Cout).double().sum(1)                                  # [M]
    Hp = H + 2 * pad
    xp = torch.zeros(N, Hp, Hp, Cin, dtype=torch.float64, device=DEV)
    tot = torch.empty(k * k, Cin, dtype=torch.float64, device=DEV)
    xp[:, pad:pad + H, pad:pad + H] = x.double()
    for tap in range(k * k):
        r_, s_ = divmod(tap, k)
        win = xp[:, r_:r_ + (Ho - 1) * s + 1:s, s_:s_ + (Ho - 1) * s + 1:s].reshape(-1, Cin)
        tot[tap] = dys @ win
    assert relerr(dw.double().sum(0), tot) < 2e-5, name


# ---------------------------------------------------------------------------------------------------------------------
# the stem: 7x7 / 2 on the packed 5-channel input
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32-exactK", "fp32", "bf16"])
def test_stem_launches_at_bench_size(mode):
    """fp32-exactK is what the fp32 step launches (reduction over the 5 real channels only); the padded form is the public
    Cin = 8 entry point and the bf16 stem."""
    bf = mode == "bf16"
    exact = mode == "fp32-exactK"
    td = td_of(mode)
    lib = _lib.lib()
    g = gen(4000)
    S, Cout, k, s, pad = 256, 64, 7, 2, 3
    Ho = S // 2
    M = N * Ho * Ho
    x = randn((N, S, S, 8), g, td)
    x[..., 5:] = 0
    w = randn((Cout, k * k, 8), g, torch.float32, 1.0 / 15.0)
    w[..., 5:] = 0
    wq = w.to(td)
    gamma, beta = torch.rand(Cout, generator=g, device=DEV) + 0.5, torch.randn(Cout, generator=g, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean, rstd, scale, shift = (torch.empty(G * Cout, device=DEV) for _ in range(4))
    nws = lib.io_conv2d_bnstats_workspace_floats(N, S, S, Cout, k, k, s, pad, G)
    ws = torch.empty(nws, device=DEV)
    y = torch.full((N, Ho, Ho, Cout), float("nan"), device=DEV, dtype=td)
    packed = torch.empty(lib.io_stem_packed_floats(5), device=DEV)
    if exact:
        _lib.check(lib.io_stem_fwd_bnstats_exact(P(x), P(w), P(y), N, S, S, 5, G, P(gamma), P(beta), P(rm), P(rv), 0.1, 1e-5,
                                                 P(mean), P(rstd), P(scale), P(shift), P(ws), nws, P(packed), ST()),
                   "stem fwd exact")
    else:
        _lib.check(lib.io_conv2d_fwd_bnstats_dt(P(x), P(wq), P(y), N, S, S, 8, Cout, k, k, s, pad, G, P(gamma), P(beta),
                                                P(rm), P(rv), 0.1, 1e-5, P(mean), P(rstd), P(scale), P(shift), P(ws), nws,
                                                int(bf), 0, ST()), "stem fwd")
    m, o = sample(g, M), sample(g, Cout)
    assert relerr(y.view(M, Cout)[m, o], ref_forward(x, wq, S, 8, k, s, m, o, None, bf)) < tol(bf)
    yv = y.view(G, M // G, Cout)
    mref = torch.stack([yv[gi].double().mean(0) for gi in range(G)])
    vref = torch.stack([yv[gi].double().var(0, unbiased=False) for gi in range(G)])
    # (zero-mean inputs: the channel means are ~1e-3 of the spread, so hold them to the spread)
    assert float((mean.view(G, Cout).double() - mref).abs().max()) < tol(bf, 2e-5, 2e-4) * float(vref.sqrt().max())
    assert relerr(rstd.view(G, Cout), 1.0 / torch.sqrt(vref + 1e-5)) < 1e-4
    # filter gradient (the network input needs no data gradient)
    dy = randn((N, Ho, Ho, Cout), g, td)
    dw = torch.full((Cout, k * k, 8), float("nan"), device=DEV)
    if exact:
        nb = lib.io_stem_wgrad_exact_workspace_bytes(N, S, S, 5)
        wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
        _lib.check(lib.io_stem_wgrad_exact(P(x), P(dy), P(dw), N, S, S, 5, P(wsb), nb, P(packed), ST()), "stem wgrad exact")
    else:
        nb = lib.io_conv2d_wgrad_workspace_bytes(N, S, S, 8, Cout, k, k, s, pad)
        wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
        _lib.check(lib.io_conv2d_wgrad_dt(P(x), P(dy), P(dw), N, S, S, 8, Cout, k, k, s, pad, P(wsb), nb, int(bf), int(bf),
                                          ST()), "stem wgrad")
    osel = torch.randperm(Cout, generator=g, device=DEV)[:8]
    csel = torch.arange(8, device=DEV)
    taps = [0, 6, 9, 17, 24, 25, 31, 40, 42, 48]          # corners, centre, a few more: each a full reduction over 8.4 M rows
    ref = ref_wgrad(x, dy, S, 8, Cout, k, s, osel, csel, taps)
    assert relerr(dw[osel][:, taps], ref[:, taps]) < 2e-5
    assert float(dw[..., 5:].abs().max()) == 0.0
    if not exact:
        return
    # what the fp32 step launches since the row-persistent stem: bn1's backward evaluated in the staging of the filter
    # gradient.  `dy` above now plays da = d relu(bn1(y)); reference: the BatchNorm-backward formula in fp64 on the device
    # from the kernel's own y / tables (the mask is the sign of fma(y - mean, scale, shift), reproduced exactly in fp64),
    # for the sampled output channels, then the same full-reduction gather as above.
    Mg = M // G
    yq, daq = y.view(G, Mg, Cout), dy.view(G, Mg, Cout)
    mu, rs = mean.view(G, 1, Cout).double(), rstd.view(G, 1, Cout).double()
    s1 = torch.zeros(G, Cout, dtype=torch.float64, device=DEV)
    s2 = torch.zeros(G, Cout, dtype=torch.float64, device=DEV)
    step = 1 << 17
    dzf = lambda gi, r0, r1: torch.where(
        ((yq[gi, r0:r1] - mean.view(G, Cout)[gi]).double() * scale.view(G, Cout)[gi].double() + shift.view(G, Cout)[gi].double()) > 0,
        daq[gi, r0:r1].double(), torch.zeros((), dtype=torch.float64, device=DEV))
    for gi in range(G):
        for r0 in range(0, Mg, step):
            r1 = min(r0 + step, Mg)
            dz = dzf(gi, r0, r1)
            s1[gi] += dz.sum(0)
            s2[gi] += (dz * ((yq[gi, r0:r1].double() - mu[gi]) * rs[gi])).sum(0)
    dysel = torch.empty(M, osel.numel(), dtype=torch.float64, device=DEV)
    for gi in range(G):
        for r0 in range(0, Mg, step):
            r1 = min(r0 + step, Mg)
            xh = (yq[gi, r0:r1].double() - mu[gi]) * rs[gi]
            full = (gamma.double() * rs[gi]) * (dzf(gi, r0, r1) - s1[gi] / Mg - xh * (s2[gi] / Mg))
            dysel[gi * Mg + r0:gi * Mg + r1] = full[:, osel]
    npart = lib.io_bn_partial_floats(M, Cout, G)
    part = torch.empty(npart, device=DEV)
    coef = torch.full((3 * G * Cout,), float("nan"), device=DEV)
    dgam, dbet = (torch.full((Cout,), float("nan"), device=DEV) for _ in range(2))
    dw2 = torch.full((Cout, k * k, 8), float("nan"), device=DEV)
    _lib.check(lib.io_stem_wgrad_exact_bn(P(x), P(dy), P(y), P(dw2), N, S, S, 5, G, P(gamma), P(mean), P(rstd), P(scale),
                                          P(shift), P(dgam), P(dbet), P(coef), P(part), npart, P(wsb), nb, P(packed), ST()),
               "stem wgrad exact + bn1 backward")
    ref2 = ref_wgrad(x, dysel, S, 8, osel.numel(), k, s, torch.arange(osel.numel(), device=DEV), csel, taps)
    assert relerr(dw2[osel][:, taps], ref2[:, taps]) < 1e-4
    assert float(dw2[..., 5:].abs().max()) == 0.0
    assert relerr(dbet, s1.sum(0)) < 1e-5 and relerr(dgam, s2.sum(0)) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# the streaming kernels the step still runs, at the bench batch (grid strides that are not powers of two, 8 M rows)
# ---------------------------------------------------------------------------------------------------------------------
def _bn_ref_tables(y, Cn, gamma):
    """fp64 batch statistics per group of a [M, C] tensor -> the fp32 tables the kernels take"""
    M = y.numel() // Cn
    yv = y.view(G, M // G, Cn)
    mean = torch.stack([yv[i].double().mean(0) for i in range(G)])
    var = torch.stack([yv[i].double().var(0, unbiased=False) for i in range(G)])
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    return (mean.float().reshape(-1).contiguous(), rstd.float().reshape(-1).contiguous(),
            (gamma.double() * rstd).float().reshape(-1).contiguous())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("H,Cn,masked", [(128, 64, True), (64, 64, False), (32, 128, False), (8, 512, False)],
                         ids=["stem-bn1", "l1-bn2", "l2-bn2", "l4-bn2"])
def test_bn_backward_passes_at_bench_size(H, Cn, masked, dtype):
    """io_bn_bwd_dt -- reduction, finalize and apply pass -- on the BatchNorm shapes that still run it in the fp32 step
    (bn2 of every block; the stem's bn1 with its ReLU mask recomputed from y): dgamma / dbeta against fp64 sums over the
    whole tensor, dy against fp64 at every element of a row sample."""
    bf = dtype == "bf16"
    td = td_of(dtype)
    lib = _lib.lib()
    g = gen(6000 + H + Cn)
    M = N * H * H
    y = randn((M, Cn), g, td, 0.7, 0.2)
    dz = randn((M, Cn), g, td)
    gamma, beta = torch.rand(Cn, generator=g, device=DEV) + 0.5, torch.randn(Cn, generator=g, device=DEV) * 0.3
    mean, rstd, scale = _bn_ref_tables(y, Cn, gamma)
    shift = beta.repeat(G).contiguous()
    npart = lib.io_bn_partial_floats(M, Cn, G)
    part, coef = torch.empty(npart, device=DEV), torch.empty(2 * G * Cn, device=DEV)
    dy = torch.full((M, Cn), float("nan"), device=DEV, dtype=td)
    dga, dbe = torch.empty(Cn, device=DEV), torch.empty(Cn, device=DEV)
    _lib.check(lib.io_bn_bwd_dt(P(dz), None, P(scale) if masked else None, P(shift) if masked else None, P(y), M, Cn, G,
                                P(gamma), P(mean), P(rstd), P(dga), P(dbe), P(dy), None, P(part), npart, P(coef), int(bf),
                                ST()), "bn_bwd")
    Mg = M // G
    s1 = torch.zeros(G, Cn, dtype=torch.float64, device=DEV)
    s2 = torch.zeros(G, Cn, dtype=torch.float64, device=DEV)
    # mask decisions within fp32 rounding of zero may fall either way (the kernel evaluates bn(y) as one fp32 fma): what
    # such elements could move the sums by is allowed on top of the tolerance (8.4 M rows: a few dozen per channel)
    k1 = torch.zeros(Cn, dtype=torch.float64, device=DEV)
    k2 = torch.zeros(Cn, dtype=torch.float64, device=DEV)
    step = 1 << 18
    for gi in range(G):
        mu, rs = mean.view(G, Cn)[gi].double(), rstd.view(G, Cn)[gi].double()
        sc, sh = scale.view(G, Cn)[gi].double(), shift.view(G, Cn)[gi].double()
        for r0 in range(gi * Mg, (gi + 1) * Mg, step):
            r1 = min(r0 + step, (gi + 1) * Mg)
            d, yy = dz[r0:r1].double(), y[r0:r1].double()
            xh = (yy - mu) * rs
            if masked:
                t = (yy - mu) * sc + sh
                edge = t.abs() < (2e-2 if bf else 2e-6)
                k1 += (d.abs() * edge).sum(0)
                k2 += ((d * xh).abs() * edge).sum(0)
                d = d * (t > 0)
            s1[gi] += d.sum(0)
            s2[gi] += (d * xh).sum(0)
    e1 = (dbe.double() - s1.sum(0)).abs() - k1
    e2 = (dga.double() - s2.sum(0)).abs() - k2
    assert float(e1.max()) < 5e-5 * float(s1.sum(0).abs().max()), (H, Cn, float(e1.max()))
    assert float(e2.max()) < 5e-5 * float(s2.sum(0).abs().max()), (H, Cn, float(e2.max()))
    rows = sample(g, M, 2048)
    gi = rows // Mg
    mu, rs = host(mean.view(G, Cn)[gi]), host(rstd.view(G, Cn)[gi])
    d, yy = host(dz[rows]), host(y[rows])
    if masked:
        t = (yy - mu) * host(scale.view(G, Cn)[gi]) + host(shift.view(G, Cn)[gi])
        sure = t.abs() > (2e-2 if bf else 1e-5)
        d = d * (t > 0)
    else:
        sure = torch.ones_like(d, dtype=torch.bool)
    xh = (yy - mu) * rs
    ref = host(gamma) * rs * (d - host(s1[gi]) / Mg - xh * host(s2[gi]) / Mg)
    assert relerr(host(dy[rows]) * sure, ref * sure) < tol(bf, 2e-5, 6e-3), (H, Cn)
    assert bool(torch.isfinite(dy).all())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_residual_bn_apply_at_bench_size(dtype):
    """io_bn_apply_dt in its three forms at the sizes the step still runs them: relu(bn(y)) of a strided block's conv1
    (2.1 M x 128), the last block's relu(bn3(y3) + identity) (32 k x 2048) and, for the bf16 step, a layer-1 block output
    (2.1 M x 256) -- exact element-wise against the same expression in torch (fp32 fma chain is the kernel's; compare at
    one output rounding)."""
    bf = dtype == "bf16"
    td = td_of(dtype)
    lib = _lib.lib()
    g = gen(6100)
    for H, Cn, with_id in ((64, 128, False), (8, 2048, True), (64, 256, True)):
        M = N * H * H
        y = randn((M, Cn), g, td, 0.8, 0.1)
        idt = torch.relu(randn((M, Cn), g, td)) if with_id else None
        mean, _, _, scale, shift = tables(g, Cn)
        out = torch.full((M, Cn), float("nan"), device=DEV, dtype=td)
        _lib.check(lib.io_bn_apply_dt(P(y), M, Cn, G, 1, P(mean), P(scale), P(shift), P(idt), None, None, None, 1, P(out),
                                      int(bf), ST()), "bn_apply")
        Mg = M // G
        for gi in range(G):
            sl = slice(gi * Mg, (gi + 1) * Mg)
            ref = (y[sl].double() - mean.view(G, Cn)[gi].double()) * scale.view(G, Cn)[gi].double() + \
                shift.view(G, Cn)[gi].double()
            if with_id:
                ref = ref + idt[sl].double()
            ref = torch.relu(ref)
            err = float((out[sl].double() - ref).abs().max() / ref.abs().max())
            assert err < tol(bf, 1e-6, 4e-3), (H, Cn, gi, err)
        del y, idt, out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pooling_and_heads_at_bench_size(dtype):
    """The stem's max-pool over relu(bn1(y)) evaluated on the fly (512 x 128 x 128 x 64 -> 64 x 64) with its backward, and
    the average pool + heads (512 x 8 x 8 x 2048) forward / backward."""
    import torch.nn.functional as F
    bf = dtype == "bf16"
    td = td_of(dtype)
    lib = _lib.lib()
    g = gen(6200)
    H, Cn = 128, 64
    y = randn((N, H, H, Cn), g, td, 0.8, 0.1)
    mean, _, _, scale, shift = tables(g, Cn)
    Ho = H // 2
    out = torch.empty(N, Ho, Ho, Cn, device=DEV, dtype=td)
    idx = torch.empty(N * Ho * Ho * Cn // 4, device=DEV, dtype=torch.int32)
    _lib.check(lib.io_maxpool_fwd_xf_dt(P(y), N, H, H, Cn, P(out), P(idx), G, P(mean), P(scale), P(shift), int(bf), ST()),
               "maxpool_xf")
    per = N // G
    for n0 in range(0, N, 64):              # reference in slices (fp32 torch on the device; max-pooling is exact)
        gi = n0 // per
        a = torch.relu((y[n0:n0 + 64].float() - mean.view(G, Cn)[gi]) * scale.view(G, Cn)[gi] + shift.view(G, Cn)[gi])
        if bf:
            a = a.bfloat16().float()
        ref = F.max_pool2d(a.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
        err = float((out[n0:n0 + 64].float() - ref).abs().max())
        assert err < (2e-2 if bf else 1e-5), (n0, err)           # (fma vs mul + add in the affine part)
    dyp = randn((N, Ho, Ho, Cn), g, td)
    dx = torch.full((N, H, H, Cn), float("nan"), device=DEV, dtype=td)
    _lib.check(lib.io_maxpool_bwd_dt(P(dyp), P(idx), N, H, H, Cn, P(dx), int(bf), ST()), "maxpool_bwd")
    # every pooled gradient is routed to exactly one input element: the sums agree exactly in fp64 per (sample, channel)
    assert relerr(dx.double().sum((1, 2)), dyp.double().sum((1, 2))) < tol(bf, 1e-6, 2e-3)
    assert bool(torch.isfinite(dx).all())
    # heads
    HW, Cc, K0 = 64, 2048, 2
    x = torch.relu(randn((N, 8, 8, Cc), g, td))
    w0 = randn((K0, Cc), g, torch.float32, 0.02)
    b0 = randn((K0,), g, torch.float32, 0.1)
    pooled, logits = torch.empty(N, Cc, device=DEV), torch.empty(N, K0, device=DEV)
    _lib.check(lib.io_avgpool_fc_fwd_dt(P(x), N, HW, Cc, P(w0), P(b0), K0, None, None, 0, P(pooled), P(logits), int(bf), ST()),
               "avgpool_fc")
    pref = x.double().mean((1, 2))
    assert relerr(pooled, pref) < 1e-6
    assert relerr(logits, pref @ w0.double().t() + b0.double()) < 1e-5
    dl = randn((N, K0), g, torch.float32)
    dxh = torch.empty(N, 8, 8, Cc, device=DEV, dtype=td)
    dw, db = torch.empty(K0, Cc, device=DEV), torch.empty(K0, device=DEV)
    _lib.check(lib.io_avgpool_fc_bwd_dt(P(dl), P(pooled), N, HW, Cc, P(w0), K0, None, 0, P(x), P(dxh), P(dw), P(db), None, None,
                                        int(bf), ST()), "avgpool_fc_bwd")
    dref = ((dl.double() @ w0.double()) / HW)[:, None, None, :] * (x > 0)
    assert relerr(dxh, dref) < tol(bf, 1e-5, 4e-3)
    assert relerr(dw, dl.double().t() @ pref) < 1e-5 and relerr(db, dl.double().sum(0)) < 1e-5


def test_sgd_pack_and_loss_at_bench_size():
    """The rest of the step at its real size: the fused SGD update on the 23.5 M-float flat buffer (torch.optim.SGD's
    formula, three steps), the pair packing of 256 pairs at 256 x 256 (both mask orders into [512, 256, 256, 8]), the
    BCE order loss and its gradient for 512 rows."""
    import torch.nn.functional as F
    from instaorder_amd import engine
    g = gen(6300)
    n = 23527872
    p0 = torch.randn(n, generator=g, device=DEV)
    par = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([par], lr=1e-3, momentum=0.9, weight_decay=1e-4)
    dp, buf = p0.clone(), torch.zeros(n, device=DEV)
    for it in range(3):
        gr = torch.randn(n, generator=g, device=DEV)
        par.grad = gr.clone()
        opt.step()
        engine.sgd_momentum(dp, gr, buf, 1e-3, 0.9, 1e-4)
    assert float((dp - par.detach()).abs().max()) < 1e-6 * float(par.detach().abs().max())
    B, S = 256, 256
    rgb = torch.randn(B, 3, S, S, generator=g, device=DEV)
    m1 = (torch.rand(B, 1, S, S, generator=g, device=DEV) > 0.6).float()
    m2 = (torch.rand(B, 1, S, S, generator=g, device=DEV) > 0.6).float()
    x8 = engine.pack_pair_directions(rgb, m1, m2)
    assert x8.shape == (2 * B, S, S, 8)
    ref1 = torch.cat([m1, m2, rgb], 1).permute(0, 2, 3, 1)
    ref2 = torch.cat([m2, m1, rgb], 1).permute(0, 2, 3, 1)
    assert torch.equal(x8[:B, ..., :5], ref1) and torch.equal(x8[B:, ..., :5], ref2)
    assert float(x8[..., 5:].abs().max()) == 0.0
    z = torch.randn(2 * B, 2, generator=g, device=DEV)
    occ = (torch.rand(2 * B, 2, generator=g, device=DEV) > 0.7).float()
    losses, dlog = engine.order_loss(z, B, 2, 0, occ_target=occ)
    z64 = z.double().cpu().requires_grad_(True)
    ref = F.binary_cross_entropy(torch.sigmoid(z64[:B]), occ[:B].double().cpu()) + \
        F.binary_cross_entropy(torch.sigmoid(z64[B:]), occ[B:].double().cpu())
    ref.backward()
    assert abs(float(losses[0]) - float(ref)) < 1e-5 * float(ref)
    assert relerr(dlog, z64.grad) < 1e-5
