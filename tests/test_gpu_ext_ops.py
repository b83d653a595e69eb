"""Kernels of the MiDaS branch against fp64 torch ops: grouped 3x3 convolution of ResNeXt
(forward / data gradient / filter gradient through the block-diagonal window form), bilinear x2 (both align_corners
modes, forward + adjoint), bias / ReLU / add / column sums, and the one-channel head."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from instaorder_amd import _lib
from test_gpu_ops import L, P, ST, nhwc, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(t):
    return t.float().to(DEV).contiguous()


@pytest.mark.parametrize("C,cg,H,stride", [(256, 8, 12, 1), (512, 16, 10, 2), (1024, 32, 6, 1), (2048, 64, 6, 2),
                                           (128, 4, 9, 1)])
def test_grouped_conv3x3(C, cg, H, stride):
    N, groups = 3, C // cg
    g = torch.Generator().manual_seed(C + cg)
    x = torch.randn(N, C, H, H, generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn(C, cg, 3, 3, generator=g, dtype=torch.float64) / np.sqrt(9 * cg)).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=1, groups=groups)
    Ho = y.shape[2]
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    wd = dev(w.detach().reshape(C, cg, 9))
    wc = torch.empty(C, 9, 64, device=DEV)
    wtc = torch.empty(C, 9, 64, device=DEV)
    _lib.check(L().io_gconv_pack(P(wd), C, cg, 9, P(wc), P(wtc), 0, ST()), "pack")
    xd = dev(x.detach().permute(0, 2, 3, 1))
    yd = torch.empty(N, Ho, Ho, C, device=DEV)
    _lib.check(L().io_gconv2d_fwd(P(xd), P(wc), P(yd), N, H, H, C, 3, 3, stride, 1, 0, ST()), "fwd")
    assert relerr(yd.permute(0, 3, 1, 2), y.detach()) < 2e-6
    dyd = dev(dy.permute(0, 2, 3, 1))
    dxd = torch.empty(N, H, H, C, device=DEV)
    _lib.check(L().io_gconv2d_dgrad(P(dyd), P(wtc), P(dxd), N, H, H, C, 3, 3, stride, 1, 0, ST()), "dgrad")
    assert relerr(dxd.permute(0, 3, 1, 2), gx) < 2e-6
    nb = L().io_gconv2d_wgrad_workspace_bytes(N, H, H, C, 3, 3, stride, 1)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dwc = torch.empty(C, 9, 64, device=DEV)
    _lib.check(L().io_gconv2d_wgrad(P(xd), P(dyd), P(dwc), N, H, H, C, 3, 3, stride, 1, P(ws), nb, 0, ST()), "wgrad")
    dw = torch.empty(C, cg, 9, device=DEV)
    _lib.check(L().io_gconv_unpack_grad(P(dwc), C, cg, 9, P(dw), ST()), "unpack")
    assert relerr(dw.view(C, cg, 3, 3), gw) < 2e-6


@pytest.mark.parametrize("align", [0, 1])
@pytest.mark.parametrize("N,H,W,C", [(2, 6, 6, 256), (1, 3, 5, 8), (2, 1, 4, 64)])
def test_upsample2x_bilinear(N, H, W, C, align):
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=bool(align))
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    gx, = torch.autograd.grad(y, [x], dy)
    xd = dev(x.detach().permute(0, 2, 3, 1))
    out = torch.empty(N, 2 * H, 2 * W, C, device=DEV)
    _lib.check(L().io_upsample2x_bilinear_fwd(P(xd), N, H, W, C, align, P(out), 0, ST()), "up fwd")
    assert relerr(out.permute(0, 3, 1, 2), y.detach()) < 1e-6
    dx = torch.empty(N, H, W, C, device=DEV)
    _lib.check(L().io_upsample2x_bilinear_bwd(P(dev(dy.permute(0, 2, 3, 1))), N, H, W, C, align, P(dx), 0, ST()), "up bwd")
    assert relerr(dx.permute(0, 3, 1, 2), gx) < 1e-6


def test_bias_relu_add_colsum():
    M, C = 3000, 128
    g = torch.Generator().manual_seed(1)
    x, b = torch.randn(M, C, generator=g), torch.randn(C, generator=g)
    xd, bd = dev(x), dev(b)
    out = torch.empty(M, C, device=DEV)
    _lib.check(L().io_bias_act(P(xd), P(bd), M, C, 1, P(out), 0, ST()), "bias_act")
    assert torch.equal(out.cpu(), torch.relu(x + b))
    _lib.check(L().io_bias_act(P(xd), None, M, C, 1, P(out), 0, ST()), "relu")
    assert torch.equal(out.cpu(), torch.relu(x))
    dy = torch.randn(M, C, generator=g)
    dx = torch.empty(M, C, device=DEV)
    _lib.check(L().io_relu_bwd(P(dev(dy)), P(out), M * C, P(dx), 0, ST()), "relu_bwd")
    assert torch.equal(dx.cpu(), dy * (x > 0))
    _lib.check(L().io_add(P(xd), P(dev(dy)), M * C, P(dx), 0, ST()), "add")
    assert torch.equal(dx.cpu(), x + dy)
    for Cc in (32, 128, 256):
        v = torch.randn(M, Cc, generator=g)
        npart = L().io_colsum_partial_floats(M, Cc)
        part = torch.empty(npart, device=DEV)
        s = torch.empty(Cc, device=DEV)
        _lib.check(L().io_colsum(P(dev(v)), M, Cc, P(s), P(part), npart, 0, ST()), "colsum")
        assert relerr(s, v.double().sum(0)) < 1e-6


@pytest.mark.parametrize("relu", [0, 1])
def test_head_one_channel(relu):
    M, C, pitch = 5000, 32, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, pitch, generator=g, dtype=torch.float64)
    w = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(1, generator=g, dtype=torch.float64).requires_grad_(True)
    xr = x[:, :C].clone().requires_grad_(True)
    z = xr @ w + b
    y = torch.relu(z) if relu else z
    dy = torch.randn(M, generator=g, dtype=torch.float64)
    gx, gw, gb = torch.autograd.grad(y, [xr, w, b], dy)
    xd = dev(x)
    out = torch.empty(M, device=DEV)
    _lib.check(L().io_head1_fwd(P(xd), M, pitch, C, P(dev(w.detach())), P(dev(b.detach())), relu, P(out), 0, ST()), "head fwd")
    assert relerr(out, y.detach()) < 1e-6
    npart = L().io_colsum_partial_floats(M, C)
    part = torch.empty(npart, device=DEV)
    dx = torch.full((M, pitch), 7.0, device=DEV)
    dw, db = torch.empty(C, device=DEV), torch.empty(1, device=DEV)
    _lib.check(L().io_head1_bwd(P(dev(dy)), P(out), P(xd), M, pitch, C, P(dev(w.detach())), relu, P(dx), P(dw), P(db), P(part),
                                npart, 0, ST()), "head bwd")
    assert relerr(dx[:, :C], gx) < 1e-6 and float(dx[:, C:].abs().max()) == 0.0
    assert relerr(dw, gw) < 1e-5 and relerr(db, gb) < 1e-5


def _bf(t):
    return t.float().bfloat16().double()


def test_grouped_conv_and_decoder_ops_bf16():
    """The same kernels on bf16 storage (dtype = 1): inputs rounded to bf16, fp64 torch reference, one output
    rounding (2^-8) on bf16 results, fp32-accumulated filter / bias gradients."""
    C, cg, H, N, stride = 512, 16, 10, 2, 2
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(N, C, H, H, generator=g)).requires_grad_(True)
    w = _bf(torch.randn(C, cg, 3, 3, generator=g) / np.sqrt(9 * cg)).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=1, groups=C // cg)
    Ho = y.shape[2]
    dy = _bf(torch.randn(y.shape, generator=g))
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    bdev = lambda t: t.float().bfloat16().to(DEV).contiguous()
    wc = torch.empty(C, 9, 64, device=DEV, dtype=torch.bfloat16)
    wtc = torch.empty_like(wc)
    _lib.check(L().io_gconv_pack(P(dev(w.detach().reshape(C, cg, 9))), C, cg, 9, P(wc), P(wtc), 1, ST()), "pack")
    xd = bdev(x.detach().permute(0, 2, 3, 1))
    yd = torch.empty(N, Ho, Ho, C, device=DEV, dtype=torch.bfloat16)
    _lib.check(L().io_gconv2d_fwd(P(xd), P(wc), P(yd), N, H, H, C, 3, 3, stride, 1, 1, ST()), "fwd")
    assert relerr(yd.float().permute(0, 3, 1, 2), y.detach()) < 6e-3
    dyd = bdev(dy.permute(0, 2, 3, 1))
    dxd = torch.empty(N, H, H, C, device=DEV, dtype=torch.bfloat16)
    _lib.check(L().io_gconv2d_dgrad(P(dyd), P(wtc), P(dxd), N, H, H, C, 3, 3, stride, 1, 1, ST()), "dgrad")
    assert relerr(dxd.float().permute(0, 3, 1, 2), gx) < 6e-3
    nb = L().io_gconv2d_wgrad_workspace_bytes(N, H, H, C, 3, 3, stride, 1)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dwc = torch.empty(C, 9, 64, device=DEV)
    _lib.check(L().io_gconv2d_wgrad(P(xd), P(dyd), P(dwc), N, H, H, C, 3, 3, stride, 1, P(ws), nb, 1, ST()), "wgrad")
    dw = torch.empty(C, cg, 9, device=DEV)
    _lib.check(L().io_gconv_unpack_grad(P(dwc), C, cg, 9, P(dw), ST()), "unpack")
    assert relerr(dw.view(C, cg, 3, 3), gw) < 2e-5
    # bilinear x2, bias + ReLU, column sums on bf16 tensors
    for Cu, al in ((64, 1), (12, 0), (40, 1)):      # 16-byte lanes (8 | C) and the 8-byte form (C = 12)
        xs = _bf(torch.randn(2, Cu, 5, 7, generator=g)).requires_grad_(True)
        up = F.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=bool(al))
        dup = _bf(torch.randn(up.shape, generator=g))
        gxs, = torch.autograd.grad(up, [xs], dup)
        out = torch.empty(2, 10, 14, Cu, device=DEV, dtype=torch.bfloat16)
        _lib.check(L().io_upsample2x_bilinear_fwd(P(bdev(xs.detach().permute(0, 2, 3, 1))), 2, 5, 7, Cu, al, P(out), 1, ST()), "up")
        assert relerr(out.float().permute(0, 3, 1, 2), up.detach()) < 6e-3
        dxs = torch.empty(2, 5, 7, Cu, device=DEV, dtype=torch.bfloat16)
        _lib.check(L().io_upsample2x_bilinear_bwd(P(bdev(dup.permute(0, 2, 3, 1))), 2, 5, 7, Cu, al, P(dxs), 1, ST()), "up bwd")
        assert relerr(dxs.float().permute(0, 3, 1, 2), gxs) < 6e-3
    M, Cc = 2000, 128
    v, b = _bf(torch.randn(M, Cc, generator=g)), torch.randn(Cc, generator=g).double()
    o = torch.empty(M, Cc, device=DEV, dtype=torch.bfloat16)
    _lib.check(L().io_bias_act(P(bdev(v)), P(dev(b)), M, Cc, 1, P(o), 1, ST()), "bias_act")
    assert relerr(o.float(), torch.relu(v + b)) < 6e-3
    npart = L().io_colsum_partial_floats(M, Cc)
    part, sm = torch.empty(npart, device=DEV), torch.empty(Cc, device=DEV)
    _lib.check(L().io_colsum(P(bdev(v)), M, Cc, P(sm), P(part), npart, 1, ST()), "colsum")
    assert relerr(sm, v.sum(0)) < 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (3, 64, 64), (1, 33, 17)])
def test_smooth_loss_value_and_gradient(B, H, W):
    """ops.smooth_loss (io_smooth_loss_fwd / _bwd) == the oracle's restatement of get_smooth_loss
    (models/supervised_order.py:214-235) in fp64, value and gradient -- incl. the gradient that reaches the elements the
    min / max normalisation selects, and an output scale (the loss weight / world size / the pair-mode factor 2)."""
    from instaorder_amd import ops
    from oracle import midas_oracle as mo
    g = torch.Generator().manual_seed(B * H + W)
    disp = torch.rand(B, 1, H, W, generator=g) * 3.0 + 0.2
    img = torch.randn(B, 3, H, W, generator=g)
    d64 = disp.double().requires_grad_(True)
    ref = mo.smooth_loss(d64, img.double()) * 0.7
    ref.backward()
    dd = disp.cuda().requires_grad_(True)
    got = ops.smooth_loss(dd, img.cuda(), 0.7)
    assert got.shape == () and abs(float(got) - float(ref)) < 1e-5 * abs(float(ref))
    (got * 1.5).backward()
    gref = d64.grad * 1.5
    err = (dd.grad.double().cpu() - gref).abs().max() / gref.abs().max()
    assert float(err) < 2e-4, float(err)
    # the two selected elements carry sums over the whole map: check them explicitly
    for b in range(B):
        flat = disp[b, 0].reshape(-1)
        for idx in (int(flat.argmin()), int(flat.argmax())):
            a, r = float(dd.grad[b, 0].reshape(-1)[idx]), float(gref[b, 0].reshape(-1)[idx])
            assert abs(a - r) < 2e-4 * float(gref.abs().max()) + 1e-3 * abs(r), (b, idx, a, r)


@pytest.mark.parametrize("B,H,W", [(6, 32, 48), (4, 64, 64), (3, 160, 208), (2, 384, 384)])   # 1, 2, 17, 64 blocks per sample
def test_disp_order_count(B, H, W):
    """ops.disp_order_count (io_disp_order_count) == the oracle's restatement of supervised_order.py:152-173 with
    scipy.ndimage.binary_erosion: every (is_overlap, depth_order) combination, an empty erosion (pair skipped), two
    different disparity maps."""
    from instaorder_amd import ops
    from oracle import midas_oracle as mo
    g = torch.Generator().manual_seed(H + B)
    d1 = torch.rand(B, 1, H, W, generator=g)
    d2 = torch.rand(B, 1, H, W, generator=g)
    m1, m2 = torch.zeros(B, 1, H, W), torch.zeros(B, 1, H, W)
    for b in range(B):
        y0, x0 = 2 + b, 3 + 2 * b
        m1[b, 0, y0:y0 + 10, x0:x0 + 12] = 1
        m2[b, 0, H - 14 - b:H - 3, 5:W - 6] = 1
    m2[B - 1] = 0
    m2[B - 1, 0, 4, 4:9] = 1                 # a one-pixel-high mask: its erosion is empty -> the pair is skipped
    order = torch.tensor([0, 1, 2, 0, 1, 0][:B])
    ovl = torch.tensor([0, 0, 0, 1, 0, 0][:B])
    ref = mo.disp_order_count(d1, d2, m1, m2, order[:B - 1].tolist() + [0], ovl[:B - 1].tolist() + [1])   # skip the last in the oracle
    got = ops.disp_order_count(d1.cuda(), d2.cuda(), m1.cuda(), m2.cuda(), order.cuda(), ovl.cuda(), 0, 1.0)
    assert abs(float(got) - ref) < 1e-6 * max(1.0, abs(ref)), (float(got), ref)
    assert ref > 0
    got2 = ops.disp_order_count(d1.cuda(), d2.cuda(), m1.cuda(), m2.cuda(), order.cuda(), ovl.cuda(), 0, 0.25)
    assert abs(float(got2) - 0.25 * ref) < 1e-6 * max(1.0, abs(ref))
