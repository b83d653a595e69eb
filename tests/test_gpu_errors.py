"""Error behaviour of the C ABI: unsupported shapes / too-small workspaces return a negative code and leave a message
in io_last_error_string(); nothing aborts, divides by zero or launches."""
import ctypes as C

import pytest
import torch

from instaorder_amd import _lib
from test_gpu_ops import L, P, ST

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bad(rc):
    assert rc < 0, rc
    assert len(L().io_last_error_string()) > 0


def buf(n=1 << 16):
    return torch.zeros(n, device=DEV)


def test_conv_entry_points_reject_bad_shapes():
    a, b, c = buf(), buf(), buf()
    bad(L().io_conv2d_fwd(P(a), P(b), P(c), 1, 8, 8, 3, 64, 3, 3, 1, 1, ST()))           # Cin not a multiple of 32
    bad(L().io_conv2d_fwd(P(a), P(b), P(c), 1, 8, 8, 32, 10, 3, 3, 1, 1, ST()))          # Cout not a multiple of 64
    bad(L().io_conv2d_fwd(P(a), P(b), P(c), 0, 8, 8, 32, 64, 3, 3, 1, 1, ST()))          # empty batch
    bad(L().io_conv2d_fwd(P(a), P(b), P(c), 1, 2, 2, 32, 64, 7, 7, 1, 0, ST()))          # window larger than the image
    bad(L().io_conv2d_dgrad(P(a), P(b), P(c), None, None, 1, 8, 8, 48, 64, 3, 3, 1, 1, ST()))
    bad(L().io_conv2d_wgrad(P(a), P(b), P(c), 1, 8, 8, 32, 64, 3, 3, 1, 1, P(a), 0, ST()))
    bad(L().io_conv2d_wgrad(P(a), P(b), P(c), 1, 8, 8, 64, 96, 3, 3, 1, 1, P(a), 1 << 16, ST()))
    # workspace too small for a split reduction
    need = L().io_conv2d_wgrad_workspace_bytes(8, 32, 32, 64, 64, 3, 3, 1, 1)
    assert need > 0
    big = torch.zeros(8 * 32 * 32 * 64, device=DEV)
    bad(L().io_conv2d_wgrad(P(big), P(big), P(c), 8, 32, 32, 64, 64, 3, 3, 1, 1, P(a), need - 1, ST()))
    bad(L().io_conv2d_fwd_dt(P(a), P(b), P(c), 1, 8, 8, 64, 64, 3, 3, 1, 1, 7, 7, ST()))  # unknown dtype


def test_bn_pool_loss_sgd_reject_bad_arguments():
    a, b = buf(), buf()
    t = [buf(64) for _ in range(8)]
    npart = L().io_bn_partial_floats(64, 64, 1)
    bad(L().io_bn_stats_finalize(P(a), 64, 6, 1, P(t[0]), P(t[1]), None, None, 0.1, 1e-5, P(t[2]), P(t[3]), P(t[4]),
                                 P(t[5]), P(b), npart, ST()))                              # C not 4 * 2^k
    bad(L().io_bn_stats_finalize(P(a), 65, 64, 2, P(t[0]), P(t[1]), None, None, 0.1, 1e-5, P(t[2]), P(t[3]), P(t[4]),
                                 P(t[5]), P(b), npart, ST()))                              # M not divisible by G
    bad(L().io_bn_stats_finalize(P(a), 64, 64, 1, P(t[0]), P(t[1]), None, None, 0.1, 1e-5, P(t[2]), P(t[3]), P(t[4]),
                                 P(t[5]), P(b), 1, ST()))                                  # partial buffer too small
    bad(L().io_bn_apply(P(a), 64, 48, 1, 0, P(t[0]), P(t[1]), P(t[2]), None, None, None, None, 1, P(b), ST()))
    idx = torch.zeros(1024, dtype=torch.int32, device=DEV)
    bad(L().io_maxpool_fwd(P(a), 1, 8, 8, 6, P(b), P(idx), ST()))                          # C not a multiple of 4
    bad(L().io_upsample2x_bilinear_fwd(P(a), 1, 4, 4, 6, 1, P(b), 0, ST()))
    bad(L().io_upsample2x_bilinear_fwd(P(a), 1, 4, 4, 8, 1, P(b), 9, ST()))                # unknown dtype
    bad(L().io_gconv_pack(P(a), 96, 8, 9, P(b), P(b), 0, ST()))                            # C not a multiple of 64
    bad(L().io_gconv_pack(P(a), 128, 24, 9, P(b), P(b), 0, ST()))                          # group width does not divide 64
    bad(L().io_head1_fwd(P(a), 16, 64, 5, P(t[0]), P(t[1]), 1, P(b), 0, ST()))
    bad(L().io_colsum(P(a), 16, 48, P(t[0]), P(b), 1 << 16, 0, ST()))                      # C does not divide 256
    bad(L().io_add(P(a), P(b), 6, P(b), 0, ST()))
    planes = (C.c_void_p * 1)(a.data_ptr())
    strides = (C.c_long * 1)(64)
    bad(L().io_pack_planes_nhwc8(planes, strides, 0, 1, 8, 8, P(b), ST()))
    bad(L().io_pack_planes_nhwc8(planes, strides, 6, 1, 8, 8, P(b), ST()))


def test_network_executor_rejects_bad_calls():
    heads = (C.c_int * 1)(2)
    assert not L().io_net_create(9, 1, heads)                                               # in_channels > 5
    assert not L().io_net_create(5, 3, heads)                                               # too many heads
    net = C.c_void_p(L().io_net_create(5, 1, heads))
    assert net
    try:
        assert L().io_net_workspace_bytes(net, 4, 100, 1) == 0                              # S not a multiple of 32
        assert L().io_net_workspace_bytes(net, 0, 64, 1) == 0
        assert L().io_net_activation_offset(net, 4, 64, 99) == -1
        bad(L().io_net_set_dtype(net, 5))
        n = L().io_net_param_floats(net)
        params, running = buf(n), buf(L().io_net_running_floats(net))
        x8 = torch.zeros(4, 64, 64, 8, device=DEV)
        logits = torch.zeros(4, 2, device=DEV)
        need = L().io_net_workspace_bytes(net, 4, 64, 1)
        ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
        bad(L().io_net_forward(net, P(params), P(running), P(x8), 4, 64, 2, 1, P(ws), need - 1, P(logits), ST()))
        bad(L().io_net_forward(net, P(params), P(running), P(x8), 4, 64, 3, 1, P(ws), need, P(logits), ST()))   # G !| N
        bad(L().io_net_forward(net, P(params), P(running), P(x8), 4, 64, 2, 0, P(ws), need, P(logits), ST()))   # eval G != 1
    finally:
        L().io_net_destroy(net)


def test_two_host_threads_on_two_streams():
    """The library keeps no global mutable state on the data path: two host threads, each on its own HIP stream with its
    own buffers, interleave launches of all three convolution kernels and get the results of the serial run."""
    import threading
    N, H, Ci, Co = 4, 16, 64, 128
    g = torch.Generator(device=DEV).manual_seed(1)
    data = []
    for _ in range(2):
        x = torch.randn(N, H, H, Ci, device=DEV, generator=g)
        w = torch.randn(Co, 9, Ci, device=DEV, generator=g) * 0.05
        dy = torch.randn(N, H, H, Co, device=DEV, generator=g)
        data.append((x, w, dy))
    nb = L().io_conv2d_wgrad_workspace_bytes(N, H, H, Ci, Co, 3, 3, 1, 1)

    def run(x, w, dy, stream, reps):
        y = torch.empty(N, H, H, Co, device=DEV)
        dw = torch.empty(Co, 9, Ci, device=DEV)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
        s = C.c_void_p(stream.cuda_stream)
        for _ in range(reps):
            assert L().io_conv2d_fwd(P(x), P(w), P(y), N, H, H, Ci, Co, 3, 3, 1, 1, s) == 0
            assert L().io_conv2d_wgrad(P(x), P(dy), P(dw), N, H, H, Ci, Co, 3, 3, 1, 1, P(ws), nb, s) == 0
        stream.synchronize()
        return y, dw

    torch.cuda.synchronize()
    serial = [run(*d, torch.cuda.current_stream(), 1) for d in data]
    out = [None, None]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def worker(i):
        out[i] = run(*data[i], streams[i], 50)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(2):
        assert torch.equal(out[i][0], serial[i][0]) and torch.equal(out[i][1], serial[i][1])


def test_invalid_labels_fail_loudly():
    """A class id outside the head (datasets can emit depth_label = -1; nn.CrossEntropyLoss device-asserts on it in the
    reference): host tensors are rejected in set_input, device tensors poison the loss with NaN -- never a silent read past
    the kernel's local arrays.  is_overlap values other than 0 / 1 belong to neither subset (supervised_order.py:62-73)."""
    import numpy as np
    import instaorder_amd as ia
    from instaorder_amd import engine, synthetic
    cfg = dict(algo="InstaOrderNet_od", lr=1e-4, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
               backbone_param=dict(in_channels=5, num_classes=[2, 3]), use_rgb=True, overlap_weight=0.1,
               distinct_weight=0.9)
    m = ia.InstaOrderNet_od(cfg, dist_model=False)
    b = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_pair_batch(3, 4, 64).items()}
    bad = b["depth_order"].clone()
    bad[1] = -1
    with pytest.raises(ValueError):
        m.set_input(b["rgb"], b["modal1"], b["modal2"], bad, b["count"], b["is_overlap"], b["occ_order"])
    # ... unless that row takes no part in the loss: is_overlap = -1 is what the reference's reader leaves on overlapped
    # pairs under remove_depth_overlap (datasets/reader.py:363-380); its boolean masks drop the row (supervised_order.py:
    # 62-73), so neither the flag nor the row's label is an error, and the loss is the one of the remaining rows
    ovl = b["is_overlap"].clone()
    ovl[1] = -1
    m.switch_to("train")
    m.set_input(b["rgb"], b["modal1"], b["modal2"], bad, b["count"], ovl, b["occ_order"])
    logs_a, _ = m.forward_only()
    ok = b["depth_order"].clone()
    ok[1] = 2
    m.set_input(b["rgb"], b["modal1"], b["modal2"], ok, b["count"], ovl, b["occ_order"])
    logs_b, _ = m.forward_only()
    assert np.isfinite(float(logs_a["loss_depth"])) and float(logs_a["loss_depth"]) == float(logs_b["loss_depth"])
    # the same label arriving on the device: NaN loss
    B = 4
    logits = torch.randn(2 * B, 5, device="cuda")
    dep = torch.tensor([0, 1, 2, 3, 0, 1, 2, 0], device="cuda")           # 3 is out of range for a 3-class head
    ov = torch.tensor([0, 1, 0, 1], device="cuda")
    occ = torch.zeros(2 * B, 2, device="cuda")
    losses, _ = engine.order_loss(logits, B, 2, 3, occ_target=occ, depth_target=dep, is_overlap=ov, overlap_weight=0.1,
                                  distinct_weight=0.9)
    assert torch.isnan(losses[0]) and torch.isnan(losses[2])
    # is_overlap = 2 on one row: that row counts in neither subset
    dep_ok = torch.tensor([0, 1, 2, 1, 1, 0, 2, 0], device="cuda")
    l_a, _ = engine.order_loss(logits, B, 2, 3, occ_target=occ, depth_target=dep_ok,
                               is_overlap=torch.tensor([0, 1, 2, 1], device="cuda"), overlap_weight=0.1, distinct_weight=0.9)
    assert torch.isfinite(l_a).all()
    z = logits.double().cpu()
    q = torch.softmax(z[:, 2:], 1)
    ce = -(torch.log_softmax(q, 1)[torch.arange(2 * B), dep_ok.cpu()])
    io = np.array([0, 1, 2, 1] * 2)
    want = 0.1 * (ce[io == 1].sum() / (io[:B] == 1).sum()) + 0.9 * (ce[io == 0].sum() / (io[:B] == 0).sum())
    assert abs(float(l_a[2]) - float(want)) < 1e-5 * abs(float(want))


def test_round3_entry_points_reject_bad_arguments():
    """The fused data gradient, the residual forward form, the coefficient producers, the staged backward and the two
    MiDaS losses: unsupported shapes are error codes with a message, nothing is launched on them."""
    a, b, c, d = buf(1 << 18), buf(1 << 18), buf(1 << 18), buf(1 << 18)
    t = [buf(4096) for _ in range(6)]
    opt = _lib.DgradFused()
    # rows per BatchNorm group not a multiple of 128 (N*H*W = 192, G = 1)
    opt.xb_y, opt.xb_coef = b.data_ptr(), t[0].data_ptr()
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 3, 8, 8, 64, 64, 1, 1, 0, 1, C.byref(opt), 0, ST()))
    # coefficient tables without the BatchNorm input they go with
    opt = _lib.DgradFused()
    opt.xb_coef = t[0].data_ptr()
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 2, 8, 8, 64, 64, 1, 1, 0, 1, C.byref(opt), 0, ST()))
    # epilogue BatchNorm without its partial-sum buffers / activation side output without the mask tables
    opt = _lib.DgradFused()
    opt.ep_y, opt.ep_mean, opt.ep_rstd = b.data_ptr(), t[0].data_ptr(), t[1].data_ptr()
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 2, 8, 8, 64, 64, 1, 1, 0, 1, C.byref(opt), 0, ST()))
    opt.ep_p1, opt.ep_p2, opt.ep_act_out = t[2].data_ptr(), t[3].data_ptr(), a.data_ptr()
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 2, 8, 8, 64, 64, 1, 1, 0, 1, C.byref(opt), 0, ST()))
    # the operand transform on a 5x5 window (taps without a same-size output grid are fine, but Cin must be a multiple of 64)
    opt = _lib.DgradFused()
    opt.xb_y, opt.xb_coef = b.data_ptr(), t[0].data_ptr()
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 2, 8, 8, 48, 64, 1, 1, 0, 1, C.byref(opt), 0, ST()))
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 2, 8, 8, 64, 64, 1, 1, 0, 1, None, 0, ST()))   # no option struct
    bad(L().io_conv2d_dgrad_fused_dt(P(a), P(c), P(d), 2, 8, 8, 64, 64, 1, 1, 0, 1, C.byref(opt), 5, ST()))   # dtype
    # residual forward form: rows per group, missing tables
    bad(L().io_conv2d_fwd_resid(P(a), P(b), P(c), P(d), None, 3, 8, 8, 64, 64, 1, P(t[0]), P(t[1]), P(t[2]), None, None, None,
                                None, 0.1, 1e-5, None, None, None, None, None, 0, ST()))
    bad(L().io_conv2d_fwd_resid(P(a), P(b), P(c), P(d), None, 2, 8, 8, 64, 64, 1, None, P(t[1]), P(t[2]), None, None, None,
                                None, 0.1, 1e-5, None, None, None, None, None, 0, ST()))
    # coefficient producers
    bad(L().io_bn_bwd_coefs_dt(P(a), P(b), 128, 6, 1, P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), P(t[5]), P(c), 1 << 18, 0,
                               ST()))                                                       # C not 4 * 2^k
    bad(L().io_bn_bwd_coefs_dt(P(a), P(b), 128, 64, 1, P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), P(t[5]), P(c), 1, 0,
                               ST()))                                                       # partial buffer too small
    bad(L().io_bn_bwd_coefs_from_tile_partials(P(a), P(b), 192, 64, 1, P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), P(t[5]),
                                               ST()))                                       # rows not whole 128-row tiles
    # exact-K stem
    bad(L().io_stem_wgrad_exact(P(a), P(b), P(c), 1, 64, 64, 9, P(d), 1 << 20, P(t[0]), ST()))
    # ... its fused-BatchNorm form is for whole 128-pixel output rows only (here Wo = 32), and wants G | N
    bad(L().io_stem_wgrad_exact_bn(P(a), P(b), P(c), P(d), 1, 64, 64, 5, 1, P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]),
                                   P(t[5]), P(t[0]), P(t[1]), P(d), 1 << 18, P(d), 1 << 20, P(t[2]), ST()))
    # ... and so is the bf16 one (256 x 256 inputs; here 64 x 64), which also wants its workspace
    bad(L().io_stem_wgrad_bn_bf16(P(a), P(b), P(c), P(d), 1, 64, 64, 1, P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), P(t[5]), P(t[0]),
                                  P(t[1]), P(d), 1 << 18, P(d), 1 << 20, ST()))
    bad(L().io_stem_wgrad_bn_bf16(P(a), P(b), P(c), P(d), 2, 256, 256, 1, P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), P(t[5]), P(t[0]),
                                  P(t[1]), P(d), 1 << 18, P(d), 1 << 10, ST()))                     # workspace too small
    # MiDaS losses
    bad(L().io_smooth_loss_fwd(P(a), P(b), 2, 1, 8, 1.0, P(t[0]), P(c), P(d), 1 << 18, ST()))          # one row: no y edges
    bad(L().io_smooth_loss_fwd(P(a), P(b), 2, 8, 8, 1.0, P(t[0]), P(c), P(d), 1, ST()))                # workspace
    lab = torch.zeros(4, dtype=torch.long, device=DEV)
    bad(L().io_disp_order_count(P(a), P(a), P(b), P(b), P(lab), P(lab), 2, 2, 8, 0, 1.0, P(t[0]), P(d), 1 << 18, ST()))   # H = 2
    bad(L().io_disp_order_count(P(a), P(a), P(b), P(b), P(lab), P(lab), 2, 8, 8, 0, 1.0, P(t[0]), P(d), 1, ST()))         # workspace


def test_rectangular_inference_entries_reject_bad_shapes():
    """io_net_forward_eval_hw / io_net_workspace_bytes_hw / io_pair_planes_u8_hw: sides that are not multiples of 32, a
    short workspace, empty outputs; training on a non-square batch is refused by the module."""
    import instaorder_amd as ia
    from instaorder_amd import engine
    net = engine.Net(5, 2)
    assert L().io_net_workspace_bytes_hw(net.handle, 2, 96, 70) == 0
    assert L().io_net_workspace_bytes_hw(net.handle, 2, 0, 64) == 0
    nb = net.workspace_bytes_hw(2, 96, 64)
    assert nb > 0
    params = torch.zeros(net.param_floats, device=DEV)
    running = torch.ones(net.running_floats, device=DEV)
    x8 = torch.zeros(2, 96, 64, 8, device=DEV)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    logits = torch.empty(2, 2, device=DEV)

    def bad(rc):
        assert rc != 0 and _lib.last_error()

    bad(L().io_net_forward_eval_hw(net.handle, P(params), P(running), P(x8), 2, 96, 70, P(ws), nb, P(logits), ST()))
    bad(L().io_net_forward_eval_hw(net.handle, P(params), P(running), P(x8), 2, 96, 64, P(ws), nb // 2, P(logits), ST()))
    assert L().io_net_forward_eval_hw(net.handle, P(params), P(running), P(x8), 2, 96, 64, P(ws), nb, P(logits), ST()) == 0
    torch.cuda.synchronize()
    bad(L().io_pair_planes_u8_hw(P(x8), 16, P(x8), P(x8), 1, 0, 64, None, None, None, P(logits), P(logits), ST()))
    cfg = dict(algo="InstaOrderNet_o", lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
               backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True)
    m = ia.InstaOrderNet_o(cfg, dist_model=False)
    m.switch_to("train")
    with pytest.raises(ValueError):
        m.net.forward_packed(torch.zeros(2, 96, 64, 8, device=DEV), 1)
    m.switch_to("eval")
    assert tuple(m.net.forward_packed(torch.zeros(2, 96, 64, 8, device=DEV), 1).shape) == (2, 2)


def test_staged_backward_rejects_bad_stage_ranges():
    import instaorder_amd as ia
    from instaorder_amd import engine
    cfg = dict(algo="InstaOrderNet_o", lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
               backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True)
    m = ia.InstaOrderNet_o(cfg, dist_model=False)
    net = m.net
    x8 = torch.zeros(2, 64, 64, 8, device=DEV)
    m.switch_to("train")
    logits, ws = net._run_forward(x8, 2, 64, 2, True)
    dl = torch.zeros_like(logits)
    for lo, hi in ((2, 2), (-1, 2), (0, 5), (3, 1)):
        with pytest.raises(RuntimeError):
            net._run_backward(x8, dl, 2, 64, 2, ws, stages=(lo, hi))
    net._run_backward(x8, dl, 2, 64, 2, ws, stages=(0, 4))
    net._pool.give(ws)
