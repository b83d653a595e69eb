#!/usr/bin/env python3
"""Winograd F(2, 3) row form against the direct implicit-GEMM kernel on the 3x3 stride-1 shapes of ResNet-50 at the bench
batch: forward in the executor's form (input transform + statistics epilogue) and the data gradient with the fused
BatchNorm-backward epilogue + activation side output.  usage: python tools/wino_bench.py [N] [S] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
G = 2


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


print("shape (H, C)      | fwd xf+stats: direct ms  TF/s | wino ms  TF/s  ratio | dgrad+bn epilogue: direct ms TF/s | wino ms  TF/s  ratio")
tot = [0.0, 0.0, 0.0, 0.0]
for H, Cc, cnt in ((S // 4, 64, 3), (S // 8, 128, 3), (S // 16, 256, 5), (S // 32, 512, 2)):
    M = N * H * H
    x = torch.randn(N, H, H, Cc, device="cuda")
    w = torch.randn(Cc, 9, Cc, device="cuda") * 0.05
    y = torch.empty(N, H, H, Cc, device="cuda")
    tabs = [torch.randn(G * Cc, device="cuda") * 0.1 for _ in range(3)]
    gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    o4 = [torch.empty(G * Cc, device="cuda") for _ in range(4)]
    nws = L.io_conv2d_bnstats_workspace_floats(N, H, H, Cc, 3, 3, 1, 1, G)
    ws = torch.empty(nws, device="cuda")
    nsc = L.io_conv2d_wino_scratch_floats(Cc, Cc)
    sc = torch.empty(nsc, device="cuda")
    fl = 2.0 * M * Cc * Cc * 9
    f_d = lambda: _lib.check(L.io_conv2d_fwd_xf_dt(P(x), P(w), P(y), N, H, H, Cc, Cc, 3, 3, 1, 1, G, P(tabs[0]), P(tabs[1]),
                                                   P(tabs[2]), P(gamma), P(beta), P(rm), P(rv), 0.1, 1e-5, P(o4[0]), P(o4[1]),
                                                   P(o4[2]), P(o4[3]), P(ws), nws, 0, ST()), "xf")
    f_w = lambda: _lib.check(L.io_conv2d_fwd_wino(P(x), P(w), P(y), N, H, H, Cc, Cc, G, P(tabs[0]), P(tabs[1]), P(tabs[2]),
                                                  P(gamma), P(beta), P(rm), P(rv), 0.1, 1e-5, P(o4[0]), P(o4[1]), P(o4[2]),
                                                  P(o4[3]), P(ws), nws, P(sc), nsc, ST()), "wino")
    t_fd, t_fw = timeit(f_d), timeit(f_w)
    # data gradient with the fused epilogue
    dy = torch.randn(N, H, H, Cc, device="cuda")
    ya = torch.randn(N, H, H, Cc, device="cuda")
    dx = torch.empty(N, H, H, Cc, device="cuda")
    aout = torch.empty(N, H, H, Cc, device="cuda")
    nt = L.io_bn_tile_partial_floats(M, Cc, G)
    p1, p2 = torch.empty(nt, device="cuda"), torch.empty(nt, device="cuda")

    def mk(wino):
        o = _lib.DgradFused()
        o.ep_y, o.ep_mean, o.ep_rstd, o.ep_scale, o.ep_shift = (ya.data_ptr(), tabs[0].data_ptr(), o4[1].data_ptr(),
                                                                tabs[1].data_ptr(), tabs[2].data_ptr())
        o.ep_p1, o.ep_p2, o.ep_act_out = p1.data_ptr(), p2.data_ptr(), aout.data_ptr()
        if wino:
            o.wino_scratch, o.wino_scratch_floats = sc.data_ptr(), nsc
        return o
    od, ow = mk(False), mk(True)
    o4[1].fill_(1.0)
    d_d = lambda: _lib.check(L.io_conv2d_dgrad_fused_dt(P(dy), P(w), P(dx), N, H, H, Cc, Cc, 3, 3, 1, G, C.byref(od), 0, ST()), "dg")
    d_w = lambda: _lib.check(L.io_conv2d_dgrad_fused_dt(P(dy), P(w), P(dx), N, H, H, Cc, Cc, 3, 3, 1, G, C.byref(ow), 0, ST()), "dgw")
    t_dd, t_dw = timeit(d_d), timeit(d_w)
    for i, t in enumerate((t_fd, t_fw, t_dd, t_dw)):
        tot[i] += cnt * t
    print("H=%3d C=%3d x%d | %7.3f %6.1f | %7.3f %6.1f  %.2fx | %7.3f %6.1f | %7.3f %6.1f  %.2fx" %
          (H, Cc, cnt, t_fd, fl / t_fd / 1e9, t_fw, fl / t_fw / 1e9, t_fd / t_fw, t_dd, fl / t_dd / 1e9, t_dw, fl / t_dw / 1e9,
           t_dd / t_dw))
    del x, w, y, dy, ya, dx, aout
print("13 layers: forward direct %.2f ms -> wino %.2f ms;  data gradient direct %.2f -> wino %.2f ms" % tuple(tot))
