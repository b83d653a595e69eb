#!/usr/bin/env python3
"""Short-K 1x1 layers in the network vs on their own: the same convolution launched plain and with the fused BatchNorm
statistics epilogue, on ROTATING buffers (four input / output sets, so nothing is cache-warm -- a microbench that
re-uses one buffer pair flatters write-heavy kernels).  usage: python tools/shortk_probe.py [bf16|fp32]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instaorder_amd import _lib

DT = 0 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else 1
TD = torch.float32 if DT == 0 else torch.bfloat16
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
NSET = 4
for (N, H, Cin, Cout) in ((512, 64, 64, 256), (512, 64, 256, 64), (512, 32, 128, 512), (512, 32, 512, 128), (512, 16, 256, 1024)):
    xs = [torch.randn(N, H, H, Cin, device="cuda").to(TD) for _ in range(NSET)]
    ys = [torch.empty(N, H, H, Cout, device="cuda", dtype=TD) for _ in range(NSET)]
    w = (torch.randn(Cout, 1, Cin, device="cuda") * 0.05).to(TD)
    gam, bet = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    mean, rstd, sc, sh = (torch.empty(Cout, device="cuda") for _ in range(4))
    nws = L.io_conv2d_bnstats_workspace_floats(N, H, H, Cout, 1, 1, 1, 0, 2)
    ws = torch.empty(nws, device="cuda")

    def plain(i):
        return L.io_conv2d_fwd_dt(P(xs[i]), P(w), P(ys[i]), N, H, H, Cin, Cout, 1, 1, 1, 0, DT, DT, S())

    def stats(i):
        return L.io_conv2d_fwd_bnstats_dt(P(xs[i]), P(w), P(ys[i]), N, H, H, Cin, Cout, 1, 1, 1, 0, 2, P(gam), P(bet), P(rm),
                                          P(rv), C.c_float(0.1), C.c_float(1e-5), P(mean), P(rstd), P(sc), P(sh), P(ws),
                                          nws, DT, 0, S())
    by = (2 if DT else 4) * N * H * H * (Cin + Cout)
    out = []
    for fn in (plain, stats):
        for i in range(NSET):
            assert fn(i) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for r in range(20):
            fn(r % NSET)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out.append("%.3f ms %.2f TB/s" % (ms, by / ms / 1e9))
    print("%4d x %3d^2  %4d -> %4d   plain %s | with statistics (+ finalize launch) %s" % (N, H, Cin, Cout, out[0], out[1]))
