#!/usr/bin/env python3
"""Times io_conv2d_dgrad_fused_dt (the data-gradient launch of the training step) per ResNet-50 shape at the bench batch,
with and without the backward operand transform (IoBwStats::xb_*), on rotating buffers.
usage: python tools/xb_bench.py [fp32|bf16] [N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instaorder_amd import _lib

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = 1 if dtype == "bf16" else 0
td = torch.bfloat16 if dt else torch.float32
lib = _lib.lib()
_lib.require_gpu()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
G = 2
# (name, H, Cin (= channels of dx), Cout (= channels of the operand), k)
SHAPES = [("L1.c3", 64, 64, 256, 1), ("L1.c2", 64, 64, 64, 3), ("L1.c1", 64, 256, 64, 1),
          ("L2.c3", 32, 128, 512, 1), ("L2.c2", 32, 128, 128, 3), ("L2.c1", 32, 512, 128, 1),
          ("L3.c3", 16, 256, 1024, 1), ("L3.c2", 16, 256, 256, 3), ("L3.c1", 16, 1024, 256, 1),
          ("L4.c3", 8, 512, 2048, 1), ("L4.c2", 8, 512, 512, 3), ("L4.c1", 8, 2048, 512, 1)]
NROT = 3
tot = [0.0, 0.0]
for name, H, Cin, Cout, k in SHAPES:
    M = N * H * H
    mk = lambda c: [torch.randn(N, H, H, c, device="cuda").to(td) for _ in range(NROT)]   # noqa: E731
    dz, yb, dyo = mk(Cout), mk(Cout), mk(Cout)
    ya, dx, ao = mk(Cin), mk(Cin), mk(Cin)
    wt = (torch.randn(Cout, k * k, Cin, device="cuda") / (Cin * k * k) ** 0.5).to(td)
    coef = torch.randn(3 * G * Cout, device="cuda")
    mean, rstd, scale, shift = (torch.rand(G * Cin, device="cuda") + 0.5 for _ in range(4))
    nt = lib.io_bn_tile_partial_floats(M, Cin, G)
    p1, p2 = torch.empty(nt, device="cuda"), torch.empty(nt, device="cuda")
    res = []
    for xb in (0, 1):
        def run(i):
            o = _lib.DgradFused()
            if xb:
                o.xb_y, o.xb_coef, o.xb_dy_out = yb[i].data_ptr(), coef.data_ptr(), dyo[i].data_ptr()
            o.ep_y, o.ep_mean, o.ep_rstd, o.ep_p1, o.ep_p2 = ya[i].data_ptr(), mean.data_ptr(), rstd.data_ptr(), p1.data_ptr(), p2.data_ptr()
            o.ep_scale, o.ep_shift, o.ep_act_out = scale.data_ptr(), shift.data_ptr(), ao[i].data_ptr()
            _lib.check(lib.io_conv2d_dgrad_fused_dt(dz[i].data_ptr(), wt.data_ptr(), dx[i].data_ptr(), N, H, H, Cin, Cout, k, k,
                                                    k // 2, G, C.byref(o), dt, st), "dgrad_fused")
        for i in range(NROT):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4
        e0.record()
        for r in range(reps):
            for i in range(NROT):
                run(i)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / (reps * NROT))
    fl = 2.0 * M * Cin * Cout * k * k
    es = 2 if dt else 4
    apply_ms = 3.0 * M * Cout * es / 5.0e9       # the apply pass the transform replaces, at 5 TB/s
    tot[0] += res[0] + apply_ms
    tot[1] += res[1]
    print("%-6s M=%8d %4d<-%4d k%d | plain %.3f ms (%.0f TF/s) + apply %.3f = %.3f | xb %.3f ms (%+.3f vs plain, %+.3f net)" % (
        name, M, Cin, Cout, k, res[0], fl / res[0] / 1e9, apply_ms, res[0] + apply_ms, res[1], res[1] - res[0],
        res[1] - res[0] - apply_ms))
    del dz, yb, dyo, ya, dx, ao
    torch.cuda.empty_cache()
print("sum over shapes: plain + apply %.2f ms, xb %.2f ms" % (tot[0], tot[1]))
