#!/usr/bin/env python3
"""Per-layer-shape timing of the implicit-GEMM kernels (forward / data-gradient / filter-gradient) for every
distinct convolution of ResNet-50 at the bench batch (N samples of SxS).  Tuning aid; prints TFLOP/s.
usage: python tools/conv_bench.py [N] [S] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

shapes = []   # (name, H, Cin, Cout, k, stride, pad, count)
H = S // 4
inC = 64
for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
    for b in range(blocks):
        stride = 2 if (b == 0 and li > 0) else 1
        shapes.append(("l%d.%d.c1" % (li + 1, b), H, inC, planes, 1, 1, 0))
        shapes.append(("l%d.%d.c2" % (li + 1, b), H, planes, planes, 3, stride, 1))
        shapes.append(("l%d.%d.c3" % (li + 1, b), H // stride, planes, planes * 4, 1, 1, 0))
        if b == 0:
            shapes.append(("l%d.%d.ds" % (li + 1, b), H, inC, planes * 4, 1, stride, 0))
        inC = planes * 4
        H //= stride
uniq = {}
for s in shapes:
    uniq.setdefault(s[1:], []).append(s[0])


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


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
flo = 0.0
print("%-28s %5s %5s %5s k s  cnt |   fwd ms   TF |  dgrad ms   TF |  wgrad ms   TF" % ("shape", "H", "Cin", "Cout"))
for (Hh, Cin, Cout, k, st, pad), names in uniq.items():
    Ho = (Hh + 2 * pad - k) // st + 1
    x = torch.randn(N, Hh, Hh, Cin, device="cuda")
    w = torch.randn(Cout, k * k, Cin, device="cuda") * 0.05
    wt = torch.randn(Cin, k * k, Cout, device="cuda") * 0.05
    y = torch.empty(N, Ho, Ho, Cout, device="cuda")
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda")
    dx = torch.empty(N, Hh, Hh, Cin, device="cuda")
    dw = torch.empty(Cout, k * k, Cin, device="cuda")
    nb = L.io_conv2d_wgrad_workspace_bytes(N, Hh, Hh, Cin, Cout, k, k, st, pad)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
    t_f = timeit(lambda: L.io_conv2d_fwd(P(x), P(w), P(y), N, Hh, Hh, Cin, Cout, k, k, st, pad, ST()))
    t_d = timeit(lambda: L.io_conv2d_dgrad(P(dy), P(wt), P(dx), None, N, Hh, Hh, Cin, Cout, k, k, st, pad, ST()))
    t_w = timeit(lambda: L.io_conv2d_wgrad(P(x), P(dy), P(dw), N, Hh, Hh, Cin, Cout, k, k, st, pad, P(ws), nb, ST()))
    c = len(names)
    tot["fwd"] += c * t_f
    tot["dgrad"] += c * t_d
    tot["wgrad"] += c * t_w
    flo += c * fl
    print("%-28s %5d %5d %5d %d %d  %3d | %8.3f %5.1f | %8.3f %5.1f | %8.3f %5.1f" %
          (names[0], Hh, Cin, Cout, k, st, c, t_f, fl / t_f / 1e9, t_d, fl / t_d / 1e9, t_w, fl / t_w / 1e9))
    del x, w, wt, y, dy, dx, dw, ws
print("total ms  fwd %.1f  dgrad %.1f  wgrad %.1f   (TF/s: %.1f %.1f %.1f)" %
      (tot["fwd"], tot["dgrad"], tot["wgrad"], flo / tot["fwd"] / 1e9, flo / tot["dgrad"] / 1e9, flo / tot["wgrad"] / 1e9))
