#!/usr/bin/env python3
"""Per-layer-shape timing of the implicit-GEMM kernels (forward / data-gradient / filter-gradient) for every
distinct convolution of ResNet-50 at the bench batch (N samples of SxS).  Tuning aid; prints TFLOP/s and the
algorithmic GB/s (operands read once + result written once).
usage: python tools/conv_bench.py [N] [S] [reps] [fp32|bf16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
DT = 1 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else 0
TD = torch.bfloat16 if DT else torch.float32
ES = 2 if DT else 4
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
print("%-12s %4s %5s %5s k s cnt |  fwd ms    TF  GB/s | dgrad ms    TF  GB/s | wgrad ms    TF  GB/s" % ("shape", "H", "Cin", "Cout"))
for (Hh, Cin, Cout, k, st, pad), names in uniq.items():
    Ho = (Hh + 2 * pad - k) // st + 1
    x = torch.randn(N, Hh, Hh, Cin, device="cuda").to(TD)
    w = (torch.randn(Cout, k * k, Cin, device="cuda") * 0.05).to(TD)
    wt = (torch.randn(Cin, k * k, Cout, device="cuda") * 0.05).to(TD)
    y = torch.empty(N, Ho, Ho, Cout, device="cuda", dtype=TD)
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(TD)
    dx = torch.empty(N, Hh, Hh, Cin, device="cuda", dtype=TD)
    dw = torch.empty(Cout, k * k, Cin, device="cuda")
    nb = L.io_conv2d_wgrad_workspace_bytes(N, Hh, Hh, Cin, Cout, k, k, st, pad)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
    t_f = timeit(lambda: L.io_conv2d_fwd_dt(P(x), P(w), P(y), N, Hh, Hh, Cin, Cout, k, k, st, pad, DT, DT, ST()))
    t_d = timeit(lambda: L.io_conv2d_dgrad_dt(P(dy), P(wt), P(dx), None, None, N, Hh, Hh, Cin, Cout, k, k, st, pad, DT,
                                              ST()))
    t_w = timeit(lambda: L.io_conv2d_wgrad_dt(P(x), P(dy), P(dw), N, Hh, Hh, Cin, Cout, k, k, st, pad, P(ws), nb, DT, DT,
                                              ST()))
    by = ES * (N * Hh * Hh * Cin + N * Ho * Ho * Cout) + ES * Cout * Cin * k * k
    c = len(names)
    tot["fwd"] += c * t_f
    tot["dgrad"] += c * t_d
    tot["wgrad"] += c * t_w
    flo += c * fl
    print("%-12s %4d %5d %5d %d %d %3d | %7.3f %6.1f %5.0f | %7.3f %6.1f %5.0f | %7.3f %6.1f %5.0f" %
          (names[0], Hh, Cin, Cout, k, st, c, t_f, fl / t_f / 1e9, by / t_f / 1e6, t_d, fl / t_d / 1e9, by / t_d / 1e6,
           t_w, fl / t_w / 1e9, by / t_w / 1e6))
    del x, w, wt, y, dy, dx, dw, ws
print("total ms  fwd %.1f  dgrad %.1f  wgrad %.1f   (TF/s: %.1f %.1f %.1f)" %
      (tot["fwd"], tot["dgrad"], tot["wgrad"], flo / tot["fwd"] / 1e9, flo / tot["dgrad"] / 1e9, flo / tot["wgrad"] / 1e9))
