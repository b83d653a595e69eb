#!/usr/bin/env python3
"""Cost of the fused input transform (BatchNorm + ReLU applied to the A operand while it is staged,
io_conv2d_fwd_xf_dt) against the plain forward convolution, for the conv2 / conv3 shapes of ResNet-50 at the bench
batch -- and against what it replaces: the separate bn_apply pass over the same tensor.
usage: python tools/xf_bench.py [N] [S] [reps] [fp32|bf16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
DT = 1 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else 0
TD = torch.bfloat16 if DT else torch.float32
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS)
    return best


shapes = []
H = S // 4
for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
    for b in range(blocks):
        stride = 2 if (b == 0 and li > 0) else 1
        shapes.append((H, planes, planes, 3, stride, 1))            # conv2 reads relu(bn1(y1))
        shapes.append((H // stride, planes, planes * 4, 1, 1, 0))   # conv3 reads relu(bn2(y2))
        H //= stride
uniq = {}
for s in shapes:
    uniq[s] = uniq.get(s, 0) + 1
G = 2
tp = tx = ta = 0.0
print("%4s %5s %5s k s cnt | plain ms | xf ms  (+%%) | bn_apply ms of the same input" % ("H", "Cin", "Cout"))
for (Hh, Cin, Cout, k, st, pad), cnt in uniq.items():
    Ho = (Hh + 2 * pad - k) // st + 1
    x = torch.randn(N, Hh, Hh, Cin, device="cuda").to(TD)
    w = (torch.randn(Cout, k * k, Cin, device="cuda") * 0.05).to(TD)
    y = torch.empty(N, Ho, Ho, Cout, device="cuda", dtype=TD)
    a = torch.empty_like(x)
    sc = torch.rand(G * Cin, device="cuda") + 0.5
    sh = torch.randn(G * Cin, device="cuda") * 0.3
    mean = torch.zeros(G * Cin, device="cuda")
    t_p = timeit(lambda: L.io_conv2d_fwd_dt(P(x), P(w), P(y), N, Hh, Hh, Cin, Cout, k, k, st, pad, DT, DT, ST()))
    t_x = timeit(lambda: L.io_conv2d_fwd_xf_dt(P(x), P(w), P(y), N, Hh, Hh, Cin, Cout, k, k, st, pad, G, P(mean), P(sc),
                                               P(sh), None, None, None, None, 0.1, 1e-5, None, None, None, None, None, 0,
                                               DT, ST()))
    t_a = timeit(lambda: L.io_bn_apply_dt(P(x), N * Hh * Hh, Cin, G, 1, P(mean), P(sc), P(sh), None, None, None, None, 1,
                                          P(a), DT, ST()))
    tp += cnt * t_p
    tx += cnt * t_x
    ta += cnt * t_a
    print("%4d %5d %5d %d %d %3d | %8.3f | %6.3f (%+5.1f) | %6.3f" % (Hh, Cin, Cout, k, st, cnt, t_p, t_x,
                                                                     100 * (t_x / t_p - 1), t_a))
    del x, w, y, a
print("per pass: plain %.2f ms, with transform %.2f ms (+%.2f), bn_apply passes replaced %.2f ms -> net %.2f ms"
      % (tp, tx, tx - tp, ta, tx - tp - ta))
