#!/usr/bin/env python3
"""One forward convolution shape, launched REPS times (for rocprofv3 --pmc / --kernel-trace on a single kernel).
usage: python tools/one_conv.py N H Cin Cout k stride pad fp32|bf16 [reps] [fwd|wgrad|wino]
(wino: the 3x3 stride-1 forward through io_conv2d_fwd_wino -- the Winograd row form)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N, H, Cin, Cout, k, st, pad = (int(v) for v in sys.argv[1:8])
DT = 1 if sys.argv[8] == "bf16" else 0
REPS = int(sys.argv[9]) if len(sys.argv) > 9 else 10
MODE = sys.argv[10] if len(sys.argv) > 10 else "fwd"
TD = torch.bfloat16 if DT else torch.float32
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
Ho = (H + 2 * pad - k) // st + 1
x = torch.randn(N, H, H, Cin, device="cuda").to(TD)
w = (torch.randn(Cout, k * k, Cin, device="cuda") * 0.05).to(TD)
y = torch.empty(N, Ho, Ho, Cout, device="cuda", dtype=TD)
if MODE == "wgrad":
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(TD)
    dw = torch.empty(Cout, k * k, Cin, device="cuda")
    nb = L.io_conv2d_wgrad_workspace_bytes(N, H, H, Cin, Cout, k, k, st, pad)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    run = lambda: L.io_conv2d_wgrad_dt(P(x), P(dy), P(dw), N, H, H, Cin, Cout, k, k, st, pad, P(ws), nb, DT, DT, S())
elif MODE == "wino":
    nsc = L.io_conv2d_wino_scratch_floats(Cin, Cout)
    sc = torch.empty(nsc, device="cuda")
    Z = C.c_void_p(0)
    run = lambda: L.io_conv2d_fwd_wino(P(x), P(w), P(y), N, H, H, Cin, Cout, 1, Z, Z, Z, Z, Z, Z, Z, 0.1, 1e-5, Z, Z, Z, Z, Z, 0,
                                       P(sc), nsc, S())
else:
    run = lambda: L.io_conv2d_fwd_dt(P(x), P(w), P(y), N, H, H, Cin, Cout, k, k, st, pad, DT, DT, S())
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(REPS):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / REPS
fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
by = (2 if DT else 4) * (N * H * H * Cin + N * Ho * Ho * Cout + Cout * Cin * k * k)
print("%.4f ms  %.1f TF/s  %.0f GB/s (algorithmic)" % (ms, fl / ms / 1e9, by / ms / 1e6))
