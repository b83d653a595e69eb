#!/usr/bin/env python3
"""Run one conv shape (fwd, dgrad, wgrad) a few times -- target for rocprofv3 --pmc runs.
usage: conv_probe.py N H Cin Cout k stride pad [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N, Hh, Cin, Cout, k, st, pad = [int(v) for v in sys.argv[1:8]]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 3
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
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
for _ in range(reps):
    L.io_conv2d_fwd(P(x), P(w), P(y), N, Hh, Hh, Cin, Cout, k, k, st, pad, ST())
    L.io_conv2d_dgrad(P(dy), P(wt), P(dx), None, None, N, Hh, Hh, Cin, Cout, k, k, st, pad, ST())
    L.io_conv2d_wgrad(P(x), P(dy), P(dw), N, Hh, Hh, Cin, Cout, k, k, st, pad, P(ws), nb, ST())
torch.cuda.synchronize()
