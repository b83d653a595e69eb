#!/usr/bin/env python3
"""The fp32 stem forward (exact-K, fused BatchNorm statistics) launched REPS times at N x S x S -- for rocprofv3 --pmc /
--kernel-trace on that one kernel; mode wgrad: its filter gradient.
usage: python tools/one_stem.py [N=512] [S=256] [reps=10] [fwd|wgrad]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
MODE = sys.argv[4] if len(sys.argv) > 4 else "fwd"
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
Ho, G, Co = S // 2, 2, 64
x = torch.randn(N, S, S, 8, device="cuda")
x[..., 5:] = 0
w = torch.randn(Co, 49, 8, device="cuda") / 15.0
w[..., 5:] = 0
y = torch.empty(N, Ho, Ho, Co, device="cuda")
gamma, beta = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
mean, rstd, scale, shift = (torch.empty(G * Co, device="cuda") for _ in range(4))
nws = L.io_conv2d_bnstats_workspace_floats(N, S, S, Co, 7, 7, 2, 3, G)
ws = torch.empty(nws, device="cuda")
packed = torch.empty(L.io_stem_packed_floats(5), device="cuda")
run = lambda: L.io_stem_fwd_bnstats_exact(P(x), P(w), P(y), N, S, S, 5, G, P(gamma), P(beta), P(rm), P(rv), 0.1, 1e-5,
                                          P(mean), P(rstd), P(scale), P(shift), P(ws), nws, P(packed), ST())
if MODE == "wgrad":
    dy = torch.randn(N, Ho, Ho, Co, device="cuda")
    dw = torch.empty(Co, 49, 8, device="cuda")
    nb = L.io_stem_wgrad_exact_workspace_bytes(N, S, S, 5)
    wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    run = lambda: L.io_stem_wgrad_exact(P(x), P(dy), P(dw), N, S, S, 5, P(wsb), nb, P(packed), ST())
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
fl = 2.0 * N * Ho * Ho * Co * 245
print("%.4f ms per call (%s)  %.1f TF/s" % (ms, "gradient + reduction + unpack" if MODE == "wgrad" else "pack + conv + statistics merge", fl / ms / 1e9))
