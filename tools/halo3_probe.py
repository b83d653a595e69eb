#!/usr/bin/env python3
"""Timing of the narrow 3x3 bf16 launches (csrc/conv_halo3.hip against conv_p256 / conv_nt_kernel): plain forward.
usage: python tools/halo3_probe.py [N] [H] [C]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instaorder_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Cc = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(N, H, H, Cc, device="cuda").to(torch.bfloat16)
w = (torch.randn(Cc, 9, Cc, device="cuda") / (3 * Cc ** 0.5)).to(torch.bfloat16)
y = torch.empty(N, H, H, Cc, device="cuda", dtype=torch.bfloat16)
for mode in (1, 2, 0):
    L.io_set_bf16_p256(mode)
    def fn():
        _lib.check(L.io_conv2d_fwd_dt(P(x), P(w), P(y), N, H, H, Cc, Cc, 3, 3, 1, 1, 1, 1, ST()), "fwd")
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    M = N * H * H
    print("mode %d route %d: %.3f ms  %.0f TF/s  %.2f TB/s" % (mode, L.io_debug_last_nt_route(), ms, 2.0 * M * Cc * 9 * Cc / ms / 1e9,
                                                           4.0 * M * Cc / ms / 1e9))
