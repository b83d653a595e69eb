#!/usr/bin/env python3
"""Run-to-run determinism of the eager (no hipGraph) InstaDepthNet_od training step, with and without ops.WeightPlan:
R repetitions of three steps from the same seeded state; prints which parameters differ between repetitions (if any).
usage: python tools/depth_eager_race.py [R]      (IO_DEPTH_STREAMS=0|1 selects the single- / multi-stream forward)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_midas as T
from helpers import synthetic

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
DIRTY = sys.argv[2] if len(sys.argv) > 2 else ""      # "nan" | "big": fill the allocator's cached blocks before each run


def dirty():
    """Blocks of many sizes written with NaN (or 1e30) on the main and the side streams, then returned to the caching
    allocator: a kernel that reads memory nobody wrote now sees that instead of the zero pages of a fresh hipMalloc."""
    if not DIRTY:
        return
    from instaorder_amd import midas_net
    val = float("nan") if DIRTY == "nan" else 1e30
    streams = [torch.cuda.current_stream()] + list(midas_net._InstaDepthBase._side.get(torch.cuda.current_device()) or [])
    for st in streams:
        with torch.cuda.stream(st):
            keep = []
            for sh in range(9, 29):                      # 512 B .. 256 MiB
                for _ in range(6 if sh < 24 else 2):
                    keep.append(torch.full(((1 << sh) // 4,), val, device="cuda"))
            del keep
    if "huge" in sys.argv:                               # what a bench-size test leaves behind: multi-GiB cached blocks
        keep = [torch.full((1 << 30,), val, device="cuda") for _ in range(8)]       # 8 x 4 GiB
        del keep
    torch.cuda.synchronize()
PRE = [a[4:] for a in sys.argv if a.startswith("pre=")]
for pre in PRE:                                        # what ran in the process before (bisecting an in-suite failure)
    if pre == "pytest":
        import pytest
        pytest.main([os.path.join(ROOT, "tests", "test_gpu_bench_scale.py"), "-q", "-m", "gpu", "-k", "test_pooling_and"])
    elif pre == "empty_cache":
        x = torch.randn(1 << 20, device="cuda")
        del x
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    elif pre == "randn":
        x = torch.randn(1 << 20, device="cuda")
        torch.cuda.synchronize()
    elif pre == "lib":
        from instaorder_amd import _lib
        _lib.require_gpu()
        _lib.lib()
algo, tag = T.CASES[0]
g, spec = T.load(tag)
S, B, seed = (int(v) for v in g["meta"])


def run(planned):
    m, _ = T.build(algo, g, spec)
    m._use_graph = False
    m.net._side_streams(2)
    dirty()
    if not planned:
        m._wplan = False
    m.switch_to("train")
    grads = []
    for it in range(3):
        t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(seed + 700 + it, B, S).items()}
        T.feed(m, algo, t)
        m.step()
        grads.append(m.optim.flat_grads.clone())
    torch.cuda.synchronize()
    names = [(n, off, k) for (n, p), (off, k) in zip(m.model.named_parameters(), m.optim._spans)]
    nan = [n for n, off, k in names if not bool(torch.isfinite(m.optim.flat_params[off:off + k]).all())]
    if nan:
        print("  planned=%s: %d parameters not finite, first: %s" % (planned, len(nan), nan[:6]))
    return m.optim.flat_params.clone(), grads, names


def diff(a, b, names, what):
    if torch.equal(a, b):
        return 0
    d = (a - b).abs()
    bad = [(n, int((d[off:off + k] > 0).sum()), k, float(d[off:off + k].max()), float(a[off:off + k].abs().max()))
           for n, off, k in names if bool((d[off:off + k] > 0).any())]
    print("  %s: %d parameters differ" % (what, len(bad)))
    groups = {}
    for row in bad:
        key = ".".join(row[0].split(".")[1:4] if row[0].startswith("module.pretrained") else row[0].split(".")[1:3])
        groups[key] = groups.get(key, 0) + 1
    print("     by module:", " ".join("%s=%d" % kv for kv in sorted(groups.items())))
    for row in bad[:12]:
        print("     %-60s %d of %d elements, max |d| %.3e (max |v| %.3e)" % row)
    return len(bad)


print("IO_DEPTH_STREAMS =", os.environ.get("IO_DEPTH_STREAMS", "1"))
base = {}
for planned in (False, True):
    for r in range(R):
        p, gs, names = run(planned)
        if planned not in base:
            base[planned] = (p, gs)
            continue
        p0, g0 = base[planned]
        n = sum(diff(a, b, names, "planned=%s rep %d step-%d gradients" % (planned, r, i)) for i, (a, b) in enumerate(zip(g0, gs)))
        n += diff(p0, p, names, "planned=%s rep %d parameters after 3 steps" % (planned, r))
        print("planned=%s rep %d: %s" % (planned, r, "identical" if n == 0 else "DIFFERENT"))
n = sum(diff(a, b, names, "plan vs per-call step-%d gradients" % i) for i, (a, b) in enumerate(zip(base[False][1], base[True][1])))
n += diff(base[False][0], base[True][0], names, "plan vs per-call parameters")
print("plan vs per-call:", "identical" if n == 0 else "DIFFERENT")
truth = os.environ.get("IO_TRUTH")
if truth and not os.path.exists(truth):
    torch.save({k: v[0].cpu() for k, v in base.items()}, truth)
elif truth:
    ref = torch.load(truth)
    for k in (False, True):
        n = diff(ref[k].cuda(), base[k][0], names, "planned=%s against the clean process" % k)
        print("planned=%s vs clean process:" % k, "identical" if n == 0 else "DIFFERENT")
