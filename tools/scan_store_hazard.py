#!/usr/bin/env python3
"""Scan device assembly for a store hazard hipcc does not cover on gfx950: `buffer_store_dwordx3/4 v[a:b], ..., sN offen` (a 12- /
16-byte buffer store whose scalar offset is an SGPR) followed within two instructions by a VALU write of one of its data
registers.  The compiler's hazard table has a wait state for this only when the scalar offset is an immediate; with an SGPR
it schedules the VALU write right behind the store, and on MI355X the store then wrote the VALU result for part of the
lanes (round 6: conv_p256_kernel's side output, found by tests/test_gpu_xop.py).  The library keeps such offsets in the
vector offset; this tool is the check.
usage:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinstaorder_amd/csrc -Iinclude -S --cuda-device-only X.hip -o X.s
        python tools/scan_store_hazard.py X.s [...]        (exit status 1 when a hit is found)"""
import re, sys
pat = re.compile(r'\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)(.*)')
wr = re.compile(r'\s*(v_\w+)\s+v(\d+)|\s*(v_\w+)\s+v\[(\d+):(\d+)\]')
for fn in sys.argv[1:]:
    lines = open(fn).read().split("\n")
    kern = None; hits = 0; total = 0
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ":" in l[:400] and "@" in l):
            kern = l.split(":")[0]
        m = pat.match(l)
        if not m: continue
        lo, hi, soff = int(m.group(2)), int(m.group(3)), m.group(5).rstrip(',')
        if not soff.startswith("s"): continue          # immediate / 0 / off: the compiler handles it
        total += 1
        # next 2 real instructions
        k = 0; j = i + 1
        while k < 2 and j < len(lines):
            t = lines[j].strip(); j += 1
            if not t or t.startswith(";") or t.startswith("."): continue
            k += 1
            if t.startswith("s_nop"): break
            w = wr.match(lines[j-1])
            if w:
                if w.group(2) is not None: a = b = int(w.group(2))
                else: a, b = int(w.group(4)), int(w.group(5))
                if not t.startswith("v_cmp") and a <= hi and b >= lo:
                    hits += 1
                    print(fn, kern[:90] if kern else None, "line", i + 1, l.strip(), "->", t)
    print(fn, "stores with SGPR soffset:", total, "hazard hits:", hits)
    bad = bad + hits if "bad" in dir() else hits
sys.exit(1 if bad else 0)
