#!/usr/bin/env python3
"""Compact table of hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin or a file): one line per kernel with
its demangled template arguments, VGPR / AGPR / SGPR counts, spills, occupancy.  usage:
  hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2>&1 | tools/kres.py [filter]"""
import re
import subprocess
import sys

txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"),
                     ("vsp", r"VGPRs Spill: (\d+)"), ("ssp", r"SGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("scr", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), text=True,
                       capture_output=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n).replace("unsigned short", "bf16").replace("void ", "")
    if flt and flt not in n:
        continue
    print("%-78s v%3d a%3d s%3d spill v%d s%d scr %d occ %d" % (n[:78], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1),
                                                       r.get("vsp", 0), r.get("ssp", 0), r.get("scr", 0), r.get("occ", -1)))
