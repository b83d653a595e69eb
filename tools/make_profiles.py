#!/usr/bin/env python3
"""gpurun_out/ artefacts of tools/refresh_profiles.sh -> profiles/rNN_* (run in the repo after the GPU call).
usage: python tools/make_profiles.py [round]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rnd = "r%02d" % (int(sys.argv[1]) if len(sys.argv) > 1 else 1)


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


for src, dst in (("bench_n1.json", "bench_n1.json"), ("bench_n1_bf16.json", "bench_n1_bf16.json"),
                 ("bench_n1_graph.json", "bench_n1_hipgraph.json"),
                 ("bench_n1_bf16_graph.json", "bench_n1_bf16_hipgraph.json")):
    if os.path.exists(os.path.join(G, src)):
        json.dump(last_json(os.path.join(G, src)), open(os.path.join(P, "%s_%s" % (rnd, dst)), "w"), indent=1)

for m in ("fwd", "infer"):
    for d in ("fp32", "bf16"):
        src = os.path.join(G, "bench_%s_%s.json" % (m, d))
        if os.path.exists(src):
            json.dump(last_json(src), open(os.path.join(P, "%s_bench_%s_%s.json" % (rnd, m, d)), "w"), indent=1)

raw = os.path.join(G, "kernel_stats_raw.csv")
if os.path.exists(raw):
    csv.field_size_limit(1 << 30)
    rows = list(csv.DictReader(open(raw)))
    with open(os.path.join(P, rnd + "_bench_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  "
                "(1x MI355X, 7 steps incl. warm-up)\n# kernel names shortened (namespaces / argument lists dropped); "
                "durations in ns\nName,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            n = r["Name"]
            m = re.search(r"(\w+)(<[^(]*>)?\(", n)
            short = (m.group(1) + (m.group(2) or "")) if m else n
            short = re.sub(r"\(anonymous namespace\)::", "", short)[:90]
            f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                                  r["MinNs"], r["MaxNs"]))

vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(G, "traffic_%s.txt" % c)
    if os.path.exists(p):
        m = re.search(r"%s: n=(\d+) mean=([0-9.e+]+)" % c, open(p).read())
        if m:
            vals[c] = (int(m.group(1)), float(m.group(2)))
if len(vals) == 2:
    b = last_json(os.path.join(G, "bench_n1.json"))
    fz, wz = vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]
    json.dump({
        "_comment": "HBM traffic of the dominant kernel from rocprofv3 --pmc (tools/collect_traffic.sh: separate "
                    "FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps 2 --warmup 1`, counters only). Units: the "
                    "counters report KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts wide "
                    "coalesced reads at half their size, so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; the raw "
                    "(uncorrected) sum is kept alongside.",
        "kernel": b["roofline"]["kernel"], "launches_profiled": vals["FETCH_SIZE"][0],
        "fetch_size_kib_mean": fz, "write_size_kib_mean": wz,
        "bytes_per_launch_corrected": int((2 * fz + wz) * 1024), "bytes_per_launch_raw": int((fz + wz) * 1024),
        "algorithmic_bytes_per_launch": int(b["kernel_classes"][b["roofline"]["kernel"]]["gbs"] * 1e6
                                            * b["roofline"]["avg_launch_ms"]),
    }, open(os.path.join(P, rnd + "_pmc_traffic.json"), "w"), indent=1)
print("profiles refreshed:", sorted(os.listdir(P)))
