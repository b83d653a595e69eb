#!/bin/bash
# rocprofv3 kernel-trace statistics of `bench.py <args>` for one library variant -> gpurun_out/kt_<tag>.csv
# usage: tools/kt_variant.sh <variant|default> <tag> "<bench args>"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
V="$1"; TAG="$2"; ARGS="$3"
if [ "$V" = default ]; then unset IO_LIB_PATH; else export IO_LIB_PATH="$R/instaorder_amd/libinstaorder_hip_$V.so"; fi
cd /tmp && rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt --output-format csv -- python3 $R/bench.py $ARGS --no-cpu-baseline > /tmp/kt_$TAG.log 2>&1
cp $(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kt_$TAG.csv
# per-dispatch trace, reduced to (kernel, start, end, grid) so that two variants can be compared launch by launch
python3 - "$(find /tmp/kt_$TAG -name '*kernel_trace.csv' | head -1)" $R/gpurun_out/ktd_$TAG.csv <<'PY'
import csv, re, sys
csv.field_size_limit(1 << 30)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        m = re.search(r"(\w+)(<[^(]*>)?\(", n)
        n = (m.group(1) + (m.group(2) or "")) if m else n
        f.write("%s|%d|%s\n" % (n[:80], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", "")))
PY
tail -1 /tmp/kt_$TAG.log | cut -c1-200
