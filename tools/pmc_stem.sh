#!/bin/bash
# PMC counters of the stem forward kernel: tools/pmc_stem.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG="$1"; shift 1
cd /tmp
i=0
for CNT in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  rocprofv3 --pmc $CNT -d /tmp/pmc_${TAG}_$i -o p --output-format csv -- python3 $R/tools/one_stem.py ${STEM_ARGS:-512 256 10 fwd} > /tmp/pmc_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" "${KERNEL:-stem_rows_kernel}" <<'PY' >> $R/gpurun_out/pmc_$TAG.txt
import csv, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[3:] if len(v) > 3 else v
    print("%-40s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
done
tail -1 /tmp/pmc_${TAG}_1.log >> $R/gpurun_out/pmc_$TAG.txt
