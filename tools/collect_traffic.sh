#!/bin/bash
# HBM traffic of the dominant kernel from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), counters only (no tracing).
# Run on the GPU box from the repo root:  bash tools/collect_traffic.sh
# Writes gpurun_out/traffic_fetch.csv / traffic_write.csv summaries (per kernel: calls, mean counter value).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c -d /tmp/pmc_$c -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/pmc_$c.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$c 'conv_nt_kernel<float, float, 128, 0,' > $R/gpurun_out/traffic_$c.txt
done
cat $R/gpurun_out/traffic_FETCH_SIZE.txt $R/gpurun_out/traffic_WRITE_SIZE.txt
