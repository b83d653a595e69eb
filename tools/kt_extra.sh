#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the secondary configurations (bf16 configs[2]-shape step, InstaDepthNet_od).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp
rm -rf /tmp/kt_bf16 /tmp/kt_dn
rocprofv3 --kernel-trace --stats -d /tmp/kt_bf16 -o kt --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/kt_bf16.log 2>&1
cp $(find /tmp/kt_bf16 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bf16_raw.csv
rocprofv3 --kernel-trace --stats -d /tmp/kt_dn -o kt --output-format csv -- python3 $R/bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --no-prof --steps 5 --warmup 2 --no-cpu-baseline > /tmp/kt_dn.log 2>&1
cp $(find /tmp/kt_dn -name '*kernel_stats.csv' | head -1) $O/kernel_stats_depthnet_bf16_raw.csv
tail -1 /tmp/kt_bf16.log | cut -c1-200; tail -1 /tmp/kt_dn.log | cut -c1-200
