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
  rocprofv3 --pmc $c -d /tmp/pmc_$c -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-secondary --no-fwd-only > /tmp/pmc_$c.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$c 'conv_nt_kernel<float, float, 128, 0,' > $R/gpurun_out/traffic_$c.txt
done
cat $R/gpurun_out/traffic_FETCH_SIZE.txt $R/gpurun_out/traffic_WRITE_SIZE.txt
# matrix-pipe occupancy of the two dominant kernels: a counters-only pass of its own
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pmc_mfma -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-secondary --no-fwd-only > /tmp/pmc_mfma.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_mfma 'conv_nt_kernel<float, float, 128, 0,' > $R/gpurun_out/mfma_busy_nt.txt
python3 $R/tools/pmc_summary.py /tmp/pmc_mfma 'conv_wgrad_kernel<float, float, 128, 128' > $R/gpurun_out/mfma_busy_wgrad.txt
python3 $R/tools/pmc_summary.py /tmp/pmc_mfma 'conv_wino4_kernel' > $R/gpurun_out/mfma_busy_wino4.txt
python3 $R/tools/pmc_summary.py /tmp/pmc_mfma 'conv_wgrad_wino4_kernel' > $R/gpurun_out/mfma_busy_wgradwino4.txt
cat $R/gpurun_out/mfma_busy_nt.txt $R/gpurun_out/mfma_busy_wgrad.txt $R/gpurun_out/mfma_busy_wino4.txt $R/gpurun_out/mfma_busy_wgradwino4.txt
