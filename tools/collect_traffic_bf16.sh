#!/bin/bash
# bf16 counterpart of tools/collect_traffic.sh: HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) and matrix-pipe
# occupancy of the two dominant bf16 kernels over `bench.py --dtype bf16 --steps 2 --warmup 1`, counters only.
# Run on the GPU box from the repo root; writes gpurun_out/bf16_*.txt (tools/pmc_summary.py format).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
# the forward / data-gradient launches of the bf16 step run four kernel families since round 5 (conv_p256.hip, conv_halo3.hip's
# two kernels, and conv_nt_kernel for what is left); the filter gradients one
# (round 6: ", false, 1>" / ", false, 2>" = the instantiations of conv_p256_kernel that transform their A operand in LDS -- XOP =
# 1: the affine forms, 2: bn_apply's form -- as families of their own NEXT TO the whole conv_p256_kernel family)
FAMS=('conv_p256_kernel' 'conv_halo3_kernel' 'stem_halo_kernel' 'conv_nt_kernel<unsigned short, unsigned short' 'conv_wgrad_bf16_tr_kernel<128, 128'
      'conv_wgrad_halo3_kernel' 'stem_wgrad_halo_kernel' ', false, 1>' ', false, 2>' 'bn_bwd_apply_kernel' 'bn_apply_kernel')
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  rocprofv3 --pmc $c -d /tmp/pmcb_$c -o t --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/pmcb_$c.log 2>&1
  : > $R/gpurun_out/bf16_traffic_$c.txt
  for f in "${FAMS[@]}"; do python3 $R/tools/pmc_summary.py /tmp/pmcb_$c "$f" >> $R/gpurun_out/bf16_traffic_$c.txt; done
done
rm -rf /tmp/pmcb_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pmcb_mfma -o t --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/pmcb_mfma.log 2>&1
: > $R/gpurun_out/bf16_mfma_busy.txt $R/gpurun_out/bf16_lds.txt
for f in "${FAMS[@]}"; do python3 $R/tools/pmc_summary.py /tmp/pmcb_mfma "$f" >> $R/gpurun_out/bf16_mfma_busy.txt; done
# LDS counters of the same families (bank conflicts of the in-LDS operand transform against the plain kernel)
rm -rf /tmp/pmcb_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/pmcb_lds -o t --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/pmcb_lds.log 2>&1
: > $R/gpurun_out/bf16_lds.txt
for f in "${FAMS[@]}"; do python3 $R/tools/pmc_summary.py /tmp/pmcb_lds "$f" >> $R/gpurun_out/bf16_lds.txt; done
cat $R/gpurun_out/bf16_traffic_FETCH_SIZE.txt $R/gpurun_out/bf16_traffic_WRITE_SIZE.txt $R/gpurun_out/bf16_mfma_busy.txt $R/gpurun_out/bf16_lds.txt
