#!/bin/bash
# PMC counters of the Winograd forward kernels on the 3x3 256 -> 256 layer of the bench batch (and the direct kernel beside
# it): matrix-pipe occupancy, issue stalls, LDS conflicts.  usage: bash tools/pmc_wino.sh  -> gpurun_out/pmc_wino_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/pmc_wino4.txt $R/gpurun_out/pmc_direct3x3.txt
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
KERNEL=conv_wino4_kernel bash $R/tools/pmc_one.sh wino4 "512 16 256 256 3 1 1 fp32 10 wino" "$P1" "$P2"
KERNEL=conv_nt_kernel bash $R/tools/pmc_one.sh direct3x3 "512 16 256 256 3 1 1 fp32 10 fwd" "$P1" "$P2"
cat $R/gpurun_out/pmc_wino4.txt $R/gpurun_out/pmc_direct3x3.txt
