# PMC counters of the narrow 3x3 bf16 launches (tools/halo3_probe.py): separate passes, counters only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp
for c in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/p_$n
  rocprofv3 --pmc $c -d /tmp/p_$n -o t --output-format csv -- python3 $R/tools/halo3_probe.py $@ > /tmp/p_$n.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/p_$n >> $R/gpurun_out/pmc_halo3.txt
done
