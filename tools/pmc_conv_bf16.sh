export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/p_$n
  rocprofv3 --pmc $c -d /tmp/p_$n -o t --output-format csv -- python3 $R/tools/conv_bench.py 512 256 1 bf16 > /tmp/p_$n.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/p_$n > $R/gpurun_out/pmc_bf16_$n.txt
done
ls -la $R/gpurun_out/pmc_bf16_*
