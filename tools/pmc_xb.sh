#!/bin/bash
# PMC counters of the data-gradient launches of tools/xb_bench.py (plain vs operand-form instantiation), per dispatch.
# usage (GPU box): bash tools/pmc_xb.sh  -> gpurun_out/pmc_xb.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rm -rf /tmp/pmc_xb
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d /tmp/pmc_xb -o p --output-format csv -- python3 $R/tools/xb_bench.py fp32 > /tmp/pmc_xb.log 2>&1
f=$(find /tmp/pmc_xb -name '*counter_collection.csv' | head -1)
python3 - "$f" > $R/gpurun_out/pmc_xb.txt <<'PY'
import csv, sys, collections, re
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "conv_nt_kernel" not in n:
        continue
    m = re.search(r"conv_nt_kernel<([^>]*)>", n)
    key = (m.group(1).replace("float, float, ", ""), r["Grid_Size"])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    g = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = g.get("SQ_WAVE_CYCLES", 1)
    print("%-40s grid %-9s n=%d  mfma_busy/simd_cycles %.3f  wait_any %.3f  wait_inst %.3f  active %.3f  valu %.3g  gui %.4g" % (
        key[0], key[1], len(cs["SQ_WAVE_CYCLES"]),
        g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g.get("GRBM_GUI_ACTIVE", 1) / 8.0 * 1024.0),
        g.get("SQ_WAIT_ANY", 0) / wc, g.get("SQ_WAIT_INST_ANY", 0) / wc, g.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        g.get("SQ_INSTS_VALU", 0), g.get("GRBM_GUI_ACTIVE", 0)))
PY
cat $R/gpurun_out/pmc_xb.txt
