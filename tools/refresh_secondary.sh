#!/bin/bash
# Secondary configurations of DESIGN.md section 5 (BASELINE configs[2..4] + the device input pipeline), one GPU.
# Writes gpurun_out/sec_*.json; copy into profiles/ by hand (names in DESIGN.md).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py --algo InstaOrderNet_od --dtype bf16 --batch 1024 --steps 5 --warmup 2 --no-cpu-baseline 2>>$O/sec.err | tail -1 > $O/sec_config3_od_bf16_1024.json
python3 bench.py --algo InstaOrderNet_od --dtype bf16 --workload images20 --steps 10 --warmup 3 --no-cpu-baseline 2>>$O/sec.err | tail -1 > $O/sec_config4_images20.json
python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --no-prof --steps 10 --warmup 3 --no-cpu-baseline 2>>$O/sec.err | tail -1 > $O/sec_config5_depthnet_bf16.json
python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype fp32 --no-prof --steps 6 --warmup 2 --no-cpu-baseline 2>>$O/sec.err | tail -1 > $O/sec_config5_depthnet_fp32.json
python3 bench.py --host-inputs u8 --steps 8 --warmup 3 --no-cpu-baseline 2>>$O/sec.err | tail -1 > $O/sec_u8_pipeline.json
python3 bench.py --host-inputs pinned --steps 8 --warmup 3 --no-cpu-baseline 2>>$O/sec.err | tail -1 > $O/sec_pinned.json
tail -2 $O/sec.err
