#!/bin/bash
# Regenerates the measurement artefacts under gpurun_out/ on the GPU box (1x MI355X):
#   bench_n1.json            the default bench line (fp32, configs[1]) incl. roofline + cpu_baseline
#   bench_n1_bf16.json       the bf16 configuration (configs[2] shape at 256 pairs)
#   kernel_stats_raw.csv     rocprofv3 --kernel-trace --stats of the same bench command
#   traffic_*.txt            PMC HBM traffic of the dominant kernel (separate passes, see collect_traffic.sh)
#   mfma_busy_*.txt          matrix-pipe occupancy counters of the two dominant kernels (same script)
# tools/make_profiles.py (run in the repo afterwards) turns them into profiles/rNN_*.
# usage: bash tools/refresh_profiles.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 bench.py --dtype bf16 --no-cpu-baseline > $O/bench_n1_bf16.json 2>> $O/bench_n1.err
# (the timed region of every line is the unprofiled hipGraph replay since round 4: no separate "graph" lines)
python3 tools/per_launch.py fp32 256 > $O/per_launch_fp32.txt 2>> $O/bench_n1.err
python3 tools/per_launch.py bf16 256 > $O/per_launch_bf16.txt 2>> $O/bench_n1.err
for m in fwd infer; do for d in fp32 bf16; do
  python3 bench.py --mode $m --dtype $d --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/bench_n1.err | tail -1 > $O/bench_${m}_${d}.json
done; done
cd /tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/kt.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats_raw.csv
cd $R
bash tools/collect_traffic.sh > $O/traffic.log 2>&1
bash tools/collect_traffic_bf16.sh > $O/traffic_bf16.log 2>&1
# secondary workloads (BASELINE configs[2], [3], [4]); the MiDaS-based net without the per-launch event profiler too
python3 bench.py --algo InstaOrderNet_od --dtype bf16 --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench_n1.err
# the reference's own _od input_size (InstaOrderNet_od/config.yaml:35), SURVEY 8(d) secondary row
python3 bench.py --algo InstaOrderNet_od --dtype bf16 --size 384 --batch 256 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c2_384.json 2>> $O/bench_n1.err
python3 bench.py --algo InstaOrderNet_od --dtype bf16 --workload images20 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench_n1.err
python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --steps 4 --warmup 2 > $O/bench_c4_prof.json 2>> $O/bench_n1.err
# (the committed configs[4] line carries roofline + cpu_baseline: the profiled pass and the CPU leg run after the timed region)
python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --steps 10 --warmup 3 > $O/bench_c4.json 2>> $O/bench_n1.err
python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype fp32 --steps 4 --warmup 3 --no-prof --no-cpu-baseline > $O/bench_c4_fp32.json 2>> $O/bench_n1.err
cd /tmp
rm -rf /tmp/kt4
rocprofv3 --kernel-trace --stats -d /tmp/kt4 -o kt --output-format csv -- python3 $R/bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --no-prof --steps 5 --warmup 2 --no-cpu-baseline > /tmp/kt4.log 2>&1
cp $(find /tmp/kt4 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_depthnet_bf16_raw.csv
rm -rf /tmp/kt2
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/kt2.log 2>&1
cp $(find /tmp/kt2 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bf16_raw.csv
cd $R
# small per-GPU batches (the reference's own 32 pairs per GPU) and the two-rank path on ONE GPU over gloo (the product path of
# --gpus 2 with the collectives on the CPU backend: what one box can show of the distributed step -- world size, per-rank times,
# staged-overlap vs flat exchange)
python3 tools/per_launch.py fp32 32 > $O/per_launch_fp32_b32.txt 2>> $O/bench_n1.err
python3 bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-only 2>> $O/bench_n1.err | tail -1 > $O/bench_b32_fp32.json
python3 bench.py --batch 64 --steps 12 --warmup 4 --no-cpu-baseline --no-fwd-only 2>> $O/bench_n1.err | tail -1 > $O/bench_b64_fp32.json
python3 bench.py --batch 32 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-only 2>> $O/bench_n1.err | tail -1 > $O/bench_b32_bf16.json
# same-box A/B of the round-6 operand forms of the bf16 256-row kernel (IO_P256_XOP) and of the MiDaS step's residual fork / direct gradients
for v in 0 1; do IO_P256_XOP=$v python3 bench.py --dtype bf16 --no-cpu-baseline --no-prof 2>> $O/bench_n1.err | tail -1 > $O/ab_bf16_xop$v.json; done
IO_DEPTH_FORK=0 IO_DEPTH_DIRECT_GRADS=0 python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --steps 10 --warmup 3 --no-prof --no-cpu-baseline 2>> $O/bench_n1.err | tail -1 > $O/ab_c4_r5forms.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --backend gloo --batch 64 --steps 6 --warmup 3 --no-cpu-baseline --no-fwd-only 2>> $O/bench_n1.err | grep '^{' | tail -1 > $O/bench_2ranks_gloo.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29632 bench.py --gpus 2 --backend gloo --algo InstaDepthNet_od --size 384 --batch 8 --dtype bf16 --steps 5 --warmup 3 --no-cpu-baseline 2>> $O/bench_n1.err | grep '^{' | tail -1 > $O/bench_c4_2ranks_gloo.json
# the two headline lines once more with the counters of THIS run behind them (bench.py quotes profiles/rNN_pmc_*.json only while
# their csrc digest is that of the sources it runs on): make the profile files here, re-run, keep the new lines
python3 tools/make_profiles.py ${RND:-6} > /dev/null 2>> $O/bench_n1.err
python3 bench.py > $O/bench_n1.json 2>> $O/bench_n1.err
python3 bench.py --dtype bf16 --no-cpu-baseline > $O/bench_n1_bf16.json 2>> $O/bench_n1.err
tail -3 $O/bench_n1.err
