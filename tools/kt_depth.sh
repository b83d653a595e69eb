export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/ktd
rocprofv3 --kernel-trace --stats -d /tmp/ktd -o kt --output-format csv -- python3 $R/bench.py --algo InstaDepthNet_od --batch 16 --size 384 --steps 4 --warmup 2 --dtype bf16 --no-prof > /tmp/ktd.log 2>&1
cp $(find /tmp/ktd -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kt_depth_stats.csv
tail -2 /tmp/ktd.log
