#!/bin/bash
# Same-box A/B of library variants (csrc/Makefile VARIANT=...): runs bench.py once per variant, interleaved ROUNDS times,
# and prints pairs/s, ms/step and the per-class times.  usage: tools/ab_bench.sh "<bench args>" ROUNDS variant [variant ...]
# ("default" = the in-tree libinstaorder_hip.so)
ARGS="$1"; ROUNDS="$2"; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset IO_LIB_PATH; else export IO_LIB_PATH="$ROOT/instaorder_amd/libinstaorder_hip_$v.so"; fi
    python "$ROOT/bench.py" $ARGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
kc=d.get('kernel_classes',{})
print('$v round $r: %.1f pairs/s %.2f ms |' % (d['value'], d['ms_per_step']), ' '.join('%s=%.2f' % (k.replace('conv_','').replace('_kernel',''), v['ms_per_step']) for k,v in list(kc.items())[:7]))
"
  done
done
