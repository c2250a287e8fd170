#!/bin/bash
# knob sweep at the other BASELINE batch shapes.   bash tools/r03_sweep.sh "<bench args>" "ENV=.. ENV=.." "ENV=.." ...
ARGS=$1; shift
for cfg in "$@"; do
    env $cfg python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-serving $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['decode_step']['kernels']
print('%-60s %8.3f ms/step  %s' % ('$cfg', d['ms_per_step'], {i: round(v['avg_us'], 2) for i, v in k.items()}))"
done
