#!/bin/bash
# postnet alone (bench.py's roofline.postnet.ms) for two builds / environments, alternating.   bash tools/postnet_ab.sh <N> "<bench args>" "ENV_A" "ENV_B"
N=$1; ARGS=$2; shift 2
for i in $(seq 1 $N); do for cfg in "$@"; do
  env $cfg python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serving $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['roofline']['postnet']
print('%-50s postnet %.3f ms  %.0f TF  (step %.3f ms)' % ('$cfg', p['ms'], p['TFLOP/s'], d['ms_per_step']))"
done; done
