#!/bin/bash
# Same-box A/B of the postnet time (roofline.postnet.ms of bench.py) for two builds / environments, alternating, N rounds.
#   bash tools/postnet_ab.sh <N> "<bench args>" "ENV_A=.." "ENV_B=.."
N=$1; ARGS=$2; shift 2
for i in $(seq 1 $N); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serving $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s %8.3f ms/step  postnet %.3f ms' % ('$cfg', d['ms_per_step'], d['roofline']['postnet']['ms']))"
  done
done
