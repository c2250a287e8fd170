#!/bin/bash
# Same-box A/B of the several-batches-in-flight figure (bench.py `serving`) for two builds / environments, alternating, N rounds.
#   bash tools/serving_ab.sh <N> "ENV_A=.." "ENV_B=.."
N=$1; shift 1
for i in $(seq 1 $N); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s %8.3f ms/step  serving %.3f ms per batch = %.3f M mel-frames/s' % ('$cfg', d['ms_per_step'], d['serving']['ms_per_step'], d['serving']['value'] / 1e6))"
  done
done
