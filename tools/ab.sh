#!/bin/bash
# Same-box A/B of two builds (or two environments) through bench.py, alternating, N rounds.
#   bash tools/ab.sh <N> "<bench args>" "ENV_A=.. ..." "ENV_B=.. ..."       e.g.  bash tools/ab.sh 3 "" "GSTTACO_LIB=old.so" "X=1"
N=$1; ARGS=$2; shift 2
for i in $(seq 1 $N); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serving $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['decode_step']['kernels']
print('%-50s %8.3f ms/step  %s' % ('$cfg', d['ms_per_step'], {i: round(v['avg_us'], 2) for i, v in k.items()}))"
  done
done
