#!/bin/bash
# Same-box A/B of two builds at the other BASELINE batch shapes: configs[2]-sized batch 128 (fp32) and the configs[4] shard
# (batch 64, bf16 mixed).   bash tools/r03_batch_ab.sh <old .so> [out file]
OLD=${1:-gst_tacotron_amd/lib/libgsttaco_r02.so}
OUT=${2:-gpurun_out/r03_batch_ab.txt}
run() { # label, env..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serving "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['decode_step']['kernels']
print('%-28s %8.3f ms/step  %7.3f M frames/s   decode launches (event-bracketed us): %s' % ('$label', d['ms_per_step'], d['value'] / 1e6, {i: round(v['avg_us'], 2) for i, v in k.items()}))"
}
{
for rep in 1 2; do
run "old  B=128 fp32" GSTTACO_LIB=$OLD -- --batch-per-gpu 128
run "new  B=128 fp32" X=1 -- --batch-per-gpu 128
run "old  B=64 bf16" GSTTACO_LIB=$OLD -- --batch-per-gpu 64 --mixed
run "new  B=64 bf16" X=1 -- --batch-per-gpu 64 --mixed
run "old  B=64 fp32" GSTTACO_LIB=$OLD -- --batch-per-gpu 64
run "new  B=64 fp32" X=1 -- --batch-per-gpu 64
run "old  B=32 fp32" GSTTACO_LIB=$OLD --
run "new  B=32 fp32" X=1 --
done
} 2>&1 | tee $OUT
