#!/bin/bash
# round 6: the split-bf16 Winograd postnet kernel against the fp32-MFMA one -- parity test + same-box timing, alternating
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_wino_split.txt
: > $OUT
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "winograd" -s 2>&1 | tail -15 >> $OUT
for i in 1 2 3; do
  GSTTACO_WINO_SPLIT=1 python tools/postnet_time.py >> $OUT 2>&1
  GSTTACO_WINO_SPLIT=0 python tools/postnet_time.py >> $OUT 2>&1
done
cat $OUT
