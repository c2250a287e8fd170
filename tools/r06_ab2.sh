#!/bin/bash
# round 6: plain split GEMM on/off (GSTTACO_GEMM_SPLIT) and the GST fork beside the (now half-chip) encoder convolutions
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or module" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab.sh 2 "" "X=1" "GSTTACO_GST_FORK=1" "GSTTACO_GST_FORK=2" 2>&1 | grep -v amdgpu.ids
python tools/r06_enc_wino.py 2>&1 | grep -v amdgpu.ids | head -3
