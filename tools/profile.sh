#!/bin/bash
# Round profiling recipe (run on the GPU box through gpurun): kernel-trace stats + HBM PMC counters in separate passes.
#   bash tools/profile.sh r01 [extra bench.py arguments, e.g. --batch-per-gpu 128]
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serving $EXTRA > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serving $EXTRA > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serving $EXTRA > "$OUT/bench_write.log" 2>&1
# MFMA utilisation: SQ (8 slots) and GRBM (2 slots) are independent blocks, one pass
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serving $EXTRA > "$OUT/bench_mfma.log" 2>&1
python3 tools/summarize_profile.py "$OUT" "$TAG"
