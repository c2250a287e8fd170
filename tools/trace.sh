#!/bin/bash
# kernel-trace stats only:  bash tools/trace.sh <tag> [bench args]
set -u
TAG=${1:-t}; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serving "$@" > "$OUT/bench_trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = sorted(glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv")))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) >= 0.05:
        print("%-100s calls %6s avg %10.1f ns  %6s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
tail -1 "$OUT/bench_trace.log" | cut -c1-200
