#!/bin/bash
# kernel-trace stats only (one rocprofv3 pass).   bash tools/trace_only.sh <tag> [bench args]
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serving "$@" > "$OUT/bench.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
st = sorted(glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv")))
rows = list(csv.DictReader(open(st[-1])))
with open(os.path.join("gpurun_out", tag + "_kernel_stats.csv"), "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows:
        if float(r["Percentage"]) >= 0.05:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
            print("%-110s %7s %9.2f us %6s %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep '^{' "$OUT/bench.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step under rocprof', d['ms_per_step'])"
