#!/bin/bash
# L2 hit rate per kernel (TCC_HIT_sum / TCC_MISS_sum), one PMC pass.   bash tools/l2_hit.sh <tag>
set -u
TAG=${1:-l2}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serving > "$OUT/bench_l2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(out, "pmc_l2", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(acc.items()):
    h, m = v["TCC_HIT_sum"], v["TCC_MISS_sum"]
    if h[1] and (h[0] + m[0]) > 0:
        print("%-90s n=%5d hit/launch %10.0f miss/launch %10.0f rate %.3f" % (k[:90], h[1], h[0] / h[1], m[0] / m[1], h[0] / (h[0] + m[0])))
PY
