#!/bin/bash
# One Inference_Step's dispatch-by-dispatch timeline (kernels + copies, start / duration / gap to the previous end) from a rocprofv3
# kernel + memory-copy trace of bench.py: where the non-decode share of the call goes.   bash tools/timeline.sh <tag> [bench args]
TAG=${1:-t}; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/timeline_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-serving "$@" > "$OUT/bench.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob(os.path.join(out, "trace", "*", "*_memory_copy_trace.csv")):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the timed steps: find the persistent decode launches, take the second to last one's call = from the previous postnet end to this call's last copy
dec = [i for i, e in enumerate(ev) if "persist_decode" in e[2] and "init" not in e[2]]
if len(dec) < 3:
    print("no persistent decode launches found", len(ev)); sys.exit(0)
a, b = dec[-3], dec[-2]
# a call = everything after the previous decode's postnet ... simpler: events between decode[-3] (exclusive) and decode[-2] (inclusive) + what follows until the next call's first event
lines = []
prev_end = ev[a][1]
t0 = None
with open(os.path.join("gpurun_out", tag + "_timeline.txt"), "w") as f:
    for i in range(a, min(b + 1, len(ev))):
        s, e, n = ev[i]
        if t0 is None: t0 = s
        ln = "%10.1f us  dur %9.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:120])
        prev_end = max(prev_end, e)
        print(ln); f.write(ln + "\n")
PY
grep '^{' "$OUT/bench.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step under rocprof', d['ms_per_step'])"
