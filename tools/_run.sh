cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
rm -rf gpurun_out/ab; mkdir -p gpurun_out/ab
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab/log
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(sorted(glob.glob('gpurun_out/ab/*/*_kernel_stats.csv'))[-1])))
for r in rows[:11]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ("%.1f"%float(r['AverageNs'])).rjust(10), r['Percentage'].rjust(6))
PY
