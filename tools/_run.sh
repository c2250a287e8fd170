cd $GRAFT_REPO_ROOT
for k in 1 0 1 0 1 0; do
  echo "keep_x $k: $(GSTTACO_KEEP_X=$k python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']['decode_step']['kernels']; print(round(d['ms_per_step'],2), {k: round(v['avg_us'],1) for k,v in r.items()})
")"
done
