cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for o in 1 0 1 0; do
  echo "overlap_gst $o: $(GSTTACO_OVERLAP_GST=$o python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*')"
done
GSTTACO_GRAPH=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
