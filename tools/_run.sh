cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/stamps.py 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_us": [0-9.]*'
