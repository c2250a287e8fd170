run() { env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serving 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), {k:round(v['avg_us'],2) for k,v in d['roofline']['decode_step']['kernels'].items()})"; }
for cfg in "$@"; do run $cfg; done
