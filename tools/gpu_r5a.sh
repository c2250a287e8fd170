set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or give_up or headline_tile or reference_inference" 2>&1 | tail -40 > gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serving > gpurun_out/r5a/bench32.json 2> gpurun_out/r5a/bench32.err
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --batch-per-gpu 128 > gpurun_out/r5a/bench128.json 2> gpurun_out/r5a/bench128.err
GSTTACO_PERSIST_ROWS=32 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --batch-per-gpu 128 > gpurun_out/r5a/bench128_launch.json 2> gpurun_out/r5a/bench128_launch.err
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --batch-per-gpu 64 > gpurun_out/r5a/bench64.json 2> gpurun_out/r5a/bench64.err
GSTTACO_PERSIST_SPLIT16=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serving > gpurun_out/r5a/bench32_split16.json 2> gpurun_out/r5a/bench32_split16.err
for f in bench32 bench128 bench128_launch bench64 bench32_split16; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5a/$f.json"))
    print("$f", "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "step_us", round(d["roofline"]["decode_step"]["us"],2), d["library_message"][:60])
except Exception as e:
    print("$f failed", e); print(open("gpurun_out/r5a/$f.err").read()[-1500:])
PY
done
