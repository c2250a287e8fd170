cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_decode_launch_is_bitwise or give_up" 2>&1 | tail -15 > gpurun_out/r5c/tests.log
tail -4 gpurun_out/r5c/tests.log
timeout 300 python tools/stamps_group.py 128 > gpurun_out/r5c/stamps128.txt 2>&1
timeout 300 python tools/stamps_group.py 64 > gpurun_out/r5c/stamps64.txt 2>&1
cat gpurun_out/r5c/stamps128.txt gpurun_out/r5c/stamps64.txt
for cfg in "128" "64" ; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --batch-per-gpu $cfg > gpurun_out/r5c/bench$cfg.json 2> gpurun_out/r5c/bench$cfg.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5c/bench$cfg.json"))
    print("bench$cfg", "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "step_us", round(d["roofline"]["decode_step"]["us"],2), d["library_message"][:60])
except Exception as e:
    print("bench$cfg failed", e); print(open("gpurun_out/r5c/bench$cfg.err").read()[-1500:])
PY
done
