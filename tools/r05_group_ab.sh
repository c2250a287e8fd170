cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_group_ab
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_decode_launch_is_bitwise or give_up" 2>&1 | tail -15 > gpurun_out/r05_group_ab/tests.log
tail -3 gpurun_out/r05_group_ab/tests.log
timeout 300 python tools/stamps_group.py 128 > gpurun_out/r05_group_ab/stamps128.txt 2>&1
cat gpurun_out/r05_group_ab/stamps128.txt
for cfg in 128 96 64 48; do for rows in 128 32; do
GSTTACO_PERSIST_ROWS=$rows timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --batch-per-gpu $cfg > gpurun_out/r05_group_ab/bench${cfg}_$rows.json 2> gpurun_out/r05_group_ab/bench${cfg}_$rows.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_group_ab/bench${cfg}_$rows.json"))
    print("batch $cfg persist_rows $rows", "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "step_us", round(d["roofline"]["decode_step"]["us"],2), d["library_message"][:60])
except Exception as e:
    print("bench$cfg failed", e); print(open("gpurun_out/r05_group_ab/bench${cfg}_$rows.err").read()[-1500:])
PY
done; done
