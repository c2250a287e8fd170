cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_decode_launch_is_bitwise" 2>&1 | tail -5 > gpurun_out/r5i/tests.log
tail -3 gpurun_out/r5i/tests.log
timeout 300 python tools/stamps_group.py 64 --mixed 2>&1 | grep -v amdgpu > gpurun_out/r5i/stamps64m.txt
cat gpurun_out/r5i/stamps64m.txt
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --batch-per-gpu 64 --mixed > gpurun_out/r5i/bench64m.json 2> gpurun_out/r5i/bench64m.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5i/bench64m.json"))
print("batch 64 mixed", "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "step_us", round(d["roofline"]["decode_step"]["us"],2), d["library_message"][:60])
PY
