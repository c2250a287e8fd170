cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
for a in "32 128" "8 187" "32 187" "32 256" "8 256"; do timeout 200 python tools/stamps_persist.py $a 2>&1 | grep -v amdgpu.ids | grep "batch\|average step" ; done > gpurun_out/r5f/stamps_tv.txt
cat gpurun_out/r5f/stamps_tv.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5f/bench32_serving.json 2> gpurun_out/r5f/bench32_serving.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5f/bench32_serving.json"))
print("ms_per_step", d["ms_per_step"], json.dumps(d["serving"])[:900])
PY
