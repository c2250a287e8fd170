cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5e/tests.log
tail -6 gpurun_out/r5e/tests.log
for fork in 0 1 0 1; do
GSTTACO_GST_FORK=$fork timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-serving > gpurun_out/r5e/bench32_fork$fork.json 2> gpurun_out/r5e/bench32_fork$fork.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5e/bench32_fork$fork.json"))
    r=d["roofline"]
    print("fork $fork", "ms_per_step", round(d["ms_per_step"],3), "value", round(d["value"]), "step_us", round(r["decode_step"]["us"],2), "frac", round(r["frac"],3), "e2e", round(r["end_to_end_frac"],3), d["library_message"][:60])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r5e/bench32_fork$fork.err").read()[-1500:])
PY
done
