cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_final
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_final/tests.log
tail -6 gpurun_out/r06_final/tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 600 python bench.py > gpurun_out/r06_final/bench.json 2> gpurun_out/r06_final/bench.err; python - <<PY
import json
d=json.load(open("gpurun_out/r06_final/bench.json"))
r=d["roofline"]
print("default bench: ms", round(d["ms_per_step"],3), "value", round(d["value"]), "frac", round(r["frac"],3), "e2e", round(r["end_to_end_frac"],3), "serving", round(d["serving"]["value"]), "cpu", round(d["cpu_baseline"]["value"]), "fallback", d["fallback_taken"])
PY
