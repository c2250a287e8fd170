#!/bin/bash
# The round's committed evidence: bench line + rocprofv3 kernel stats + HBM / MFMA counters for the headline configuration
# (BASELINE configs[1]) and the two other single-GPU shapes (configs[2]-sized batch 128 fp32; the configs[4] shard, batch 64 bf16).
#   bash tools/r03_profiles.sh <tag>         e.g. r03a -> gpurun_out/prof_r03a_cfg{2,3,5}/summary/*
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/profile.sh ${TAG}_cfg2 > gpurun_out/${TAG}_cfg2_profile.log 2>&1
bash tools/profile.sh ${TAG}_cfg3 --batch-per-gpu 128 > gpurun_out/${TAG}_cfg3_profile.log 2>&1
bash tools/profile.sh ${TAG}_cfg5 --batch-per-gpu 64 --mixed > gpurun_out/${TAG}_cfg5_profile.log 2>&1
python bench.py > gpurun_out/${TAG}_cfg2_bench.json 2> gpurun_out/${TAG}_cfg2_bench.err
python bench.py --batch-per-gpu 128 --no-serving > gpurun_out/${TAG}_cfg3_bench.json 2> gpurun_out/${TAG}_cfg3_bench.err
python bench.py --batch-per-gpu 64 --mixed --no-serving > gpurun_out/${TAG}_cfg5_bench.json 2> gpurun_out/${TAG}_cfg5_bench.err
python tools/stamps.py > gpurun_out/${TAG}_stamps.txt 2>&1
python tools/stamps_batch.py 128 > gpurun_out/${TAG}_stamps_b128.txt 2>&1
bash tools/l2_hit.sh ${TAG}_l2 > gpurun_out/${TAG}_l2_hit.txt 2>&1
for c in cfg2 cfg3 cfg5; do python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_${c}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("${c}", round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 3), "M frames/s | frac", round(r["frac"], 3), "rocprof", r.get("frac_rocprofv3"), "| step", r["step_bound"], round(r["step_frac"], 3),
      "| postnet", round(r["postnet"]["ms"], 3), round(r["postnet"]["frac"], 3), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
