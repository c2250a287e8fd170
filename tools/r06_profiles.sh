#!/bin/bash
# Round 6's committed evidence: bench line + rocprofv3 kernel stats + HBM / MFMA counters for the headline configuration
# (BASELINE configs[1]: the persistent decode launch) and the two other single-GPU shapes (configs[2]-sized batch 128 fp32; the
# configs[4] shard, batch 64 bf16: the group kernel and the bf16 kernel of the persistent launch).
#   bash tools/r06_profiles.sh <tag> [cfg2|cfg3|cfg5 ...]         e.g. r06 -> gpurun_out/prof_r05a_cfg{2,3,5}/summary/*
TAG=${1:-r06}
shift || true
CFGS=${*:-cfg2 cfg3 cfg5}
cd "${GRAFT_REPO_ROOT:-.}"
for c in $CFGS; do
  case $c in
    cfg2) EXTRA="" ;;
    cfg3) EXTRA="--batch-per-gpu 128" ;;
    cfg5) EXTRA="--batch-per-gpu 64 --mixed" ;;
  esac
  bash tools/profile.sh ${TAG}_$c $EXTRA > gpurun_out/${TAG}_${c}_profile.log 2>&1
  if [ $c = cfg2 ]; then python bench.py > gpurun_out/${TAG}_${c}_bench.json 2> gpurun_out/${TAG}_${c}_bench.err
  else python bench.py $EXTRA --no-serving > gpurun_out/${TAG}_${c}_bench.json 2> gpurun_out/${TAG}_${c}_bench.err; fi
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_${c}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("${c}", round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 3), "M frames/s | bound", r["bound"], "frac", round(r["frac"], 3), "in-run", round(r["frac_in_run"], 3),
      "| step", r["step_bound"], round(r["step_frac"], 3), round(r["decode_step"]["us"], 2), "us | postnet", round(r["postnet"]["ms"], 3), round(r["postnet"]["frac"], 3),
      "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
