cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/r05_profiles.sh r05a cfg2 cfg3 cfg5 2>&1 | tail -20
mkdir -p gpurun_out/r05a_summ
for c in cfg2 cfg3 cfg5; do cp gpurun_out/prof_r05a_$c/summary/* gpurun_out/r05a_summ/ 2>/dev/null; cp gpurun_out/r05a_${c}_bench.json gpurun_out/r05a_summ/r05a_${c}_bench.json; done
timeout 200 python tools/stamps_persist.py 32 128 2>&1 | grep -v amdgpu > gpurun_out/r05a_summ/r05a_stamps_persist.txt
timeout 200 python tools/stamps_group.py 128 2>&1 | grep -v amdgpu > gpurun_out/r05a_summ/r05a_stamps_group_b128.txt
timeout 200 python tools/stamps_group.py 64 --mixed 2>&1 | grep -v amdgpu > gpurun_out/r05a_summ/r05a_stamps_bf16_b64.txt
ls gpurun_out/r05a_summ
