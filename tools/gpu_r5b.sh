cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 300 python tools/stamps_group.py 128 > gpurun_out/r5b/stamps128.txt 2>&1
timeout 300 python tools/stamps_group.py 64 > gpurun_out/r5b/stamps64.txt 2>&1
cat gpurun_out/r5b/stamps128.txt gpurun_out/r5b/stamps64.txt
