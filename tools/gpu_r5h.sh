cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 300 python tools/stamps_group.py 64 --mixed > gpurun_out/r5h/stamps64m.txt 2>&1
cat gpurun_out/r5h/stamps64m.txt
