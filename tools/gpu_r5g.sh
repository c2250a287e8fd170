cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "persistent_decode_launch_is_bitwise or config5 or mixed" 2>&1 | tail -30 > gpurun_out/r5g/tests.log
tail -12 gpurun_out/r5g/tests.log
