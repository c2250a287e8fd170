cd $GRAFT_REPO_ROOT
bash tools/ab.sh 3 "" "GSTTACO_LIB=$GRAFT_REPO_ROOT/tools/_ab/libgsttaco_tv128.so" "X=1" 2>&1 | tee gpurun_out/r5k_ab.txt
