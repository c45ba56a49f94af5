cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/c3/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/c3/gpu_tests.log
