cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
( time timeout 1700 python -m pytest tests -q -m gpu ) > gpurun_out/c10/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/c10/gpu_tests.log
