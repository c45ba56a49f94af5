cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
timeout 900 python -m pytest tests/test_demo_pair.py tests/test_box3d_gpu.py -x -q -m gpu -s -k "full_flow or well_conditioned or batched" > gpurun_out/c9/t.log 2>&1; echo "rc=$?"
grep -B2 -A12 "Error\|well-conditioned fixture" gpurun_out/c9/t.log | head -60; tail -3 gpurun_out/c9/t.log
