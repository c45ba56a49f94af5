cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c17
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "calibration or activation_scales or program" > gpurun_out/c17/t.log 2>&1; echo rc=$?
tail -12 gpurun_out/c17/t.log
