cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -s -k "dynamic_range or activation_scales or range_guard" > gpurun_out/split16_dynamic_range_r03.txt 2>&1; echo rc=$?
grep -v amdgpu gpurun_out/split16_dynamic_range_r03.txt | tail -20
