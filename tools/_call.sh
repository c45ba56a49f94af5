#!/bin/bash
mkdir -p gpurun_out/lazy
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_box3d_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -3
python tools/lazy_probe.py 2>&1 | grep -v amdgpu.ids | grep "3 in flight" | tee gpurun_out/lazy/probe3.log
