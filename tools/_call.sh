#!/bin/bash
mkdir -p gpurun_out/lazy
python tools/limit_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lazy/limit_probe2.log
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_box3d_gpu.py -q -m gpu -k "same_bits or lazy or row_limit or kept or every_plan" 2>&1 | tail -3
python tools/lazy_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lazy/probe2.log
