cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/c7/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/c7/tests.log
timeout 900 python bench.py --config 4 --steps 6 --warmup 2 --layers-out gpurun_out/c7/layers_cfg4.txt > gpurun_out/c7/bench4.json 2> gpurun_out/c7/bench4.err; echo "bench4 rc=$?"
tail -3 gpurun_out/c7/bench4.err; cut -c1-300 gpurun_out/c7/bench4.json
