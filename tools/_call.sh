cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/c11/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/c11/tests.log
for i in 1 2 3; do
for v in 0 1; do
SRCNN_FUSE_UPSAMPLE=$v timeout 400 python bench.py --no-cpu-baseline --no-f32-leg --no-3d-leg > gpurun_out/c11/bench_${v}_$i.json 2> gpurun_out/c11/bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/c11/bench_${v}_$i.json'))
print('fuse=$v: %.1f pairs/s (3 in flight)  one at a time %.1f  conv_ms %.3f launches %d' % (d['value'], d['config']['one_pair_at_a_time']['value'], d['roofline']['conv_ms_per_step'], d['roofline']['launches_per_step']))
PY
done
done
