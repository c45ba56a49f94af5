cd $GRAFT_REPO_ROOT
for i in 1 2; do for tol in 0 0.03 0.08; do
SRCNN_LDS_TIE_SIGN=-1 SRCNN_LDS_TIE_TOL=$tol python bench.py --no-cpu-baseline --no-f32-leg --no-3d-leg --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prefer LARGE LDS, tie tol $tol: %.1f pairs/s (3 in flight), %.1f one at a time, conv %.3f ms' % (d['value'], d['config']['one_pair_at_a_time']['value'], d['roofline']['conv_ms_per_step']))"
done; done
