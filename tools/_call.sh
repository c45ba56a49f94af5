cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c13; mkdir -p $O
SRCNN_TAP_INNER=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests (tap inner) rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do for v in 0 1; do
SRCNN_TAP_INNER=$v python bench.py --no-cpu-baseline --no-f32-leg --no-3d-leg --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tap inner = $v: %.1f pairs/s (3 in flight), %.1f one at a time, conv %.3f ms' % (d['value'], d['config']['one_pair_at_a_time']['value'], d['roofline']['conv_ms_per_step']))"
done; done
for v in 0 1; do SRCNN_TAP_INNER=$v SWEEP=0 python tools/conv_bench.py f16s 2>/dev/null | grep -v "^ " > $O/conv_bench_tap$v.txt; done
paste -d'\n' $O/conv_bench_tap0.txt $O/conv_bench_tap1.txt | grep "3x3" | cut -c1-110
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    D=$O/p_${v}_$(echo $C | cut -d' ' -f1)
    SRCNN_TAP_INNER=$v rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o p -- python $R/tools/one_conv.py f16s 4 4 8 2 1 rpn > $D.log 2>&1
  done
  echo "tap inner = $v (rpn conv P2, 256x256 tile)"; python $R/tools/pmc_kernel_sum.py $O conv_f16s; rm -rf $O/p_*
done
