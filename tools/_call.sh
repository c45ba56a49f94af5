cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c15; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-f32-leg --no-3d-leg --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f pairs/s (3 in flight), %.1f one at a time, conv %.3f ms' % (d['value'], d['config']['one_pair_at_a_time']['value'], d['roofline']['conv_ms_per_step']))"
done
cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    D=$O/p_$(echo $C | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o p -- python $R/tools/one_conv.py f16s 4 4 8 2 1 rpn > $D.log 2>&1
done
python $R/tools/pmc_kernel_sum.py $O conv_f16s; rm -rf $O/p_*
