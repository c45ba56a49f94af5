#!/bin/bash
# Regenerates every measured artefact of a round on the GPU box into gpurun_out/<tag>/ (copy what is to be judged to profiles/).
#   usage (inside gpurun): bash tools/refresh_profiles.sh r01
TAG=${1:-r04}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m oracle.build > /dev/null 2>&1
rm -f /tmp/plans_${TAG}.json
python bench.py --no-pmc --plans /tmp/plans_${TAG}.json --layers-out $O/layers_${TAG}_config1.txt > $O/bench_${TAG}_f16x3.json 2> $O/bench_err.log
# the other BASELINE configs through the same contract (configs[2]: batch 8 + the whole 3-D flow; configs[4]: ResNet-50, 2x resolution, batch 4)
python bench.py --no-pmc --config 2 --steps 12 --warmup 2 --layers-out $O/layers_${TAG}_config2.txt > $O/bench_${TAG}_config2.json 2>> $O/bench_err.log
python bench.py --no-pmc --config 4 --steps 8 --warmup 2 --layers-out $O/layers_${TAG}_config4.txt > $O/bench_${TAG}_config4.json 2>> $O/bench_err.log
python bench.py --no-pmc --precision f32 --no-cpu-baseline --no-f32-leg --no-3d-leg > $O/bench_${TAG}_f32.json 2>> $O/bench_err.log
python bench.py --no-pmc --streams 1 --no-cpu-baseline --no-f32-leg --no-3d-leg > $O/bench_${TAG}_f16x3_1inflight.json 2>> $O/bench_err.log
python tools/demo_pipeline.py > $O/full_pipeline_${TAG}.txt 2>&1
python tools/enqueue_probe.py > $O/host_enqueue_${TAG}.txt 2>&1
python tools/time_forward.py f16x3 > $O/stage_times_${TAG}_f16x3.txt 2>&1
SWEEP=1 python tools/conv_bench.py --no-pmc f16s > $O/conv_microbench_${TAG}_f16x3_split16.txt 2>&1
python tools/conv_bench.py --no-pmc f32 > $O/conv_microbench_${TAG}_f32.txt 2>&1
python tools/gemm_ceiling.py > $O/gemm_ceiling_${TAG}.txt 2>&1
# what the tuner's LDS cap does to the multi-stream headline (co-residency experiment)
( for cap in 160 128 96 64; do SRCNN_MAX_LDS_KB=$cap python bench.py --no-pmc --no-cpu-baseline --no-f32-leg --no-3d-leg --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tuner LDS cap $cap KB per workgroup: %.1f pairs/s with the default pairs in flight, %.1f one at a time, conv %.3f ms/step' % (d['value'], d['config']['one_pair_at_a_time']['value'], d['roofline']['conv_ms_per_step']))"; done ) > $O/lds_cap_${TAG}.txt 2>&1
# same-box A/B of the one-launch stereo RPN conv (conv mode 2)
( for i in 1 2 3; do for v in 0 1; do SRCNN_RPN_PAIR=$v python bench.py --no-pmc --no-cpu-baseline --no-f32-leg --no-3d-leg --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stereo RPN conv as one launch = $v: %.1f pairs/s (default in flight), %d conv launches per step' % (d['value'], d['roofline']['launches_per_step']))"; done; done ) > $O/rpn_pair_launch_${TAG}.txt 2>&1
# keypoint head on the kept detections (pipeline.LAZY_KPTS): same-box A/B of the step, duration of the tower's launches against
# the device-side row limit, and the M-fast tile order of the fully connected shapes
python tools/lazy_probe.py 2>&1 | grep -v amdgpu.ids > $O/lazy_keypoint_head_${TAG}.txt
python tools/limit_probe.py 2>&1 | grep -v amdgpu.ids > $O/row_limit_${TAG}.txt
( for mf in 0 1; do echo "SRCNN_M_FAST=$mf"; SRCNN_M_FAST=$mf ONLY=box. SWEEP=1 python tools/conv_bench.py f16s 2>&1 | grep -v amdgpu.ids; done ) > $O/m_fast_${TAG}.txt
# the 3-D-box metric's yardstick and the SPLIT16 range / scale tests, with their printed numbers
python -m pytest tests/test_box3d_conditioning.py tests/test_box3d_gpu.py tests/test_demo_pair.py -q -s -k "spread or well_conditioned or full_flow" > $O/box3d_conditioning_${TAG}.txt 2>&1
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -s -k "dynamic_range or activation_scales or range_guard" > $O/split16_dynamic_range_${TAG}.txt 2>&1
python -m pytest tests/test_pipeline_gpu.py -q -s -k soak > $O/three_in_flight_soak_${TAG}.txt 2>&1
python tools/nms_bench.py --no-pmc > $O/nms_microbench_${TAG}.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# one pair at a time and pre-tuned plans: every conv launch in this trace is a steady-state launch, alone on the chip
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-3d-leg --streams 1 --plans /tmp/plans_${TAG}.json > $O/prof_bench.log 2>&1
cp $O/prof/*kernel_stats.csv $O/${TAG}_f16x3_bench_kernel_stats.csv 2>/dev/null || find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_f16x3_bench_kernel_stats.csv \;
python $R/tools/stats_avg.py $O/${TAG}_f16x3_bench_kernel_stats.csv > $O/${TAG}_f16x3_bench_conv_avg.txt 2>&1
grep -o '"avg_launch_ms": [0-9.]*' $O/prof_bench.log >> $O/${TAG}_f16x3_bench_conv_avg.txt
python $R/tools/trace_analyze.py $O/prof 12 > $O/timeline_${TAG}_f16x3.txt 2>&1
rm -rf $O/prof
# the 3-D stage's kernels (bench's full_3d_flow leg: 'host' then 'device' solver placement, 3 pairs in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3d -o p -- python $R/bench.py --no-pmc --steps 12 --warmup 3 --no-cpu-baseline --no-f32-leg --plans /tmp/plans_${TAG}.json > $O/prof3d_bench.log 2>&1
find $O/prof3d -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_3d_flow_kernel_stats.csv \;
python - <<PY > $O/${TAG}_3d_stage_kernels.txt 2>&1
import csv
rows = list(csv.DictReader(open('$O/${TAG}_3d_flow_kernel_stats.csv')))
print('kernels of the 3-D stage inside bench.py --no-pmc full_3d_flow leg (rocprofv3 --kernel-trace --stats): name, calls, avg us, total ms')
for r in rows:
    n = r['Name']
    if any(k in n for k in ('solve4', 'solve3', 'infer_boundary', 'align_inputs', 'upsample2x', 'sample_kernel', 'cost_kernel', 'argmin', 'make_enum', 'finish_kernel', 'class_', 'pack_')):
        print('%-70s %6s %10.1f %10.2f' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $O/prof3d
mkdir -p $O/pmc
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=$O/pmc/$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o p -- python $R/bench.py --no-pmc --steps 4 --warmup 1 --no-cpu-baseline --no-f32-leg --no-3d-leg --streams 1 --plans /tmp/plans_${TAG}.json > $D.log 2>&1
done
python $R/tools/pmc_sum.py $O/pmc 5 $O/pmc_${TAG}_traffic.json > $O/pmc_${TAG}_f16x3_bench_sums.txt 2>&1
rm -rf $O/pmc
# the headline line once more, with roofline.traffic measured live by bench.py itself (child runs under rocprofv3 --pmc); the PMC
# traffic file of THESE sources written above is the fallback it would quote on a hash match
cp $O/pmc_${TAG}_traffic.json $R/profiles/pmc_${TAG}_traffic.json
cd $R
python bench.py --plans /tmp/plans_${TAG}.json > $O/bench_${TAG}_f16x3.json 2>> $O/bench_err.log
ls -la $O
# PMC of the dominant kernel: the 256x256 / 8-wave tile on the RPN conv at P2 (both eyes in one launch), separate --pmc passes
bash $R/tools/pmc_passes.sh $O/pmc_dom -- python $R/tools/one_conv.py f16s 4 4 8 2 1 rpn
( echo "PMC of conv_f16s_kernel<2,4,true,4,2> (256x256 tile, 8 waves of 64x128, 2-stage ring, in-place B fragments) on the RPN conv at P2";
  echo "(M = 149100, N = 512, K = 2304: 352 GFLOP algorithmic per launch), per-launch averages of 10 launches, separate rocprofv3 --pmc passes:";
  python $R/tools/pmc_kernel_sum.py $O/pmc_dom conv_f16s ) > $O/pmc_${TAG}_dominant_kernel_256x256.txt 2>&1
rm -rf $O/pmc_dom
ls -la $O
