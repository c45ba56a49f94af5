#!/bin/bash
# Regenerates every measured artefact of a round on the GPU box into gpurun_out/<tag>/ (copy what is to be judged to profiles/).
#   usage (inside gpurun): bash tools/refresh_profiles.sh r01
TAG=${1:-r02}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m oracle.build > /dev/null 2>&1
rm -f /tmp/plans_${TAG}.json
python bench.py --plans /tmp/plans_${TAG}.json > $O/bench_${TAG}_f16x3.json 2> $O/bench_err.log
python bench.py --precision f32 --no-cpu-baseline --no-f32-leg --no-3d-leg > $O/bench_${TAG}_f32.json 2>> $O/bench_err.log
python bench.py --streams 1 --no-cpu-baseline --no-f32-leg --no-3d-leg > $O/bench_${TAG}_f16x3_1inflight.json 2>> $O/bench_err.log
python tools/demo_pipeline.py > $O/full_pipeline_${TAG}.txt 2>&1
python tools/enqueue_probe.py > $O/host_enqueue_${TAG}.txt 2>&1
python tools/block_bench.py > $O/block_bench_${TAG}.txt 2>&1
python tools/time_forward.py f16x3 > $O/stage_times_${TAG}_f16x3.txt 2>&1
SWEEP=1 python tools/conv_bench.py f16s > $O/conv_microbench_${TAG}_f16x3_split16.txt 2>&1
python tools/conv_bench.py f32 > $O/conv_microbench_${TAG}_f32.txt 2>&1
python tools/bench_configs.py > $O/configs_3_5_${TAG}.txt 2>&1
python tools/nms_bench.py > $O/nms_microbench_${TAG}.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# one pair at a time and pre-tuned plans: every conv launch in this trace is a steady-state launch, alone on the chip
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-3d-leg --streams 1 --plans /tmp/plans_${TAG}.json > $O/prof_bench.log 2>&1
cp $O/prof/*kernel_stats.csv $O/${TAG}_f16x3_bench_kernel_stats.csv 2>/dev/null || find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_f16x3_bench_kernel_stats.csv \;
python $R/tools/stats_avg.py $O/${TAG}_f16x3_bench_kernel_stats.csv > $O/${TAG}_f16x3_bench_conv_avg.txt 2>&1
grep -o '"avg_launch_ms": [0-9.]*' $O/prof_bench.log >> $O/${TAG}_f16x3_bench_conv_avg.txt
python $R/tools/trace_analyze.py $O/prof 12 > $O/timeline_${TAG}_f16x3.txt 2>&1
rm -rf $O/prof
# the 3-D stage's kernels (bench's full_3d_flow leg: 'host' then 'device' solver placement, 3 pairs in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3d -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-f32-leg --plans /tmp/plans_${TAG}.json > $O/prof3d_bench.log 2>&1
find $O/prof3d -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_3d_flow_kernel_stats.csv \;
python - <<PY > $O/${TAG}_3d_stage_kernels.txt 2>&1
import csv
rows = list(csv.DictReader(open('$O/${TAG}_3d_flow_kernel_stats.csv')))
print('kernels of the 3-D stage inside bench.py full_3d_flow leg (rocprofv3 --kernel-trace --stats): name, calls, avg us, total ms')
for r in rows:
    n = r['Name']
    if any(k in n for k in ('solve4', 'solve3', 'infer_boundary', 'align_inputs', 'upsample2x', 'sample_kernel', 'cost_kernel', 'argmin', 'make_enum', 'finish_kernel', 'class_', 'pack_')):
        print('%-70s %6s %10.1f %10.2f' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $O/prof3d
mkdir -p $O/pmc
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=$O/pmc/$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-f32-leg --no-3d-leg --streams 1 --plans /tmp/plans_${TAG}.json > $D.log 2>&1
done
python $R/tools/pmc_sum.py $O/pmc 5 $O/pmc_${TAG}_traffic.json > $O/pmc_${TAG}_f16x3_bench_sums.txt 2>&1
rm -rf $O/pmc
# the headline line once more, now that the PMC traffic file of THESE sources exists (bench.py quotes it only on a hash match)
cp $O/pmc_${TAG}_traffic.json $R/profiles/pmc_${TAG}_traffic.json
cd $R
python bench.py --plans /tmp/plans_${TAG}.json > $O/bench_${TAG}_f16x3.json 2>> $O/bench_err.log
ls -la $O
