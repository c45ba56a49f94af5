#!/bin/bash
# Round 6: the measured artefacts of the FINAL sources on one GPU box into gpurun_out/r06/ (what is to be judged is copied to
# profiles/, indexed by profiles/INDEX.json).   usage (inside gpurun): bash tools/refresh_profiles_r06.sh
TAG=r06
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# the headline line exactly as the driver runs it (PMC traffic + non-conv table live, per-layer table alone on the chip, in-mix
# per-layer tables, sustained leg, parity block)
python bench.py --layers-out $O/layers_${TAG}_config1.txt --mix-out $O/mix_layers_${TAG}.txt > $O/bench_${TAG}_f16x3.json 2> $O/bench_err.log
python bench.py --config 3 > $O/bench_${TAG}_config3.json 2>> $O/bench_err.log
python bench.py --no-pmc --config 2 --steps 12 --warmup 2 > $O/bench_${TAG}_config2.json 2>> $O/bench_err.log
python bench.py --no-pmc --config 4 --steps 8 --warmup 2 > $O/bench_${TAG}_config4.json 2>> $O/bench_err.log
python tools/time_forward.py f16x3 > $O/stage_times_${TAG}_f16x3.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rm -f /tmp/plans_${TAG}.json
python $R/bench.py --no-pmc --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-3d-leg --no-mix-layers --streams 1 --plans /tmp/plans_${TAG}.json > /dev/null 2>&1
# kernel-level evidence for roofline.avg_launch_ms: one pair at a time, plans preloaded (steady-state launches only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-3d-leg --no-mix-layers --streams 1 --plans /tmp/plans_${TAG}.json > $O/prof_bench.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_f16x3_bench_kernel_stats.csv \;
python $R/tools/stats_avg.py $O/${TAG}_f16x3_bench_kernel_stats.csv > $O/${TAG}_f16x3_bench_conv_avg.txt 2>&1
grep -o '"avg_launch_ms": [0-9.]*' $O/prof_bench.log >> $O/${TAG}_f16x3_bench_conv_avg.txt
python $R/tools/trace_analyze.py $O/prof 12 > $O/timeline_${TAG}_f16x3.txt 2>&1
rm -rf $O/prof
# the 3-D stage's kernels (dense alignment, device solvers, borders, packing) inside the full_3d_flow leg
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3d -o p -- python $R/bench.py --no-pmc --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-mix-layers > $O/prof3d_bench.log 2>&1
python - <<PY > $O/${TAG}_3d_stage_kernels.txt
import csv, glob
rows = []
for f in glob.glob("$O/prof3d/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
keep = ('solve4', 'solve3', 'upsample2x', 'cost_kernel', 'argmin', 'sample_kernel', 'left_sample', 'make_enum', 'finish_kernel', 'infer_boundary',
        'class_select_sort', 'pack_detections', 'align_inputs', 'gather_rows', 'decode_kept', 'stem_pack', 'preprocess')
print('kernels of the 3-D stage inside bench.py --no-pmc full_3d_flow leg (rocprofv3 --kernel-trace --stats), round 6: name, calls, avg us, total ms')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if any(k in r['Name'] for k in keep):
        print('%-74s %6d %10.1f %10.2f' % (r['Name'][:74], int(r['Calls']), float(r['TotalDurationNs']) / int(r['Calls']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $O/prof3d
ls -la $O
