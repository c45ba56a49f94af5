#!/bin/bash
# Round 5, second half: the measured artefacts of the FINAL sources on one GPU box into gpurun_out/r05b/ (what is to be judged is
# copied to profiles/).  Reduced form of refresh_profiles_r05.sh (the A/Bs of this half have their own files in profiles/).
#   usage (inside gpurun): bash tools/refresh_profiles_r05b.sh
TAG=r05b
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# the headline line exactly as the driver runs it (PMC traffic live, per-layer table alone on the chip, in-mix per-layer tables)
python bench.py --steps 20 --warmup 5 --layers-out $O/layers_${TAG}_config1.txt --mix-out $O/mix_layers_${TAG}_bench.txt > $O/bench_${TAG}_f16x3.json 2> $O/bench_err.log
python bench.py --config 3 > $O/bench_${TAG}_config3.json 2>> $O/bench_err.log
python bench.py --no-pmc --config 2 --steps 12 --warmup 2 > $O/bench_${TAG}_config2.json 2>> $O/bench_err.log
python bench.py --no-pmc --config 4 --steps 8 --warmup 2 > $O/bench_${TAG}_config4.json 2>> $O/bench_err.log
python tools/time_forward.py f16x3 > $O/stage_times_${TAG}_f16x3.txt 2>&1
python tools/skip_probe.py --out $O/skip_probe_${TAG}.txt > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
rm -f /tmp/plans_${TAG}.json
python $R/bench.py --no-pmc --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-3d-leg --no-mix-layers --streams 1 --plans /tmp/plans_${TAG}.json > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-3d-leg --no-mix-layers --streams 1 --plans /tmp/plans_${TAG}.json > $O/prof_bench.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_f16x3_bench_kernel_stats.csv \;
python $R/tools/stats_avg.py $O/${TAG}_f16x3_bench_kernel_stats.csv > $O/${TAG}_f16x3_bench_conv_avg.txt 2>&1
grep -o '"avg_launch_ms": [0-9.]*' $O/prof_bench.log >> $O/${TAG}_f16x3_bench_conv_avg.txt
python $R/tools/trace_analyze.py $O/prof 12 > $O/timeline_${TAG}_f16x3.txt 2>&1
rm -rf $O/prof
ls -la $O
