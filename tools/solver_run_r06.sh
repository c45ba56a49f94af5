set -x
R=$(pwd); O=$R/gpurun_out/r06f; mkdir -p $O
python -m pytest tests/test_box3d_gpu.py -m gpu -q -s 2>&1 | grep -v Warning | grep "lanes\|device vs host\|passed\|failed\|Error\|assert" | head -20
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3d -o p -- python $R/bench.py --no-pmc --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-mix-layers > $O/prof3d_bench.log 2>&1
python - <<PY > $O/3d_stage_kernels.txt
import csv, glob
rows = []
for f in glob.glob("$O/prof3d/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if any(k in r['Name'] for k in ('solve4', 'solve3')):
        print('%-74s %6d %10.1f %10.2f' % (r['Name'][:74], int(r['Calls']), float(r['TotalDurationNs']) / int(r['Calls']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $O/prof3d
cat $O/3d_stage_kernels.txt
cd $R
python bench.py --no-pmc --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-mix-layers > $O/bench_3d.json 2>$O/bench_3d.err
python -c "
import json; d = json.load(open('$O/bench_3d.json')); print(d['value']); f = d.get('full_3d_flow') or d['config'].get('full_3d_flow'); print({k: v['value'] for k, v in f.items() if isinstance(v, dict)})"
