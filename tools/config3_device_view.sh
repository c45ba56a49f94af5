# device-side view of BASELINE configs[3] against the same flow fed resident tensors (round 6): kernel traces of both, tools/device_busy.py
R=$(pwd); O=$R/gpurun_out/r06h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c3 -o t -- python $R/bench.py --config 3 --steps 400 --no-pmc --no-cpu-baseline --no-parity --no-sustained > $O/bench_c3.json 2> $O/bench_c3.err
python $R/tools/device_busy.py $O/trace_c3 > $O/device_busy_config3.txt 2>&1
rm -rf $O/trace_c3
rocprofv3 --kernel-trace --output-format csv -d $O/trace_t -o t -- python $R/tools/flow3d_breakdown.py --slots 4 --frames 400 > $O/flow_tensors.txt 2>&1
python $R/tools/device_busy.py $O/trace_t --skip 0.75 --tail 0.02 > $O/device_busy_tensors.txt 2>&1
rm -rf $O/trace_t
cd $R
python -c "
import json; d = json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print('config3 under the tracer:', d['value'], d['config']['host_ms_per_pair'])"
tail -3 $O/flow_tensors.txt
cat $O/device_busy_config3.txt; cat $O/device_busy_tensors.txt
