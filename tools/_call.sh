cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
( time python bench.py --steps 30 ) > gpurun_out/c16/bench.json 2> gpurun_out/c16/bench.err; echo rc=$?
tail -3 gpurun_out/c16/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/c16/bench.json').read().strip().splitlines()[0])
r=d['roofline']; print(d['value'], r['traffic'], r['traffic_over_algorithmic'], r['traffic_note'][:160])
PY
