cd /root/repo
for q in 1 2 3 4; do for s in 2 3; do
echo "GPU_MAX_HW_QUEUES=$q streams=$s"; GPU_MAX_HW_QUEUES=$q python bench.py --steps 60 --no-cpu-baseline --no-f32-leg --no-3d-leg --streams $s 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', r['value'], 'pairs/s', r['ms_per_step'], 'ms; one at a time', r['config']['one_pair_at_a_time'] and r['config']['one_pair_at_a_time']['value'])"
done; done
