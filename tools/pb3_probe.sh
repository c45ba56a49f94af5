cd /root/repo
python bench.py --steps 60 --no-cpu-baseline --no-f32-leg --no-3d-leg 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A (PB_MAX_NS=2):', r['value'], 'pairs/s; one at a time', r['config']['one_pair_at_a_time']['value'], 'kernel-level TF', r['roofline']['achieved'])"
sed -i 's/#define SRCNN_PB_MAX_NS 2 /#define SRCNN_PB_MAX_NS 3 /' stereo_rcnn_amd/csrc/conv_f16s.hip
grep -n "define SRCNN_PB_MAX_NS" stereo_rcnn_amd/csrc/conv_f16s.hip
python -m stereo_rcnn_amd.csrc.build > /tmp/build.log 2>&1; tail -1 /tmp/build.log
python -m pytest tests/test_ops_gpu.py -x -q -k "every_plan" 2>&1 | tail -1
python bench.py --steps 60 --no-cpu-baseline --no-f32-leg --no-3d-leg 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B (PB_MAX_NS=3):', r['value'], 'pairs/s; one at a time', r['config']['one_pair_at_a_time']['value'], 'kernel-level TF', r['roofline']['achieved'])"
python tools/conv_bench.py f16s 2>/dev/null | grep -A0 "fpn.smooth3\|rpn.conv\|kpts 3x3\|l2.conv2"
