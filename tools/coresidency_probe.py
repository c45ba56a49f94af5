"""Do the launches that are MFMA-bound by shape keep the HBM-bound ones of the other forwards off the chip?  (dev probe)

The 256x256 tile owns its CU (8 waves x 256 registers, 128 KB of LDS); the 128x128 8-wave tile (64 KB, 127 registers) leaves room for
a workgroup of another forward's launch.  The throughput tuner moved the big layers to 256x256 one at a time; this probe moves them
back JOINTLY and measures the four-in-flight step.     python tools/coresidency_probe.py [streams]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import engine, fixture, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
serving.enter(S)
run = tune.StepRunner(m, l, r, info, S)
with torch.no_grad():
    for _ in range(2):
        run.measure(24)
base = dict(engine._TUNED)
big = [k for k, v in base.items() if tuple(v[:4]) == (4, 4, 8, 2) and k[0] == 'f16x3' and k[1] == 2]
print('%d shape keys on the 256x256 tile:' % len(big))
for k in big:
    print('   ', tune._fmt_key(k))


def measure(label, repl, only=None):
    engine._TUNED.clear()
    engine._TUNED.update(base)
    for k in big:
        if only is None or only(k):
            engine._TUNED[k] = tuple(repl) + (1,)
    engine.PLAN_EPOCH += 1
    with torch.no_grad():
        t = run.measure(24)
    print('%-60s %.3f ms/step (%.1f pairs/s)' % (label, t, 1e3 / t), flush=True)


is_kpts = lambda k: k[4] == 14 and k[5] == 14              # (OH, OW) of the keypoint tower
measure('shipped', (4, 4, 8, 2))
measure('all of them on 128x128 / 8 waves / 2 stages', (2, 2, 8, 2))
measure('all of them on 256x128 / 3 stages', (4, 2, 8, 3))
measure('keypoint tower only on 128x128', (2, 2, 8, 2), is_kpts)
measure('all but the keypoint tower on 128x128', (2, 2, 8, 2), lambda k: not is_kpts(k))
measure('all of them on 128x128 / 4 waves', (2, 2, 4, 2))
measure('shipped (again: drift)', (4, 4, 8, 2))
