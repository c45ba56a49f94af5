"""Experiment (round 5): fewer, fatter workgroups on the small-M trunk layers + more forwards in flight.

The in-mix residency table (profiles/mix_layers_r05.txt) shows the M = 9500 layers (layer3: 69 launches, 27 % of the step) at
36-51 % MFMA-busy while their workgroups sit on a CU, against 65 % for the 256x256-tile launches; a lone 256x256 workgroup runs its
K loop at 84 % (1.46 of 1.74 us per K tile, profiles/row_limit_r03.txt) where a lone 128x128 one runs at 56 %.  With several
forwards in flight the chip is filled by OTHER forwards, so a layer is worth what it costs in CU-time, not in latency: 38 (or 152)
fat workgroups per launch instead of 150-600 thin ones.  This probe swaps the plans of the layer3 shapes and measures the
S-in-flight headline step for several S.
    python tools/fat_tiles_probe.py [--S 4,5,6,7] [--steps 24]
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import engine, fixture, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--S', default='4,6,7')
ap.add_argument('--steps', type=int, default=24)
args = ap.parse_args()
SS = [int(v) for v in args.S.split(',')]
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
serving.enter(max(SS))
runners = {S: tune.StepRunner(m, l, r, info, S) for S in SS}
with torch.no_grad():
    for S in SS:
        runners[S].run(S)
        torch.cuda.synchronize()


def keys(cin, cout, k, hw=38, mode=0):
    return [key for key in engine._TUNED if key[0] == 'f16x3' and key[1] == 2 and key[2] == hw and key[6] == cin and key[7] == cout
            and key[8] == k and key[12] == mode and 'x2' not in key and 'lim' not in key]


L3 = {'conv1': keys(1024, 256, 1), 'conv2': keys(256, 256, 3), 'conv3': keys(256, 1024, 1)}
L4 = {'conv1': keys(2048, 512, 1, 19), 'conv2': keys(512, 512, 3, 19), 'conv3': keys(512, 2048, 1, 19)}
OTH = {'smooth1': keys(256, 256, 3, 38), 'rpnP4': keys(256, 512, 3, 38, 2), 'rpnP3': keys(256, 512, 3, 75, 2), 'smooth2': keys(256, 256, 3, 75),
       'l2conv3': keys(128, 512, 1, 75), 'lat2': keys(512, 256, 1, 75)}
print('layer3 keys:', {k: [engine._TUNED[q] for q in v] for k, v in L3.items()})
print('layer4 keys:', {k: [engine._TUNED[q] for q in v] for k, v in L4.items()})
print('other keys :', {k: [engine._TUNED[q] for q in v] for k, v in OTH.items()})
base_plans = dict(engine._TUNED)
W, H = (4, 4, 8, 2, 1), (4, 2, 8, 3, 1)
SETS = [
    ('shipped', {}),
    ('l3.conv3 256x256', {'conv3': W}),
    ('l3.conv2 256x256', {'conv2': W}),
    ('l3.conv1 256x256', {'conv1': W}),
    ('l3 all 256x256', {'conv1': W, 'conv2': W, 'conv3': W}),
    ('l3 conv1/2 256x128, conv3 256x256', {'conv1': H, 'conv2': H, 'conv3': W}),
    ('l3 all 256x256 + l4 conv2/conv3 256x256 unsplit', {'conv1': W, 'conv2': W, 'conv3': W, 'l4': 1}),
    ('l3 all 256x256 + P4-level smooth / rpn 256x256', {'conv1': W, 'conv2': W, 'conv3': W, 'p4': 1}),
]


def apply(spec):
    engine._TUNED.clear()
    engine._TUNED.update(base_plans)
    for name, plan in spec.items():
        if name in L3:
            for k in L3[name]:
                engine._TUNED[k] = plan
    if spec.get('l4'):
        for name in ('conv2', 'conv3'):
            for k in L4[name]:
                engine._TUNED[k] = W
    if spec.get('p4'):
        for name in ('smooth1', 'rpnP4'):
            for k in OTH[name]:
                engine._TUNED[k] = W
    engine.PLAN_EPOCH += 1


with torch.no_grad():
    for label, spec in SETS:
        apply(spec)
        res = []
        for S in SS:
            serving.enter(S)
            t = runners[S].measure(args.steps)
            res.append('S=%d %.3f ms (%.1f/s)' % (S, t, 1e3 / t))
        print('%-52s %s' % (label, '   '.join(res)), flush=True)
    apply({})
    res = []
    for S in SS:
        serving.enter(S)
        t = runners[S].measure(args.steps)
        res.append('S=%d %.3f ms (%.1f/s)' % (S, t, 1e3 / t))
    print('%-52s %s' % ('shipped (again: drift)', '   '.join(res)), flush=True)
