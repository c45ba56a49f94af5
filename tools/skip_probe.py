"""What do the NON-conv stages of the forward cost inside the several-forwards-in-flight mix?  (dev tool)

mix_table.marginal prices every conv group by issuing it more often; the remainder of the step (0.73 ms of 6.5 in
profiles/mix_layers_r05.txt) is "the non-conv launches and whatever is not additive".  This tool leaves stages OUT instead
(engine.DEBUG_SKIP: their outputs keep the previous frame's values, so everything downstream still runs on valid data) and
measures the headline step again: base - t_skip is what the stage costs the mix.

    python tools/skip_probe.py [--streams 4] [--steps 40] [--out file]
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from stereo_rcnn_amd import _lib, engine, fixture, serving, tune
from stereo_rcnn_amd import postprocess as hpost
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=4)
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--out', default=None)
ap.add_argument('--cases', default=None, help='semicolon-separated cases, each a +-joined list of stage names (default: all stages, one by one)')
args = ap.parse_args()
serving.before_hip()
dev = torch.device('cuda:0')
print('shipped plans loaded:', tune.load_shipped_plans())
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]


class Runner(tune.StepRunner):
    post = True

    def step(self, slot):
        im_l, im_r, im_info = self.inputs
        out = self.model(im_l, im_r, im_info, slot=slot, kpts=self.kpts, alias_outputs=True)
        if NULLS[0]:
            _L.srcnn_debug_null_launches(NULLS[0], _lib.stream())
        if self.post:
            det = hpost.decode_detections(out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], im_info[0:1])
            hpost.class_nms_device(det, 1, 0.05)


run = Runner(m, l, r, info, args.streams)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


_L = _lib.lib()
_L.srcnn_debug_skip_mask.argtypes = [ctypes.c_int]
_L.srcnn_debug_skip_mask.restype = None
_L.srcnn_debug_null_launches.argtypes = [ctypes.c_int, ctypes.c_void_p]
_L.srcnn_debug_null_launches.restype = ctypes.c_int
NULLS = [0]          # empty kernels launched behind every forward (case 'null:<n>')
PROPOSAL_BITS = {'prop:memset': 128, 'prop:hist': 1, 'prop:decode': 2, 'prop:pair_mask': 4, 'prop:scan': 8, 'prop:intersect': 16, 'prop:compact': 32, 'prop:rank': 64}


def measure(skip=(), post=True):
    mask = sum(PROPOSAL_BITS.get(x, 0) for x in skip)
    NULLS[0] = sum(int(x.split(':')[1]) for x in skip if x.startswith('null:'))
    skip = [x for x in skip if x not in PROPOSAL_BITS and not x.startswith('null:')]
    _L.srcnn_debug_skip_mask(mask)
    try:
        return _measure(skip, post)
    finally:
        _L.srcnn_debug_skip_mask(0)
        NULLS[0] = 0


def _measure(skip=(), post=True):
    engine.DEBUG_SKIP = frozenset(skip)
    engine.PLAN_EPOCH += 1
    run.post = post
    t = run.measure(args.steps)
    engine.DEBUG_SKIP = frozenset()
    engine.PLAN_EPOCH += 1
    run.post = True
    return t


with torch.no_grad():
    for _ in range(2):
        run.measure(args.steps)
    base = [run.measure(args.steps)]
    say('%d forwards in flight, headline step %.3f ms' % (args.streams, base[0]))
    cases = [('maxpool',), ('subsample',), ('rpn_scores',), ('proposals',), ('prop:hist',), ('prop:compact',), ('prop:rank',), ('prop:decode',),
             ('prop:pair_mask',), ('prop:scan',), ('prop:intersect',), ('roi_align7',), ('roi_align14',), ('box_tail', 'kpts_tail'),
             ('decode+class_nms',),
             ('maxpool', 'subsample', 'rpn_scores', 'proposals', 'roi_align', 'box_tail', 'kpts_tail', 'decode+class_nms')]
    if args.cases:
        cases = [tuple(c.split('+')) for c in args.cases.split(';')]
    tot = 0.0
    for c in cases:
        post = 'decode+class_nms' not in c
        t = measure([x for x in c if x != 'decode+class_nms'], post)
        b = run.measure(args.steps)
        ref = 0.5 * (base[-1] + b)
        base.append(b)
        say('  without %-40s %.3f ms vs %.3f -> costs %.1f us per forward in the mix' % (' + '.join(c) if len(c) < 4 else 'ALL of the above', t, ref, (ref - t) * 1e3))
        if len(c) < 4 and not c[0].startswith('prop:'):
            tot += (ref - t) * 1e3
    say('  sum of the single stages: %.1f us' % tot)
if args.out:
    open(args.out, 'w').write('\n'.join(lines) + '\n')
