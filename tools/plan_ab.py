"""Same-box A/B of plan-file changes in BOTH serving regimes (round 5): the headline step (keypoint branch for all 300 rois inside
the forward) and the pipeline's default form of it (branch on the kept detections only).  A plan that wins the first can lose the
second: the mixes differ.     python tools/plan_ab.py <base plans json> <candidate plans json>
Prints both steps for: base, candidate, and base + each single key the candidate changes."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import serving
serving.before_hip()
serving.USE_SHIPPED_PLANS = False
from stereo_rcnn_amd import engine, fixture, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

base_f, cand_f = sys.argv[1], sys.argv[2]
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
load = lambda f: {tuple(k): tuple(v) for k, v in json.load(open(f))}
base, cand = load(base_f), load(cand_f)
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
serving.enter(S, plans=False)
engine._TUNED.update(base)
full = tune.StepRunner(m, l, r, info, S, kpts=True)
kept = tune.StepRunner(m, l, r, info, S, kpts=False)
kept.streams = full.streams
with torch.no_grad():
    for rn in (full, kept):
        rn.run(S)
        torch.cuda.synchronize()
insitu = {k: v for k, v in engine._TUNED.items() if k not in base and k not in cand}


def measure(plans, label):
    engine._TUNED.clear()
    engine._TUNED.update(insitu)
    engine._TUNED.update(plans)
    engine.PLAN_EPOCH += 1
    with torch.no_grad():
        a = full.measure(24)
        b = kept.measure(24)
    print('%-70s all 300 rois %.3f ms (%.1f/s)   kept only %.3f ms (%.1f/s)' % (label, a, 1e3 / a, b, 1e3 / b), flush=True)


measure(base, 'base ' + os.path.basename(base_f))
both = dict(base)
both.update(cand)
measure(both, 'candidate ' + os.path.basename(cand_f))
for k, v in cand.items():
    if base.get(k) != v:
        one = dict(base)
        one[k] = v
        measure(one, 'base + %s: %s -> %s' % (tune._fmt_key(k), base.get(k), v))
measure(base, 'base again (drift)')
