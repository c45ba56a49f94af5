"""Throughput-tunes the conv plans of the headline workload on this GPU and writes the shipped plan file
(stereo_rcnn_amd/plans/mi355x.json): see stereo_rcnn_amd/tune.py.   usage: python tools/tune_headline.py [--streams 3] [--out name.json]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine, fixture, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=3)
ap.add_argument('--out', default='mi355x.json')
ap.add_argument('--rounds', type=int, default=2)
ap.add_argument('--cands', type=int, default=5)
ap.add_argument('--min-gain', type=float, default=0.004)
ap.add_argument('--steps', type=int, default=24)
ap.add_argument('--start', default='', help='plan file to start from (default: in-situ isolated tuning)')
args = ap.parse_args()
from stereo_rcnn_amd import streams as _st
_st.ensure_hw_queues()
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
if args.start:
    print('starting from', args.start, engine.load_plans(args.start), 'plans')
run = tune.StepRunner(m, l, r, info, args.streams)
with torch.no_grad():
    for s in range(args.streams):          # first touch per slot: in-situ tuning (isolated objective), program recording
        run.run(args.streams)
        torch.cuda.synchronize()
noise = [run.measure(24) for _ in range(5)]
print('noise check, 5 x median-of-3 of 24 steps: ' + ' '.join('%.3f' % t for t in noise) + ' ms/step')
base, final, changes = tune.tune_throughput(m, l, r, info, streams=args.streams, rounds=args.rounds, cands_per_shape=args.cands, min_gain=args.min_gain, steps=args.steps, log=print)
tune.save_shipped(args.out, {'gpu': torch.cuda.get_device_name(0), 'streams': args.streams, 'workload': 'BASELINE configs[1], network input 600x1987, batch 1',
                             'ms_per_step_before': round(base, 3), 'ms_per_step_after': round(final, 3),
                             'changes': [[list(k), list(a), list(b), round(t, 3)] for k, a, b, t in changes]})
print('wrote', tune.shipped_plans_path(args.out))
