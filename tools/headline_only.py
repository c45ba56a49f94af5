"""Only the multi-stream headline loop (stereo_rcnn_amd.tune.StepRunner: S batch-1 forwards in flight, forward + decode + class
NMS), for kernel traces: warm-up, then N steps.   usage: python tools/headline_only.py [--streams 3] [--steps 40] [--no-shipped-plans]"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=3)
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--no-shipped-plans', action='store_true')
args = ap.parse_args()
from stereo_rcnn_amd import streams as _st
_st.ensure_hw_queues()
dev = torch.device('cuda:0')
if not args.no_shipped_plans:
    print('shipped plans loaded:', tune.load_shipped_plans())
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
run = tune.StepRunner(m, l, r, info, args.streams)
with torch.no_grad():
    for _ in range(3):
        run.run(args.streams)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print('%d streams, %d steps: %.3f ms/step = %.1f pairs/s' % (args.streams, args.steps, dt / args.steps * 1e3, args.steps / dt))
