"""Where the streamed 3-D flow's time per pair goes (dev tool): the same pair through growing parts of the flow, S pairs in flight.
    python tools/flow3d_breakdown.py [--slots 4,6] [--frames 96]"""
import argparse
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--slots', default='4,6')
ap.add_argument('--frames', type=int, default=96)
args = ap.parse_args()
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
frame = (l, r, info, bench.demo_calib(), (375, 1242, 3), float(info[0, 2]))
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
pipeline.LAZY_KPTS = True
for S in [int(v) for v in args.slots.split(',')]:
    serving.enter(S)
    run = tune.StepRunner(m, l, r, info, S, kpts=False)
    with torch.no_grad():
        run.run(2 * S)
        torch.cuda.synchronize()
    print('S=%d  step only (forward without keypoints + decode + class NMS + keypoints on kept): %.3f ms' % (S, run.measure(48, 2)), flush=True)
    for label, kw in (('device solver, no dense alignment', dict(solver='device', dense_align=False)),
                      ('host solver, no dense alignment', dict(solver='host', dense_align=False)),
                      ('device solver, full', dict(solver='device', dense_align=True)),
                      ('host solver, full', dict(solver='host', dense_align=True))):
        list(pipeline.detect_3d_stream(m, [frame] * (2 * S), slots=S, **kw))
        torch.cuda.synchronize()
        ts = []
        for rep in range(2):
            pipeline.TIMERS = {}
            t0 = time.perf_counter()
            outs = list(pipeline.detect_3d_stream(m, [frame] * args.frames, slots=S, **kw))
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / args.frames * 1e3)
            tm, pipeline.TIMERS = pipeline.TIMERS, None
        print('S=%d  %-36s %s ms/pair  (last run: host solves %.2f ms, waiting for the GPU %.2f ms per pair; %d objects)'
              % (S, label, ' '.join('%.3f' % t for t in ts), tm.get('solve_s', 0) / args.frames * 1e3, tm.get('gpu_wait_s', 0) / args.frames * 1e3, len(outs[0])), flush=True)
