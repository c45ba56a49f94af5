"""The full 3-D flow (pipeline.detect_3d_stream, solver='host') at several pipeline depths (dev tool).
usage: [SRCNN_SIDE_STREAMS=pool] python tools/flow3d_probe.py --slots 3,4,5 --frames 36"""
import argparse
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, streams, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--slots', default='3,4')
ap.add_argument('--frames', type=int, default=36)
ap.add_argument('--lazy', default='0,1')
args = ap.parse_args()
streams.ensure_hw_queues()
dev = torch.device('cuda:0')
tune.load_shipped_plans()
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
frame = (l, r, info, bench.demo_calib(), (375, 1242, 3), float(info[0, 2]))
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
for lazy in [int(v) for v in args.lazy.split(',')]:
    pipeline.LAZY_KPTS = bool(lazy)
    for S in [int(v) for v in args.slots.split(',')]:
        list(pipeline.detect_3d_stream(m, [frame] * (2 * S), slots=S, solver='host'))
        torch.cuda.synchronize()
        ts = []
        for rep in range(2):
            t0 = time.perf_counter()
            outs = list(pipeline.detect_3d_stream(m, [frame] * args.frames, slots=S, solver='host'))
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / args.frames * 1e3)
        print('side=%s keypoints on kept only=%d slots=%d: %s ms/pair -> %.1f pairs/s, %d objects' % (
            streams.SIDE_KIND, lazy, S, ' '.join('%.3f' % t for t in ts), 1e3 / min(ts), len(outs[0])), flush=True)
