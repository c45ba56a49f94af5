"""Where the host's time goes in the batch form of the full 3-D flow (BASELINE configs[2]: one forward over B pairs, then the 3-D
stage per image), two batches in flight as bench.py --config 2 runs it (dev tool).
    python tools/batch_host_probe.py [--batch 8] [--steps 8] [--profile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, streams
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--slots', type=int, default=2)
ap.add_argument('--profile', action='store_true')
ap.add_argument('--lazy', type=int, default=0)
args = ap.parse_args()
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
B, S = args.batch, args.slots
l, r, info = bench.make_batch(2, 0, 375, 1242, dev) if B == 8 else [torch.cat([q[k] for q in [fixture.make_inputs(3 + b, 375, 1242) for b in range(B)]], 0).to(dev) for k in range(3)]
calib, shape = bench.demo_calib(), (375, 1242, 3)
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
pipeline.LAZY_KPTS = bool(args.lazy)
serving.enter(S)
ss = streams.main_streams(S)
pending = {}


def step(k):
    slot = k % S
    with torch.cuda.stream(ss[slot]):
        old = pending.pop(slot, None)
        if old is not None:
            pipeline.collect_3d_batch(old)
        pending[slot] = pipeline.launch_3d_batch(m, l, r, info, [calib] * B, [shape] * B, slot=slot, solver='host')


def drain():
    for slot in sorted(pending):
        pipeline.collect_3d_batch(pending.pop(slot))


with torch.no_grad():
    for k in range(2 * S):
        step(k)
    drain()
    torch.cuda.synchronize()
    for rep in range(2):
        pipeline.TIMERS = {}
        pr = cProfile.Profile() if args.profile and rep == 1 else None
        t0 = time.perf_counter()
        if pr:
            pr.enable()
        tl = 0.0
        for k in range(args.steps):
            step(k)
        drain()
        if pr:
            pr.disable()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tm, pipeline.TIMERS = pipeline.TIMERS, None
        n = args.steps
        print('B=%d S=%d: %.2f ms per batch (host loop %.2f), %.1f pairs/s; per batch: host solves %.2f ms, waiting for the GPU %.2f ms, other timers %s'
              % (B, S, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3, n * B / (t2 - t0), tm.get('solve_s', 0) / n * 1e3, tm.get('gpu_wait_s', 0) / n * 1e3,
                 {k: round(v / n * 1e3, 2) for k, v in tm.items() if k not in ('solve_s', 'gpu_wait_s')}), flush=True)
        if pr:
            pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
