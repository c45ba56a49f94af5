"""Hardware-queue placement probe (dev tool): the multi-stream headline step under the stream kinds of stereo_rcnn_amd/streams.py.
usage: SRCNN_MAIN_STREAMS=dedicated SRCNN_SIDE_STREAMS=none python tools/queue_probe.py --streams 3,4 --steps 120"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, streams, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', default='3')
ap.add_argument('--steps', type=int, default=120)
ap.add_argument('--repeats', type=int, default=3)
ap.add_argument('--no-shipped-plans', action='store_true')
ap.add_argument('--kpts', type=int, default=1)
ap.add_argument('--sync-every', default='0', help='comma list: device synchronisation every N steps inside the timed loop (0 = never): re-aligns the phases of the forwards in flight')
ap.add_argument('--then-single', type=int, default=0, help='afterwards: N forwards one at a time on the null stream (latency mode)')
args = ap.parse_args()
streams.ensure_hw_queues()
dev = torch.device('cuda:0')
if not args.no_shipped_plans:
    tune.load_shipped_plans()
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
for S in [int(v) for v in args.streams.split(',')]:
    run = tune.StepRunner(m, l, r, info, S, kpts=bool(args.kpts))
    with torch.no_grad():
        for _ in range(2):
            run.run(max(S, 1))
            torch.cuda.synchronize()
        for sync_every in [int(v) for v in args.sync_every.split(',')]:
            ts = []
            for _ in range(args.repeats):
                t0 = time.perf_counter()
                if sync_every > 0:
                    for _k in range(0, args.steps, sync_every):
                        run.run(min(sync_every, args.steps - _k))
                        torch.cuda.synchronize()
                else:
                    run.run(args.steps)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / args.steps * 1e3)
            print('main=%-9s side=%-9s queues=%-2s S=%d sync every %d: %s ms/step -> %.1f pairs/s (best)' % (
                streams.MAIN_KIND, streams.SIDE_KIND, os.environ.get('GPU_MAX_HW_QUEUES', '4'), S, sync_every, ' '.join('%.3f' % t for t in ts), 1e3 / min(ts)), flush=True)
if args.then_single:
    streams.set_pairs_in_flight(1)
    run1 = tune.StepRunner(m, l, r, info, 1)
    with torch.no_grad():
        for rep in range(3):
            run1.run(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run1.run(args.then_single)
            torch.cuda.synchronize()
            print('   then one at a time (branches on side streams: %s): %.3f ms/step' % (streams.branch_overlap(), (time.perf_counter() - t0) / args.then_single * 1e3), flush=True)
