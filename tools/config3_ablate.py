"""BASELINE configs[3]'s loop (test_net.run_split) with parts of its host side removed, to find what the flow loses against the same
flow fed resident tensors (dev tool):
    A  as shipped (PNG files decoded on threads, result files + records written)
    B  read_image served from memory (no PNG decode; the page-locked ring copy stays)
    C  B + no result files / records
    T  A with the PNG decode on THREADS of the loop's process (round 5's form; A decodes in worker processes since round 6)
    python tools/config3_ablate.py [--frames 600] [--threads 16]"""
import argparse
import os
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, test_net
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=600)
ap.add_argument('--threads', type=int, default=16)
ap.add_argument('--legs', default='S,R,Q,P,A,S,R,Q,P,A')
args = ap.parse_args()
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
pipeline.LAZY_KPTS = True
base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
root = tempfile.mkdtemp(prefix='srcnn_abl_', dir=base)
ids = fixture.write_kitti_tree(root, args.frames, 16)
res = os.path.join(root, 'res')
cache = {}


def cached_read(path):
    real = os.path.realpath(path)
    if real not in cache:
        cache[real] = test_net.read_png_rgb(path)
    return cache[real]


test_net.run_split(m, root, ids[:16], res, dev, solver='host', slots=4, prefetch=args.threads)
for p in ids[:16]:
    cached_read(os.path.join(root, 'image_2', p + '.png')); cached_read(os.path.join(root, 'image_3', p + '.png'))
write = pipeline.write_kitti_results
# the detector alone on the same data: page-locked images / preprocessed device tensors, the 16 distinct pairs in turn or pair 0 only
from stereo_rcnn_amd import engine
from stereo_rcnn_amd.model.utils import kitti_utils
calib = kitti_utils.read_obj_calibration(os.path.join(root, 'calib', ids[0] + '.txt'))
pins, tens = [], []
for p in ids[:16]:
    lu, ru = cached_read(os.path.join(root, 'image_2', p + '.png')), cached_read(os.path.join(root, 'image_3', p + '.png'))
    pins.append((torch.from_numpy(lu.copy()).pin_memory(), torch.from_numpy(ru.copy()).pin_memory(), calib))
    l, sc = engine.preprocess(torch.from_numpy(lu).to(dev), 600)
    r, _ = engine.preprocess(torch.from_numpy(ru).to(dev), 600)
    info = torch.tensor([[l.shape[2], l.shape[3], sc]], dtype=torch.float32).to(dev)
    tens.append((l, r, info, calib, lu.shape, float(sc)))
for leg in [v for v in args.legs.split(',') if v in 'PQRS']:
    src = {'P': pins, 'Q': pins[:1], 'R': tens, 'S': tens[:1]}[leg]
    frames = [src[i % len(src)] for i in range(args.frames)]
    list(pipeline.detect_3d_stream(m, frames[:16], slots=4, solver='host'))
    torch.cuda.synchronize()
    pipeline.TIMERS = {}
    t0 = time.perf_counter()
    outs = list(pipeline.detect_3d_stream(m, frames, slots=4, solver='host'))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pt, pipeline.TIMERS = pipeline.TIMERS, None
    print('%s  %.1f pairs/s  %.3f ms per pair   (%s, %s)  objects per pair %.1f  solves %.2f  waiting for the GPU %.2f ms per pair'
          % (leg, args.frames / dt, dt / args.frames * 1e3, 'page-locked images' if leg in 'PQ' else 'device tensors',
             '16 distinct pairs' if leg in 'PR' else 'one pair', sum(len(o) for o in outs) / len(outs), pt.get('solve_s', 0) / args.frames * 1e3,
             pt.get('gpu_wait_s', 0) / args.frames * 1e3), flush=True)
for leg in [v for v in args.legs.split(',') if v in 'ABCT']:
    kw = dict(solver='host', slots=4, prefetch=args.threads, records=[])
    pipeline.write_kitti_results = write
    if leg in ('B', 'C'):
        kw['read_image'] = cached_read
    if leg == 'C':
        pipeline.write_kitti_results = lambda *a, **k: None
        kw['records'] = None
    test_net.DECODE_PROCESSES = leg != 'T'
    timers, pipeline.TIMERS = {}, {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n, nobj, _ = test_net.run_split(m, root, ids, res, dev, timers=timers, **kw)
    dt = time.perf_counter() - t0
    pt, pipeline.TIMERS = pipeline.TIMERS, None
    print('%s  %.1f pairs/s  %.3f ms per pair   decode %.2f  files %.2f  solves %.2f  waiting for the GPU %.2f ms per pair'
          % (leg, n / dt, dt / n * 1e3, timers['decode_s'] / n * 1e3, timers['write_s'] / n * 1e3, pt.get('solve_s', 0) / n * 1e3,
             pt.get('gpu_wait_s', 0) / n * 1e3), flush=True)
