"""dev tool: the streamed 3-D flow fed (a) preprocessed tensors, (b) uint8 device images (fused preprocessing), (c) uint8 HOST arrays
through the pinned staging pool + H2D as test_net.run_split does -- same pair every frame, no PNG decode, no result files."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, test_net
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False); m.create_architecture(); m.load_state_dict(fixture.make_state_dict(3)); m.cuda().eval()
m.precision = 'f16x3'; m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
calib = bench.demo_calib()
lu_h, ru_h = fixture.synthetic_pair(3, 375, 1242)
lu, ru = torch.from_numpy(lu_h).to(dev), torch.from_numpy(ru_h).to(dev)
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
pipeline.LAZY_KPTS = True
N = 96
pinned = test_net._PinnedPool()
def staged():
    for _ in range(N):
        yield (pinned.stage(lu_h, dev), pinned.stage(ru_h, dev), calib)
pin_l = torch.empty(lu_h.shape, dtype=torch.uint8, pin_memory=True); pin_l.numpy()[...] = lu_h
pin_r = torch.empty(ru_h.shape, dtype=torch.uint8, pin_memory=True); pin_r.numpy()[...] = ru_h
def memcpy_only():
    for _ in range(N):
        pin_l.numpy()[...] = lu_h; pin_r.numpy()[...] = ru_h
        yield (lu, ru, calib)
def h2d_only():
    for _ in range(N):
        yield (pin_l.to(dev, non_blocking=True), pin_r.to(dev, non_blocking=True), calib)
copy_stream = torch.cuda.Stream()
def h2d_side_stream():
    for _ in range(N):
        with torch.cuda.stream(copy_stream):
            a, b = pin_l.to(dev, non_blocking=True), pin_r.to(dev, non_blocking=True)
        torch.cuda.current_stream().wait_stream(copy_stream)
        yield (a, b, calib)
for label, mk in (('pinned host images read by the kernel (zero-copy)', lambda: [(pin_l, pin_r, calib)] * N), ('host memcpy into pinned only (device images reused)', memcpy_only), ('H2D from a fixed pinned buffer (null stream)', h2d_only),
                  ('H2D on a side stream', h2d_side_stream), ('tensors', lambda: [(l, r, info, calib, (375, 1242, 3), float(info[0, 2]))] * N),
                  ('uint8 device images', lambda: [(lu, ru, calib)] * N),
                  ('uint8 host arrays via pinned staging + H2D', staged)):
    list(pipeline.detect_3d_stream(m, [(lu, ru, calib)] * 8 if label != 'tensors' else list(mk())[:8], slots=4, solver='host'))
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        outs = list(pipeline.detect_3d_stream(m, mk(), slots=4, solver='host'))
        torch.cuda.synchronize()
        print('%-46s %.3f ms/pair (%d objects)' % (label, (time.perf_counter() - t) / N * 1e3, len(outs[0])), flush=True)
