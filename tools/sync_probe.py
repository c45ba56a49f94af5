"""Which host calls of the 3-D flows block on the device (dev tool): torch's sync debug mode ('warn') around (a) the streamed flow
fed tensors, (b) fed page-locked images, (c) the batch form, (d) the KITTI loop -- every warning's innermost frame inside this
repo, counted.  Blocking reads the flow needs (the record's event wait is an event synchronize, not a stream one) do not show;
a `float(device_tensor)` behind an enqueued forward does.
    python tools/sync_probe.py"""
import collections
import os
import sys
import tempfile
import traceback
import warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, test_net
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
calib, shape = bench.demo_calib(), (375, 1242, 3)
frame = (l, r, info, calib, shape, float(info[0, 2]))
lu, ru = fixture.synthetic_pair(3, 375, 1242)
import numpy as np
lp, rp = torch.from_numpy(np.ascontiguousarray(lu)).pin_memory(), torch.from_numpy(np.ascontiguousarray(ru)).pin_memory()
l8, r8, i8 = bench.make_batch(2, 0, 375, 1242, dev)
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
root = tempfile.mkdtemp(prefix='kitti_')
ids = fixture.write_kitti_tree(root, 12, distinct=4)

seen = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' not in str(message):
        return
    inner = [f for f in traceback.extract_stack() if ROOT in f.filename and 'sync_probe' not in f.filename]
    where = '%s:%d %s' % (os.path.relpath(inner[-1].filename, ROOT), inner[-1].lineno, inner[-1].line) if inner else 'outside the repo'
    seen[(leg, where)] += 1


def legs():
    yield 'streamed flow, tensors (host solver)', lambda: list(pipeline.detect_3d_stream(m, [frame] * 12, slots=4, solver='host'))
    yield 'streamed flow, tensors (device solver)', lambda: list(pipeline.detect_3d_stream(m, [frame] * 12, slots=4, solver='device'))
    yield 'streamed flow, page-locked images', lambda: list(pipeline.detect_3d_stream(m, [(lp, rp, calib)] * 12, slots=4, solver='host'))
    yield 'batch form, B = 8', lambda: pipeline.detect_3d_batch(m, l8, r8, i8, [calib] * 8, [shape] * 8)
    yield 'one pair, detect_3d', lambda: pipeline.detect_3d(m, l, r, info, calib, shape)
    if ids is not None:
        yield 'KITTI loop (test_net.run_split)', lambda: test_net.run_split(m, root, ids, os.path.join(root, 'res'), dev, solver='host', slots=4, prefetch=4)


with torch.no_grad():
    for leg, fn in legs():
        for lazy in (False, True):
            pipeline.LAZY_KPTS = lazy
            fn()                                   # first touch: tuning, programs, buffers (synchronising by design)
        torch.cuda.synchronize()
        warnings.showwarning = hook
        warnings.simplefilter('always')
        torch.cuda.set_sync_debug_mode('warn')
        try:
            fn()
        finally:
            torch.cuda.set_sync_debug_mode('default')
        torch.cuda.synchronize()
        n = sum(v for (lg, _), v in seen.items() if lg == leg)
        print('%-44s %d synchronising torch calls' % (leg, n))
        for (lg, where), v in sorted(seen.items()):
            if lg == leg:
                print('    %3d x  %s' % (v, where))
