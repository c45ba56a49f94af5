"""Times the full forward (eager and hipGraph replay) at BASELINE size (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, _lib
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval()
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
m.precision = sys.argv[1] if len(sys.argv) > 1 else 'f32'
print('precision', m.precision)
from stereo_rcnn_amd import engine
engine.PRECISION = m.precision
for mode in (False, True):
    m.use_graph = mode
    for _ in range(3):
        m(l, r, info)
    torch.cuda.synchronize()
    n = 10
    t = time.time()
    for _ in range(n):
        m(l, r, info)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print('graph=%s  %.2f ms/pair  %.1f pairs/s  (%.1f TFLOP/s algorithmic)' % (mode, dt * 1e3, 1 / dt, 1.9546 / dt), flush=True)
# per-stage timing (eager)
m.use_graph = False
plan = m._get_plan(1, l.shape[2], l.shape[3])
plan.fmt = 1 if m.precision == 'f16x3' else 0
for name in ('trunk', 'fpn_rpn', 'proposals', 'heads'):
    fn = getattr(plan, name)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print('%-10s %.3f ms' % (name, e0.elapsed_time(e1) / 5), flush=True)
for ov in (False, True):
    plan.overlap = ov
    for g in (False,):
        for _ in range(3): plan.run(g, m.precision)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): plan.run(g, m.precision)
        torch.cuda.synchronize(); print('overlap=%s: %.2f ms/forward' % (ov, (time.time() - t) * 100), flush=True)
L = _lib.lib()
L.srcnn_prof_enable(1)
plan.launch_all(); torch.cuda.synchronize()
import ctypes
ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
L.srcnn_prof_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt))
L.srcnn_prof_enable(0)
print('conv engine: %d launches, %.3f ms, %.1f executed TFLOP/s' % (cnt.value, ms.value, fl.value / ms.value / 1e9))
