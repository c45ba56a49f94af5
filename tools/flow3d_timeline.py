"""dev tool: where the loop thread of pipeline.detect_3d_stream spends a frame (tensor inputs, solver='host', 4 slots): per-phase waits."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False); m.create_architecture(); m.load_state_dict(fixture.make_state_dict(3)); m.cuda().eval()
m.precision = 'f16x3'; m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
frame = (l, r, info, bench.demo_calib(), (375, 1242, 3), float(info[0, 2]))
pipeline.HOST_SOLVER_THREADS = int(os.environ.get('SOLVER_THREADS', sdist.host_solver_threads()))
pipeline.LAZY_KPTS = True
T = collections.defaultdict(float)
C = collections.defaultdict(int)
orig_step = pipeline.step_3d
def step_3d(st, block=True, _worker=False):
    if st.phase == 0:
        return
    ph = st.phase
    ready = st.event.query()
    if not block and not ready:
        C['opportunistic pass, not ready'] += 1
        return
    t0 = time.perf_counter()
    st.event.synchronize()
    t1 = time.perf_counter()
    orig_step(st, block, _worker)
    t2 = time.perf_counter()
    key = ('blocking' if block else 'opportunistic') + ' phase %d' % ph
    T[key + ' wait'] += t1 - t0
    T[key + ' work (solve + launches)'] += t2 - t1
    C[key] += 1
pipeline.step_3d = step_3d
orig_launch, orig_collect, orig_model = pipeline.launch_3d, pipeline.collect_3d, m.forward
def launch_3d(*a, **k):
    t0 = time.perf_counter(); out = orig_launch(*a, **k); T['launch_3d (python: decode, nms, kept kpts, pack, borders, d2h)'] += time.perf_counter() - t0; return out
pipeline.launch_3d = launch_3d
def collect_3d(st):
    t0 = time.perf_counter(); out = orig_collect(st); T['collect_3d total (incl. blocking phases)'] += time.perf_counter() - t0; return out
pipeline.collect_3d = collect_3d
import types
def fwd(*a, **k):
    t0 = time.perf_counter(); out = orig_model(*a, **k); T['forward enqueue (program replay)'] += time.perf_counter() - t0; return out
m.forward = fwd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4
list(pipeline.detect_3d_stream(m, [frame] * 8, slots=slots, solver='host')); torch.cuda.synchronize()
T.clear(); C.clear()
t = time.perf_counter()
list(pipeline.detect_3d_stream(m, [frame] * N, slots=slots, solver='host')); torch.cuda.synchronize()
tot = (time.perf_counter() - t) / N * 1e3
print('slots %d: %.3f ms per pair; per pair on the loop thread:' % (slots, tot))
for k in sorted(T):
    print('  %-75s %.3f ms' % (k, T[k] / N * 1e3))
print('  counts per pair:', {k: round(v / N, 2) for k, v in C.items()})
