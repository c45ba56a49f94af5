import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import distributed as sdist
from stereo_rcnn_amd import fixture, pipeline, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False); m.create_architecture(); m.load_state_dict(fixture.make_state_dict(3)); m.cuda().eval()
m.precision = 'f16x3'; m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
frame = (l, r, info, bench.demo_calib(), (375, 1242, 3), float(info[0, 2]))
pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
pipeline.LAZY_KPTS = True
order = sys.argv[1] if len(sys.argv) > 1 else 'flow,full,flow,lazy,flow'
def flow(tag):
    list(pipeline.detect_3d_stream(m, [frame] * 8, slots=4, solver='host')); torch.cuda.synchronize()
    t = time.perf_counter(); list(pipeline.detect_3d_stream(m, [frame] * 96, slots=4, solver='host')); torch.cuda.synchronize()
    print('%s: flow %.3f ms/pair' % (tag, (time.perf_counter() - t) / 96 * 1e3), flush=True)
serving.enter(4)
for i, what in enumerate(order.split(',')):
    if what == 'flow':
        flow('step %d' % i)
    else:
        run = tune.StepRunner(m, l, r, info, 4, kpts=(what == 'full'))
        with torch.no_grad():
            run.run(8); torch.cuda.synchronize()
        print('step %d: %s headline %.3f ms' % (i, what, run.measure(24, 2)), flush=True)
