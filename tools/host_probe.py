import os, time, sys
t0 = time.time()
import torch
print('import torch %.1fs' % (time.time() - t0), 'cpus', os.cpu_count(), 'torch threads', torch.get_num_threads(), flush=True)
try:
    print('affinity', len(os.sched_getaffinity(0)))
except Exception as e:
    print(e)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_rcnn_amd import fixture
t = time.time(); sd = fixture.make_state_dict(3); print('make_state_dict %.1fs' % (time.time() - t), flush=True)
t = time.time(); l, r, info = fixture.make_inputs(3, 120, 400, target_short=192); print('make_inputs small %.1fs' % (time.time() - t), flush=True)
from oracle import net
for nt in (8, 16, os.cpu_count()):
    torch.set_num_threads(nt)
    t = time.time(); net.forward(sd, l, r, info); print('oracle small threads=%d %.1fs' % (nt, time.time() - t), flush=True)
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
t = time.time(); m = resnet(('__background__', 'Car'), 101); m.create_architecture(); print('build modules %.1fs' % (time.time() - t), flush=True)
t = time.time(); m.load_state_dict(sd); m.cuda(); m.eval(); print('load+cuda %.1fs' % (time.time() - t), flush=True)
t = time.time(); plan = m._get_plan(1, 192, 640); torch.cuda.synchronize(); print('weights+plan %.1fs' % (time.time() - t), flush=True)
