"""dev tool: host time to enqueue ONE forward on an idle GPU (no back-pressure), eager Python launch code vs the native
launch program, and the GPU time of the same forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval(); m.precision = 'f16x3'
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
with torch.no_grad():
    for mode in ('eager', 'program', 'graph'):
        m.use_program, m.use_graph = mode == 'program', mode == 'graph'
        for _ in range(3):
            m(l, r, info)
        torch.cuda.synchronize()
        host, total = [], []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m(l, r, info)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e3)
            total.append((t2 - t0) * 1e3)
        host.sort(); total.sort()
        print('%-8s host enqueue of one forward on an idle GPU: median %.2f ms (min %.2f); enqueue + GPU: median %.2f ms'
              % (mode, host[5], host[0], total[5]), flush=True)
