"""Autotuner audit (dev tool): runs one forward at BASELINE size, prints for every conv shape the plan the in-situ tuner
chose and its candidates' times, then re-times the best few candidates of the heaviest shapes with more repetitions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, engine
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval()
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
engine.PRECISION = m.precision = 'f16x3'
with torch.no_grad():
    m(l, r, info)
torch.cuda.synchronize()
rows = []
for key, log in engine._TUNE_LOG.items():
    log = sorted(log, key=lambda e: e[1])
    rows.append((key, log))
rows.sort(key=lambda kv: -kv[1][0][1])
for key, log in rows:
    prec, B, H, W, OH, OW, cin, cout, kh, kw, s, p = key[:12]
    print('B%d %dx%d cin%d cout%d k%d s%d: ' % (B, H, W, cin, cout, kh, s) +
          '  '.join('%d%d%d%d/s%d:%.1fus' % (pl + (t * 1e3,)) for pl, t in log[:6]))
