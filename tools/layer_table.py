"""Per-layer roofline table of one forward at BASELINE configs[1] (dev / profiling tool).
    python tools/layer_table.py [f16x3|f32] [out.txt]
Every conv launch alone on the chip (one stream, eager), time from the library's HIP events on the launch stream."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, layer_table
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda()
m.eval()
m.precision = prec
m.use_graph = False
m.use_program = False
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
with torch.no_grad():
    m(l, r, info)                               # tunes the plans
    for pl in m._plans.values():
        pl.overlap = False                      # one stream: every launch alone on the chip
    rows = layer_table.measure(lambda: m(l, r, info), reps=5, precision=prec)
txt = layer_table.format_table(rows, 'conv engine %s, 375x1242 pair (network input 600x1987), MI355X; peaks: MFMA 2.5 PF (f16, x3 issued) / '
                               'HBM 6.3 TB/s achievable' % prec)
print(txt)
if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
        f.write(txt + '\n')
