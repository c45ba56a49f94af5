"""Runs ONE conv shape repeatedly with a forced plan (for rocprofv3 --pmc passes; dev tool).
usage: one_conv.py <precision: f32|f16x3|f16s> <mr> <nr> <waves> <stages> <splits> [shape-name]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine
prec, plan = sys.argv[1], tuple(int(v) for v in sys.argv[2:7])
name = sys.argv[7] if len(sys.argv) > 7 else 'rpn'
split = prec == 'f16s'
if split:
    prec = 'f16x3'
SH = {'rpn': (2, 150, 497, 256, 512, 3, 1, 1), 'l3c2': (2, 38, 125, 256, 256, 3, 1, 1), 'l3c1': (2, 38, 125, 1024, 256, 1, 1, 0)}
B, H, W, cin, cout, k, s, p = SH[name]
dev = torch.device('cuda:0')
engine.PRECISION = prec
engine.AUTOTUNE = True
x = torch.randn(B, H, W, cin, device=dev)
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
cw = engine.prep_conv(w, torch.zeros(cout), s, p, True, device=dev)
OH, OW = engine.conv_out_hw(H, W, k, k, s, p)
y = torch.empty(B, OH, OW, cout, device=dev)
kw = {}
if split:
    x = engine.act_convert(x, 0, 1)
    kw = dict(x_fmt=1, y_fmt=1)
engine._TUNED[engine._shape_key(cw, B, H, W, OH, OW, cin, prec, (kw.get('x_fmt', 0), kw.get('y_fmt', 0), 0))] = plan
for _ in range(10):
    engine.conv2d(cw, x, B, H, W, y, OH, OW, **kw)
torch.cuda.synchronize()
