"""Runs ONE conv shape repeatedly with a forced plan (for rocprofv3 --pmc passes; dev tool).
usage: one_conv.py <precision> <mr> <nr> <splits> [shape-name]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine
prec, mr, nr, sp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
name = sys.argv[5] if len(sys.argv) > 5 else 'rpn'
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
engine._TUNED[engine._shape_key(cw, B, H, W, OH, OW, cin, prec)] = (mr, nr, sp)
for _ in range(10):
    engine.conv2d(cw, x, B, H, W, y, OH, OW)
torch.cuda.synchronize()
