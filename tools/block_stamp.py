"""dev tool: phase times of the fused bottleneck kernel (csrc/conv_block.hip) from its debug stamps.
usage: SRCNN_BLK_VARIANT=v SRCNN_BLK_FLAGS=f python tools/block_stamp.py <C>"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereo_rcnn_amd import _lib, engine

C = int(sys.argv[1])
B, H, W = {64: (2, 150, 497), 128: (2, 75, 249), 256: (2, 38, 125)}[C]
dev = torch.device('cuda:0')
S = _lib.FMT_SPLIT16
g = torch.Generator().manual_seed(0)
w2 = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
w3 = torch.randn(4 * C, C, 1, 1, generator=g) * (2.0 / C) ** 0.5
bn = lambda c: {'weight': torch.rand(c, generator=g) + 0.5, 'bias': torch.randn(c, generator=g) * 0.1,
                'running_mean': torch.randn(c, generator=g) * 0.1, 'running_var': torch.rand(c, generator=g) + 0.5}
c2 = engine.prep_conv(w2, None, 1, 1, True, bn(C), dev)
c3 = engine.prep_conv(w3, None, 1, 0, True, bn(4 * C), dev)
xw = engine.act_convert(torch.randn(B, H, W, 4 * C, generator=g).to(dev), 0, S)
m1 = engine.act_convert(torch.randn(B, H, W, C, generator=g).to(dev), 0, S)
y = torch.empty_like(xw)
run = lambda: engine.conv_block(c2, c3, m1, B, H, W, y, xw)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
M = B * H * W
nblk = -(-M // (16384 // C))
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.srcnn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.srcnn_debug_set_stamp_buffer.restype = None
L.srcnn_debug_set_stamp_buffer(buf.data_ptr())
run(); torch.cuda.synchronize()
L.srcnn_debug_set_stamp_buffer(None)
st = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
st = st[st[:, 6] == 1]
d = lambda a, b: np.median(st[:, b] - st[:, a]) / 100.0
print('C=%d variant=%s flags=%s: %.1f us/launch (%d workgroups) | per workgroup (median, us): set-up %.1f, conv2 loop %.1f, hand-off %.1f, '
      'conv3 loop %.1f (of which the 4 epilogues %.1f), total %.1f; launch span %.1f'
      % (C, os.environ.get('SRCNN_BLK_VARIANT', '-'), os.environ.get('SRCNN_BLK_FLAGS', '-'), us, nblk, d(0, 1), d(1, 2), d(2, 3), d(3, 4),
         np.median(st[:, 5]) / 100.0, d(0, 4), (st[:, 4].max() - st[:, 0].min()) / 100.0))
