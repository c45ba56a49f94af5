"""Where does a conv_f16s workgroup spend its time?  (dev tool; uses the library's debug stamp hook)

usage: stamp_conv.py <mr> <nr> <waves> <stages> <splits> <shape-name>
Each workgroup records s_memtime at: entry, after address set-up, after the first K tile landed, after the K loop,
after the accumulator tile is in LDS, at the end.  Prints medians per phase in shader clocks and the kernel's span.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereo_rcnn_amd import engine, _lib

plan = tuple(int(v) for v in sys.argv[1:6])
name = sys.argv[6] if len(sys.argv) > 6 else 'l3c3'
SH = {'rpn': (2, 150, 497, 256, 512, 3, 1, 1), 'l3c2': (2, 38, 125, 256, 256, 3, 1, 1), 'l3c1': (2, 38, 125, 1024, 256, 1, 1, 0),
      'l3c3': (2, 38, 125, 256, 1024, 1, 1, 0), 'l1c3': (2, 150, 497, 64, 256, 1, 1, 0), 'l2c3': (2, 75, 249, 128, 512, 1, 1, 0),
      'smooth': (2, 150, 497, 256, 256, 3, 1, 1)}
B, H, W, cin, cout, k, s, p = SH[name]
dev = torch.device('cuda:0')
engine.PRECISION = 'f16x3'
x = engine.act_convert(torch.randn(B, H, W, cin, device=dev), 0, 1)
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
cw = engine.prep_conv(w, torch.zeros(cout), s, p, True, device=dev)
OH, OW = engine.conv_out_hw(H, W, k, k, s, p)
y = torch.empty(B, OH, OW, cout, device=dev)
engine._TUNED[engine._shape_key(cw, B, H, W, OH, OW, cin, 'f16x3', (1, 1, 0))] = plan
run = lambda: engine.conv2d(cw, x, B, H, W, y, OH, OW, x_fmt=1, y_fmt=1)
for _ in range(3):
    run()
torch.cuda.synchronize()
M = B * OH * OW
nblk = -(-M // (64 * plan[0])) * -(-cout // (64 * plan[1])) * plan[4]
buf = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
L = _lib.lib()
L.srcnn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.srcnn_debug_set_stamp_buffer.restype = None
L.srcnn_debug_set_stamp_buffer(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
L.srcnn_debug_set_stamp_buffer(None)
st = buf.cpu().numpy().reshape(nblk, 16).astype(np.int64)
st = st[(st[:, 7] & 1) == 1]
xcc = (st[:, 7] >> 8) & 0xF
cu = ((st[:, 6] >> 8) & 0xF) | (((st[:, 6] >> 13) & 0x7) << 4) | (((st[:, 6] >> 12) & 1) << 7)   # cu_id, se_id, sh_id
ph = np.diff(st[:, :6], axis=1)
names = ['set-up', 'first tile', 'K loop', 'acc -> LDS', 'bias/res/store']
print('%s plan %s: %d workgroups (%d stamped), kernel %.1f us by events (includes the launch call)' % (name, plan, nblk, len(st), e0.elapsed_time(e1) * 1e3))
for i, n in enumerate(names):
    print('  %-16s median %7d clk  p90 %7d' % (n, np.median(ph[:, i]), np.percentile(ph[:, i], 90)))
print('  of the K loop, wave 0 sat in the vmcnt wait (data not landed) median %7d clk, in the barrier median %7d clk' % (np.median(st[:, 10]), np.median(st[:, 11])))
tot = st[:, 5] - st[:, 0]
print('  %-16s median %7d clk  p90 %7d' % ('workgroup total', np.median(tot), np.percentile(tot, 90)))
# launch shape on the chip-wide 100 MHz clock
rt0, rt1 = st[:, 8], st[:, 9]
base = rt0.min()
span = (rt1.max() - base) / 100.0
starts = np.sort(rt0 - base) / 100.0
ends = np.sort(rt1 - base) / 100.0
ncu = len(set(zip(xcc.tolist(), cu.tolist())))
print('  kernel span %.1f us on %d CUs; %.2f workgroups in flight per CU on average' % (span, ncu, ((rt1 - rt0).sum() / 100.0) / span / ncu))
q = lambda a, f: a[min(len(a) - 1, int(len(a) * f))]
print('  starts (us): 10%% %.1f  25%% %.1f  50%% %.1f  75%% %.1f  90%% %.1f  last %.1f' % tuple(q(starts, f) for f in (.1, .25, .5, .75, .9, 1.0)))
print('  ends   (us): 10%% %.1f  25%% %.1f  50%% %.1f  75%% %.1f  90%% %.1f  last %.1f' % tuple(q(ends, f) for f in (.1, .25, .5, .75, .9, 1.0)))
print('  workgroup duration (us): median %.1f  p90 %.1f' % (np.median(rt1 - rt0) / 100.0, np.percentile(rt1 - rt0, 90) / 100.0))
per_cu = {}
for x, c in zip(xcc.tolist(), cu.tolist()):
    per_cu[(x, c)] = per_cu.get((x, c), 0) + 1
v = np.array(list(per_cu.values()))
print('  workgroups per CU: min %d  median %d  max %d' % (v.min(), np.median(v), v.max()))
