"""Micro-benchmark of the conv engine on the network's GEMM shapes (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine

dev = torch.device('cuda:0')
SHAPES = [
    # name, B, H, W, cin, cout, k, stride, pad
    ('l1.conv2 3x3 64', 2, 150, 497, 64, 64, 3, 1, 1),
    ('l1.conv3 1x1 64->256', 2, 150, 497, 64, 256, 1, 1, 0),
    ('l1.conv1 1x1 256->64', 2, 150, 497, 256, 64, 1, 1, 0),
    ('l2.conv2 3x3 128', 2, 75, 249, 128, 128, 3, 1, 1),
    ('l2.conv3 1x1 128->512', 2, 75, 249, 128, 512, 1, 1, 0),
    ('l3.conv2 3x3 256', 2, 38, 125, 256, 256, 3, 1, 1),
    ('l3.conv3 1x1 256->1024', 2, 38, 125, 256, 1024, 1, 1, 0),
    ('l3.conv1 1x1 1024->256', 2, 38, 125, 1024, 256, 1, 1, 0),
    ('l4.conv2 3x3 512', 2, 19, 63, 512, 512, 3, 1, 1),
    ('l4.conv3 1x1 512->2048', 2, 19, 63, 512, 2048, 1, 1, 0),
    ('fpn.smooth3 3x3 256 P2', 2, 150, 497, 256, 256, 3, 1, 1),
    ('rpn.conv 3x3 256->512 P2', 2, 150, 497, 256, 512, 3, 1, 1),
    ('rpn.head 1x1 1024->24 P2', 1, 150, 497, 1024, 24, 1, 1, 0),
    ('box.top0 GEMM 300x25088x2048', 300, 1, 1, 25088, 2048, 1, 1, 0),
    ('box.top3 GEMM 300x2048x2048', 300, 1, 1, 2048, 2048, 1, 1, 0),
    ('kpts 3x3 256 (300x14x14)', 300, 14, 14, 256, 256, 3, 1, 1),
    ('hbm fpn.lateral3 1x1 256->256 P2', 2, 150, 497, 256, 256, 1, 1, 0),
    ('hbm fpn.lateral2 1x1 512->256 P3', 2, 75, 249, 512, 256, 1, 1, 0),
    ('hbm l2.downsample 1x1/2 256->512', 2, 150, 497, 256, 512, 1, 2, 0),
    ('hbm l1.downsample 1x1 64->256', 2, 150, 497, 64, 256, 1, 1, 0),
]
PREC = sys.argv[1] if len(sys.argv) > 1 else 'f32'
SPLIT = PREC == 'f16s'          # f16x3 arithmetic with SPLIT16 activations (DMA-to-LDS kernel)
if SPLIT:
    PREC = 'f16x3'
engine.PRECISION = PREC
print('precision', PREC, 'split16 activations' if SPLIT else '')
ONLY = os.environ.get('ONLY')          # substring filter of the shape names
for name, B, H, W, cin, cout, k, s, p in SHAPES:
    if ONLY and ONLY not in name:
        continue
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    cw = engine.prep_conv(w, torch.zeros(cout), s, p, True, device=dev)
    OH, OW = engine.conv_out_hw(H, W, k, k, s, p)
    y = torch.empty(B, OH, OW, cout, device=dev)
    kw = {}
    if SPLIT:
        x = engine.act_convert(x, 0, 1)
        kw = dict(x_fmt=1, y_fmt=1 if cout % 8 == 0 else 0)
    for _ in range(3):
        engine.conv2d(cw, x, B, H, W, y, OH, OW, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        engine.conv2d(cw, x, B, H, W, y, OH, OW, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * OH * OW * cout * cin * k * k
    plan = [v for kk, v in engine._TUNED.items() if kk[0] == PREC][-1] if engine._TUNED else None
    print('%-34s M=%7d N=%5d K=%6d  %8.3f ms  %7.1f TFLOP/s  plan(mr,nr,waves,stages,splits)=%s' % (name, B * OH * OW, cout, cin * k * k, ms, fl / ms / 1e9, plan), flush=True)
    if os.environ.get('SWEEP'):
        log = sorted(list(engine._TUNE_LOG.values())[-1], key=lambda e: e[1])
        best_per_tile = {}
        for pl, t in log:
            best_per_tile.setdefault(pl[:4], (pl[4], t))
        print('      ' + '  '.join('%s/s%d:%.0fTF' % (''.join(map(str, tl)), sp, fl / t / 1e9) for tl, (sp, t) in best_per_tile.items()), flush=True)
