"""Tuner-objective probe (dev tool): every candidate plan of the trunk / FPN / head GEMM shapes timed ALONE on the chip and with
three copies of the launch in flight on three HIP streams (engine.TUNE_MODE 'isolated' vs 'concurrent').  Prints, per shape,
the plan each objective picks and, per tile variant, us per launch under both objectives -- the evidence behind
profiles/tune_objective_r04.txt.   usage: [ONLY=substr] python tools/conc_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine

dev = torch.device('cuda:0')
SHAPES = [
    # name, B, H, W, cin, cout, k, stride, pad, residual
    ('l1.conv2 3x3 64', 2, 150, 497, 64, 64, 3, 1, 1, False),
    ('l1.conv3 1x1 64->256 +res', 2, 150, 497, 64, 256, 1, 1, 0, True),
    ('l1.conv1 1x1 256->64', 2, 150, 497, 256, 64, 1, 1, 0, False),
    ('l2.conv2 3x3 128', 2, 75, 249, 128, 128, 3, 1, 1, False),
    ('l2.conv3 1x1 128->512 +res', 2, 75, 249, 128, 512, 1, 1, 0, True),
    ('l2.conv1 1x1 512->128', 2, 75, 249, 512, 128, 1, 1, 0, False),
    ('l3.conv1 1x1 1024->256', 2, 38, 125, 1024, 256, 1, 1, 0, False),
    ('l3.conv2 3x3 256', 2, 38, 125, 256, 256, 3, 1, 1, False),
    ('l3.conv3 1x1 256->1024 +res', 2, 38, 125, 256, 1024, 1, 1, 0, True),
    ('l4.conv1 1x1 2048->512', 2, 19, 63, 2048, 512, 1, 1, 0, False),
    ('l4.conv2 3x3 512', 2, 19, 63, 512, 512, 3, 1, 1, False),
    ('l4.conv3 1x1 512->2048 +res', 2, 19, 63, 512, 2048, 1, 1, 0, True),
    ('fpn.smooth3 3x3 256 P2', 2, 150, 497, 256, 256, 3, 1, 1, False),
    ('fpn.smooth2 3x3 256 P3', 2, 75, 249, 256, 256, 3, 1, 1, False),
    ('rpn.conv 3x3 256->512 P2', 2, 150, 497, 256, 512, 3, 1, 1, False),
    ('rpn.conv 3x3 256->512 P3', 2, 75, 249, 256, 512, 3, 1, 1, False),
    ('kpts 3x3 256 (300x14x14)', 300, 14, 14, 256, 256, 3, 1, 1, False),
    ('box.top0 GEMM 300x25088x2048', 300, 1, 1, 25088, 2048, 1, 1, 0, False),
]
ONLY = os.environ.get('ONLY')
engine.PRECISION = 'f16x3'
tot = {'isolated': [0.0, 0.0], 'concurrent': [0.0, 0.0]}      # objective -> [sum of its pick's isolated us, sum of its pick's concurrent us]
for name, B, H, W, cin, cout, k, s, p, res in SHAPES:
    if ONLY and ONLY not in name:
        continue
    x = engine.act_convert(torch.randn(B, H, W, cin, device=dev), 0, 1)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    cw = engine.prep_conv(w, torch.zeros(cout), s, p, True, device=dev)
    OH, OW = engine.conv_out_hw(H, W, k, k, s, p)
    y = torch.empty(B, OH, OW, cout, device=dev)
    r = engine.act_convert(torch.randn(B, OH, OW, cout, device=dev), 0, 1) if res else None
    logs, picks = {}, {}
    for mode in ('isolated', 'concurrent'):
        engine.set_tune_mode(mode, 3)
        n0 = set(engine._TUNE_LOG)
        engine.conv2d(cw, x, B, H, W, y, OH, OW, x_fmt=1, y_fmt=1, residual=r, res_fmt=1 if res else 0)
        torch.cuda.synchronize()
        key = [kk for kk in engine._TUNE_LOG if kk not in n0][0]
        logs[mode] = dict(engine._TUNE_LOG[key])
        picks[mode] = engine._TUNED[key]
    fl = 2.0 * B * OH * OW * cout * cin * k * k
    print('%-32s M=%7d N=%5d K=%6d  %6.2f GFLOP' % (name, B * OH * OW, cout, cin * k * k, fl / 1e9))
    for mode in ('isolated', 'concurrent'):
        pl = picks[mode]
        ti, tc = logs['isolated'][pl] * 1e3, logs['concurrent'][pl] * 1e3
        tot[mode][0] += ti
        tot[mode][1] += tc
        print('   %-10s objective picks %-16s: %7.1f us alone (%5.0f TF), %7.1f us per launch with 3 in flight (%5.0f TF)'
              % (mode, pl, ti, fl / ti / 1e6, tc, fl / tc / 1e6))
    rows = sorted(logs['concurrent'], key=lambda pl: logs['concurrent'][pl])[:8]
    print('      ' + '  '.join('%d%d%d%d/s%d:%.0f|%.0f' % (pl + (logs['isolated'][pl] * 1e3, logs['concurrent'][pl] * 1e3)) for pl in rows)
          + '   (plan: us alone | us per launch, 3 in flight)', flush=True)
print('sum over these shapes (one launch each): isolated-objective picks %.0f us alone / %.0f us with 3 in flight; '
      'concurrent-objective picks %.0f / %.0f' % (tot['isolated'][0], tot['isolated'][1], tot['concurrent'][0], tot['concurrent'][1]))
