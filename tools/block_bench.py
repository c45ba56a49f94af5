"""dev tool: the fused bottleneck tail (srcnn_conv_block) vs the two stand-alone launches (autotuned plans), trunk shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import _lib, engine

dev = torch.device('cuda:0')
S = _lib.FMT_SPLIT16
g = torch.Generator().manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, C, B, H, W in (('layer1', 64, 2, 150, 497), ('layer2', 128, 2, 75, 249), ('layer3', 256, 2, 38, 125)):
    w1 = torch.randn(C, 4 * C, 1, 1, generator=g) * (2.0 / (4 * C)) ** 0.5
    w2 = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    w3 = torch.randn(4 * C, C, 1, 1, generator=g) * (2.0 / C) ** 0.5
    bn = lambda c: {'weight': torch.rand(c, generator=g) + 0.5, 'bias': torch.randn(c, generator=g) * 0.1,
                    'running_mean': torch.randn(c, generator=g) * 0.1, 'running_var': torch.rand(c, generator=g) + 0.5}
    c1 = engine.prep_conv(w1, None, 1, 0, True, bn(C), dev)
    c2 = engine.prep_conv(w2, None, 1, 1, True, bn(C), dev)
    c3 = engine.prep_conv(w3, None, 1, 0, True, bn(4 * C), dev)
    xw = engine.act_convert(torch.randn(B, H, W, 4 * C, generator=g).to(dev), 0, S)
    m1 = torch.empty(B, H, W, C, device=dev)
    m2 = torch.empty_like(m1)
    y = torch.empty_like(xw)
    f1 = lambda: engine.conv2d(c1, xw, B, H, W, m1, H, W, precision='f16x3', x_fmt=S, y_fmt=S)
    f2 = lambda: engine.conv2d(c2, m1, B, H, W, m2, H, W, precision='f16x3', x_fmt=S, y_fmt=S)
    f3 = lambda: engine.conv2d(c3, m2, B, H, W, y, H, W, residual=xw, precision='f16x3', x_fmt=S, y_fmt=S, res_fmt=S)
    fb = lambda: engine.conv_block(c2, c3, m1, B, H, W, y, xw)
    t1, t2, t3, tb = timeit(f1), timeit(f2), timeit(f3), timeit(fb)
    M = B * H * W
    fl = 2.0 * M * C * (9 * C + 4 * C)
    print('%s C=%3d M=%6d: conv1 %.1f us | conv2 %.1f + conv3 %.1f = %.1f us  vs fused %.1f us (%.0f TF algorithmic, %d workgroups)'
          % (name, C, M, t1, t2, t3, t2 + t3, tb, fl / tb / 1e6, -(-M // (16384 // C))), flush=True)
