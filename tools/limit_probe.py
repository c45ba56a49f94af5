"""dev probe: duration of the keypoint tower's launches against the device-side row limit (rois), explicit plans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
R = 300


def run(name, cw, s, cout_buf, plans, mode_out=1):
    x = engine.act_convert((torch.randn(R, s, s, 256, generator=g) * 0.5).to(dev), 0, 1)
    OH = s
    y = torch.zeros((R, OH, OH, cout_buf), device=dev)
    for plan in plans:
        row = []
        for lim in (300, 128, 64, 32, 8, 1):
            t = torch.tensor([lim], dtype=torch.int32, device=dev)
            kw = dict(precision='f16x3', x_fmt=1, y_fmt=mode_out, plan=plan, m_limit=t, m_limit_mul=s * s)
            for _ in range(3):
                engine.conv2d(cw, x, R, s, s, y, OH, OH, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                engine.conv2d(cw, x, R, s, s, y, OH, OH, **kw)
            e1.record(); e1.synchronize()
            row.append('%d:%.1f' % (lim, e0.elapsed_time(e1) * 100))
        print(name, plan, ' '.join(row), 'us', flush=True)


w3 = engine.prep_conv(torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g), 1, 1, True, device=dev)
run('3x3 256->256 @14', w3, 14, 256, [(4, 4, 8, 2, 1), (4, 4, 8, 2, 4), (2, 2, 8, 4, 1), (2, 2, 8, 4, 4), (2, 2, 4, 2, 1), (1, 1, 4, 4, 1)])
w1 = engine.prep_conv(torch.randn(6, 256, 1, 1, generator=g) / 16, torch.randn(6, generator=g), 1, 0, False, device=dev)
run('1x1 256->6 @28', w1, 28, 6, [(1, 1, 4, 2, 1), (2, 1, 4, 2, 1)], mode_out=0)
