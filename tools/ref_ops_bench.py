"""The reference's own NMS / ROIAlign kernels (oracle/_ref, built unchanged for gfx950) timed next to the product's on the
same MI355X and inputs (dev tool; the oracle package is used here as a checker/baseline only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ref_ops
from stereo_rcnn_amd import _lib
from stereo_rcnn_amd.model.nms.nms_gpu import nms_gpu
from stereo_rcnn_amd.model.roi_align.functions.roi_align import RoIAlignFunction

dev = torch.device('cuda:0')


def wall(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


g = np.random.default_rng(0)
for size, label in ((40.0, 'mostly kept'), (150.0, 'about a third kept')):
    n = 6000
    cx, cy = g.uniform(0, 1987, n), g.uniform(0, 600, n)
    w, h = size * g.uniform(0.5, 1.5, n), size * g.uniform(0.5, 1.5, n)
    d = torch.from_numpy(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, np.linspace(1, 0, n)], 1).astype(np.float32)).to(dev)
    kr = ref_ops.nms(d, 0.7)
    kp = nms_gpu(d, 0.7).view(-1)
    assert torch.equal(kr, kp)
    print('NMS 6000 boxes (%s, %d kept): reference nms_cuda_compute %.0f us   product srcnn_nms %.0f us   (identical keep lists)'
          % (label, int(kr.numel()), wall(lambda: ref_ops.nms(d, 0.7)), wall(lambda: nms_gpu(d, 0.7))))
feat = torch.randn(2, 256, 150, 497, device=dev)
x1, y1 = g.uniform(0, 1800, 300), g.uniform(0, 500, 300)
rois = torch.from_numpy(np.stack([g.integers(0, 2, 300), x1, y1, x1 + g.uniform(20, 300, 300), y1 + g.uniform(20, 150, 300)], 1).astype(np.float32)).to(dev)
for a in (8, 15):
    o_r = ref_ops.roi_align_forward(feat, rois, a, a, 0.25, 'nofma')
    f = RoIAlignFunction(a, a, 0.25)
    assert torch.equal(f(feat, rois), o_r)
    print('ROIAlign 300 rois x 256 ch x %dx%d on a 150x497 map: reference ROIAlignForward %.0f us   product roi_align_forward_cuda %.0f us   (bit-equal)'
          % (a, a, wall(lambda: ref_ops.roi_align_forward(feat, rois, a, a, 0.25)), wall(lambda: f(feat, rois))))
