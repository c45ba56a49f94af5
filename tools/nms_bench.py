"""Times the device NMS (pair mask + greedy scan) on 6000 score-ordered boxes at several keep rates (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import _lib

dev = torch.device('cuda:0')
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
g = torch.Generator().manual_seed(0)
for name, size in (('tiny boxes (nearly all kept)', 4.0), ('small', 40.0), ('medium', 120.0), ('large (few kept)', 400.0)):
    cx = torch.rand(n, generator=g) * 1987
    cy = torch.rand(n, generator=g) * 600
    w = size * (0.5 + torch.rand(n, generator=g))
    h = size * (0.5 + torch.rand(n, generator=g))
    dets = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, torch.linspace(1, 0, n)], 1).to(dev)
    keep = torch.zeros((n,), dtype=torch.int32, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = _lib.workspace(L.srcnn_nms_workspace_bytes(n), dev, "nms")
    def run():
        _lib.check(L.srcnn_nms(keep.data_ptr(), dets.data_ptr(), num.data_ptr(), n, 5, 0.7, ws.data_ptr(), ws.numel(),
                               _lib.stream()), "srcnn_nms")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print('%-32s n=%d kept=%5d  %.1f us per NMS (mask + scan)' % (name, n, int(num[0]), e0.elapsed_time(e1) / 20 * 1e3), flush=True)
