"""dev tool: time of one srcnn_dense_align call's upsample2x_kernel alone (HIP events around a dense-alignment call with R = 1 and the
rest of the call subtracted is awkward: this just times the whole call for R = 1 and R = 10 and prints rocprof-free numbers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.dense_align import KITTI_DEMO_CALIB as calib, project_box
from stereo_rcnn_amd import fixture
from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
dev = torch.device('cuda:0')
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
for R in (1, 10):
    poses = torch.tensor([[0.5 * i - 2, 1.6, 12.0 + 2 * i, 1.6, 1.5, 4.0, 0.3] for i in range(R)])
    boxes = torch.tensor([project_box(calib, p) for p in poses], dtype=torch.float32)
    kp = torch.zeros(R, 5); kp[:, 3], kp[:, 4] = boxes[:, 0], boxes[:, 2]
    a = [t.to(dev) for t in (boxes, kp, poses)]
    for _ in range(3):
        align_parallel(calib, float(info[0, 2]), l, r, *a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        align_parallel(calib, float(info[0, 2]), l, r, *a)
    e1.record(); e1.synchronize()
    print('R=%d: %.1f us per srcnn_dense_align call' % (R, e0.elapsed_time(e1) / 20 * 1e3))
