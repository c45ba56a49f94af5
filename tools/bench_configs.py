"""Throughput of the other BASELINE.json configs on one MI355X (dev tool; not the headline bench).
  config 3: ResNet-101 FPN + stereo RPN + ROIAlign + dense_align, batch=8, full pipeline
  config 5: ResNet-50 trunk, 2x input resolution (network input 1200x3974), batch=4
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereo_rcnn_amd import fixture, pipeline, postprocess
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
from tools.demo_pipeline import demo_calib

dev = torch.device('cuda:0')


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


def config3(B=8):
    m = resnet(('__background__', 'Car'), 101); m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval(); m.precision = 'f16x3'
    pairs = [fixture.make_inputs(3 + i, 375, 1242) for i in range(B)]
    l = torch.cat([p[0] for p in pairs]).to(dev); r = torch.cat([p[1] for p in pairs]).to(dev)
    info = torch.cat([p[2] for p in pairs]).to(dev)
    calib = demo_calib()
    # synthetic objects for the dense-alignment stage (10 per image); the scipy solvers are host
    # code and are timed separately in tools/demo_pipeline.py
    from oracle.dense_align import project_box   # helper only (box projection), not on the timed path
    rng = np.random.default_rng(0)
    poses = []
    for _ in range(10):
        z = rng.uniform(8, 50)
        poses.append([rng.uniform(-0.5, 0.5) * z * 0.7, 1.6, z, 1.6, 1.5, 4.0, rng.uniform(-3.1, 3.1)])
    poses = torch.tensor(poses, dtype=torch.float32)
    boxes = torch.tensor([project_box(calib, p) for p in poses], dtype=torch.float32)
    boxes[:, 0::2].clamp_(0, 1241); boxes[:, 1::2].clamp_(0, 374)
    kp = torch.zeros(10, 5); kp[:, 3] = boxes[:, 0]; kp[:, 4] = boxes[:, 2]
    boxes, kp, poses = boxes.to(dev), kp.to(dev), poses.to(dev)

    def step():
        with torch.no_grad():
            out = m(l, r, info)
            for b in range(B):          # decode / dense-align looped per image, as the reference's B=1 post-processing implies
                det = postprocess.decode_detections(out[0][b:b + 1], out[1][b:b + 1], out[2][b:b + 1], out[3][b:b + 1],
                                                    out[4][b:b + 1], out[5][b * 300:(b + 1) * 300],
                                                    out[6][b * 300:(b + 1) * 300], out[7][b * 300:(b + 1) * 300], info[b:b + 1])
                postprocess.class_nms_device(det, 1, 0.05)
                align_parallel(calib, 1.6, l[b:b + 1], r[b:b + 1], boxes, kp, poses)
    dt = timed(step, 5)
    print('config 3 (R-101, batch=%d, forward + decode + class NMS + dense-align of 10 objects/image): %.1f ms/batch, %.1f pairs/s'
          % (B, dt * 1e3, B / dt), flush=True)


def config5(B=4):
    m = resnet(('__background__', 'Car'), 50); m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(5, layers=fixture.R50)); m.cuda(); m.eval(); m.precision = 'f16x3'
    # 2484x750 source images at 2x the default test scale -> 1200x3974 network input (BASELINE.json configs[4])
    g = torch.Generator().manual_seed(5)
    l = (torch.randn(B, 3, 1200, 3974, generator=g) * 50).to(dev)
    r = (torch.randn(B, 3, 1200, 3974, generator=g) * 50).to(dev)
    info = torch.tensor([[1200., 3974., 1.6]] * B).to(dev)
    print('config 5 network input', tuple(l.shape), flush=True)

    def step():
        with torch.no_grad():
            m(l, r, info)
    dt = timed(step, 3)
    print('config 5 (R-50, network input %dx%d, batch=%d, forward): %.1f ms/batch, %.2f pairs/s, peak memory %.1f GB'
          % (l.shape[3], l.shape[2], B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)


if __name__ == '__main__':
    which = sys.argv[1:] or ['3', '5']
    if '3' in which:
        config3()
    if '5' in which:
        config5()
