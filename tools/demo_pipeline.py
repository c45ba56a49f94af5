"""demo.py-equivalent on a synthetic pair (no trained weights / KITTI data exist offline): full flow
network -> decode -> NMS -> 3-D solve -> dense alignment -> rectification, with timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereo_rcnn_amd import fixture, pipeline
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
from stereo_rcnn_amd.model.utils import kitti_utils


class Calib(object):
    pass


def demo_calib():
    c = kitti_utils.FrameCalibrationData()
    c.p2 = np.array([721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884]).reshape(3, 4)
    c.p3 = np.array([721.5377, 0, 609.5593, -339.5242, 0, 721.5377, 172.854, 2.199936, 0, 0, 1, 0.002729905]).reshape(3, 4)
    c.t_cam2_cam0 = np.array([c.p2[0, 3] / c.p2[0, 0], 0, 0])
    return c


if __name__ == '__main__':
    dev = torch.device('cuda:0')
    m = resnet(('__background__', 'Car'), 101); m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval()
    m.precision = 'f16x3'
    l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
    calib = demo_calib()
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        objs = pipeline.detect_3d(m, l, r, info, calib, (375, 1242, 3))
        torch.cuda.synchronize(); dt = time.time() - t0
        print('pass %d (serial solvers): %d objects solved, %d aligned, %.1f ms' % (it, len(objs), sum(o['aligned'] for o in objs), dt * 1e3))
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    with pipeline.SolverPool(workers) as pool:
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            objs2 = pipeline.detect_3d(m, l, r, info, calib, (375, 1242, 3), pool=pool)
            torch.cuda.synchronize(); dt = time.time() - t0
            print('pass %d (%d solver processes): %d objects solved, %d aligned, %.1f ms' % (it, workers, len(objs2), sum(o['aligned'] for o in objs2), dt * 1e3))
        same = len(objs) == len(objs2) and all(np.array_equal(a['xyz'], b['xyz']) for a, b in zip(objs, objs2))
        print('pool results identical to serial:', same)
        # streaming form: forward of pair k+1, solver stages of pairs k, k-1 .. overlap
        frames = [(l, r, info, calib, (375, 1242, 3), float(info[0, 2]))] * 24
        list(pipeline.detect_3d_stream(m, frames[:4], pool))
        torch.cuda.synchronize(); t0 = time.time()
        res = list(pipeline.detect_3d_stream(m, frames, pool))
        torch.cuda.synchronize(); dt = time.time() - t0
        ok = all(len(o) == len(objs) and all(np.array_equal(a['xyz'], b['xyz']) for a, b in zip(objs, o)) for o in res)
        print('streaming pipeline: %d pairs in %.1f ms = %.1f ms/pair = %.1f pairs/s (%d objects each), identical to serial: %s'
              % (len(res), dt * 1e3, dt * 1e3 / len(res), len(res) / dt, len(objs), ok))
    for o in objs[:5]:
        print('score %.3f box %s xyz %s theta %.2f' % (o['score'], np.round(o['box_left'], 1), np.round(o['xyz'], 2), o['theta']))
