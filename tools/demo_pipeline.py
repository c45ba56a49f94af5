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


def agreement(a, b, tol):
    """fraction of a's objects with a same-box partner in b whose final (x, y, z, theta) is within tol"""
    if not a:
        return 1.0
    hit = 0
    for o in a:
        best = min(b, key=lambda q: np.abs(q['box_left'] - o['box_left']).max(), default=None)
        if best is not None and np.abs(best['box_left'] - o['box_left']).max() < 1e-3 and \
                max(np.abs(best['xyz'] - o['xyz']).max(), abs(best['theta'] - o['theta'])) < tol:
            hit += 1
    return hit / len(a)


if __name__ == '__main__':
    dev = torch.device('cuda:0')
    m = resnet(('__background__', 'Car'), 101); m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval()
    m.precision = 'f16x3'
    l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
    lu8, ru8 = [torch.from_numpy(x).to(dev) for x in fixture.synthetic_pair(3, 375, 1242)]
    calib = demo_calib()
    shape = (375, 1242, 3)
    res = {}
    for mode in ('device', 'host', 'host_py', 'scipy'):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            objs = pipeline.detect_3d(m, l, r, info, calib, shape, solver=mode)
            torch.cuda.synchronize(); dt = time.time() - t0
        res[mode] = objs
        print('one pair at a time, solver=%-6s: %d objects solved, %d aligned, %.1f ms/pair' % (mode, len(objs), sum(o['aligned'] for o in objs), dt * 1e3))
    for a, b in (('device', 'scipy'), ('host', 'scipy'), ('device', 'host')):
        print('final 3-D boxes %s vs %s: within 1e-6 %.2f, 1e-4 %.2f, 1e-2 %.2f of the objects'
              % (a, b, agreement(res[a], res[b], 1e-6), agreement(res[a], res[b], 1e-4), agreement(res[a], res[b], 1e-2)))
    N = 48
    for slots in (1, 2, 3, 4):
        frames = [(l, r, info, calib, shape, float(info[0, 2]))] * N
        list(pipeline.detect_3d_stream(m, frames[:2 * slots], slots=slots, solver='device'))
        torch.cuda.synchronize(); t0 = time.time()
        out = list(pipeline.detect_3d_stream(m, frames, slots=slots, solver='device'))
        torch.cuda.synchronize(); dt = time.time() - t0
        ok = all(len(o) == len(res['device']) and all(np.array_equal(a['xyz'], b['xyz']) for a, b in zip(res['device'], o)) for o in out)
        print('streaming, device 3-D stage, %d pairs in flight: %.2f ms/pair = %.1f pairs/s (%d objects each), identical to serial: %s'
              % (slots, dt * 1e3 / N, N / dt, len(res['device']), ok))
    for slots in (1, 2, 3, 4):
        frames = [(l, r, info, calib, shape, float(info[0, 2]))] * N
        list(pipeline.detect_3d_stream(m, frames[:2 * slots], slots=slots, solver='host'))
        torch.cuda.synchronize(); t0 = time.time()
        out = list(pipeline.detect_3d_stream(m, frames, slots=slots, solver='host'))
        torch.cuda.synchronize(); dt = time.time() - t0
        ok = all(len(o) == len(res['scipy']) and all(np.array_equal(a['xyz'], b['xyz']) for a, b in zip(res['scipy'], o)) for o in out)
        print('streaming, host Newton-CG between device stages, %d pairs in flight: %.2f ms/pair = %.1f pairs/s, identical to the scipy flow: %s'
              % (slots, dt * 1e3 / N, N / dt, ok))
    frames = [(lu8, ru8, calib)] * N
    list(pipeline.detect_3d_stream(m, frames[:6], slots=3))
    torch.cuda.synchronize(); t0 = time.time()
    out = list(pipeline.detect_3d_stream(m, frames, slots=3))
    torch.cuda.synchronize(); dt = time.time() - t0
    print('streaming from uint8 images (fused preprocessing), 3 in flight: %.2f ms/pair = %.1f pairs/s, %d objects'
          % (dt * 1e3 / N, N / dt, len(out[0])))
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    with pipeline.SolverPool(workers) as pool:
        frames = [(l, r, info, calib, shape, float(info[0, 2]))] * 24
        list(pipeline.detect_3d_stream(m, frames[:4], pool, solver='scipy'))
        torch.cuda.synchronize(); t0 = time.time()
        out = list(pipeline.detect_3d_stream(m, frames, pool, solver='scipy'))
        torch.cuda.synchronize(); dt = time.time() - t0
        print('streaming, scipy arrangement with %d solver processes (round 1 design): %.2f ms/pair = %.1f pairs/s'
              % (workers, dt * 1e3 / 24, 24 / dt))
    for o in res['device'][:5]:
        print('score %.3f box %s xyz %s theta %.2f' % (o['score'], np.round(o['box_left'], 1), np.round(o['xyz'], 2), o['theta']))
