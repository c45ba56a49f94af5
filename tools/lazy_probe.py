"""dev probe: forward + decode + class NMS with the keypoint branch inside the forward (all 300 rois) vs after class NMS on the
kept detections only (Plan.kpts_for_kept), three pairs in flight and one at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, postprocess as hpost
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture(); m.load_state_dict(fixture.make_state_dict(3)); m.cuda().eval()
m.precision = 'f16x3'; m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]

def step(slot, lazy):
    out = m(l, r, info, slot=slot, kpts=not lazy)
    det = hpost.decode_detections(out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], info)
    keep_idx, num = hpost.class_nms_device(det, 1, 0.05)
    if lazy:
        plan = m._get_plan(1, l.shape[2], l.shape[3], slot)
        plan.kpts_for_kept(out[0][0].contiguous(), keep_idx, num, info, det['kpts'], 'f16x3')
    return num

with torch.no_grad():
    for S in (1, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        for lazy in (False, True, False, True):
            for k in range(2 * S):
                with torch.cuda.stream(streams[k % S]):
                    n = step(k % S, lazy)
            torch.cuda.synchronize()
            t = time.perf_counter(); N = 60
            for k in range(N):
                with torch.cuda.stream(streams[k % S]):
                    n = step(k % S, lazy)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / N
            print('%d in flight, keypoint branch %s: %.3f ms/pair = %.1f pairs/s (kept %d)' % (S, 'after NMS on kept' if lazy else 'in the forward ', dt * 1e3, 1 / dt, int(n[0])), flush=True)

from stereo_rcnn_amd import engine
for key, log in engine._TUNE_LOG.items():
    if 'lim' in key:
        print('row-limited launch', key[1:15], sorted(log, key=lambda r: r[1])[:12])
