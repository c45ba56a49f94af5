"""dev tool: is srcnn_dense_align repeatable under concurrency?  Same inputs on 3 streams, many times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereo_rcnn_amd import fixture, pipeline, _lib
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
from tools.demo_pipeline import demo_calib

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval(); m.precision = 'f16x3'
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
calib = demo_calib()
with torch.no_grad():
    out = m(l, r, info)
    st = pipeline.launch_3d(out, l, r, info, float(info[0, 2]), calib, (375, 1242, 3))
    st.event.synchronize()
    boxes, borders, poses, valid = st.boxes.clone(), st.borders.clone(), st.poses.clone(), st.valid.clone()
    kp = torch.zeros(boxes.shape[0], 5, device=dev)
    kp[:, 3:5] = borders
    ref_s, ref_d = align_parallel(calib, float(info[0, 2]), l, r, boxes, kp, poses, valid=valid)
    torch.cuda.synchronize()
    ref_s, ref_d = ref_s.clone(), ref_d.clone()
    L = _lib.lib()
    rec0 = st.rec.clone()
    cal = (float(calib.p2[0, 0]), float(calib.p2[0, 2]), float(calib.p2[1, 2]), float(calib.p2[0, 3] - calib.p3[0, 3]))
    recs = [rec0.clone() for _ in range(3)]
    states = [torch.zeros((300, 4), dtype=torch.float64, device=dev) for _ in range(3)]
    for mode in ('dense_align alone on 3 streams', 'dense_align on 2 streams beside forwards on a third',
                 'solve4 then dense_align on each of 3 streams', 'dense_align on 2 streams beside solve4 kernels on a third'):
        streams = [torch.cuda.Stream() for _ in range(3)]
        bad = 0
        res = []
        for k in range(150):
            s = streams[k % 3]
            with torch.cuda.stream(s):
                if mode.startswith('solve4 then'):
                    _lib.check(L.srcnn_solve_4dof(recs[k % 3].data_ptr(), 300, _lib.REC_COLS, 375, 1242, *cal, 0.05, states[k % 3].data_ptr(), _lib.stream()))
                if mode.endswith('forwards on a third') and k % 3 == 2:
                    m(l, r, info, slot=1)
                elif mode.endswith('solve4 kernels on a third') and k % 3 == 2:
                    _lib.check(L.srcnn_solve_4dof(recs[2].data_ptr(), 300, _lib.REC_COLS, 375, 1242, *cal, 0.05, states[2].data_ptr(), _lib.stream()))
                else:
                    a, b = align_parallel(calib, float(info[0, 2]), l, r, boxes, kp, poses, valid=valid)
                    res.append((k, a, b))
        torch.cuda.synchronize()
        for k, a, b in res:
            ok = valid > 0
            if not (torch.equal(a[ok], ref_s[ok]) and torch.equal(b[ok][ref_s[ok] > 0], ref_d[ok][ref_s[ok] > 0])):
                bad += 1
                j = torch.nonzero(ok & ((a != ref_s) | (b != ref_d)))[:, 0].tolist()
                print('  run %d: objects %s status %s vs %s dis %s vs %s' % (k, j, a[j].tolist(), ref_s[j].tolist(), b[j].tolist(), ref_d[j].tolist()))
        print('%s: %d of %d runs differ' % (mode, bad, len(res)), flush=True)
