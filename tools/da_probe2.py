"""dev tool: when dense_align beside a forward gives the rare other answer, which workspace region differs from a clean call?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereo_rcnn_amd import fixture, pipeline, _lib
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel, MAX_PIXELS
from tools.demo_pipeline import demo_calib

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval(); m.precision = 'f16x3'
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
calib = demo_calib()
al = lambda v: (v + 255) // 256 * 256
H, W, R = 600, 1987, 300
img = 3 * 2 * H * 2 * W * 4
regions, off = [], 0
for name, size in (('up_l', img), ('up_r', img), ('uvz', R * MAX_PIXELS * 12), ('cnt', 2 * R * 4), ('left_val', R * MAX_PIXELS * 12),
                   ('depth_enum0', 50 * R * 4), ('cost0', 50 * R * 4), ('best0', R * 4),
                   ('depth_enum1', 50 * R * 4), ('cost1', 50 * R * 4), ('best1', R * 4)):
    regions.append((name, off, size)); off += al(size)
with torch.no_grad():
    out = m(l, r, info)
    st = pipeline.launch_3d(out, l, r, info, float(info[0, 2]), calib, (375, 1242, 3))
    st.event.synchronize()
    boxes, borders, poses, valid = st.boxes.clone(), st.borders.clone(), st.poses.clone(), st.valid.clone()
    kp = torch.zeros(boxes.shape[0], 5, device=dev); kp[:, 3:5] = borders
    sA, sF = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sA):
        ref_s, ref_d = align_parallel(calib, float(info[0, 2]), l, r, boxes, kp, poses, valid=valid)
    torch.cuda.synchronize()
    wsA = _lib._workspaces[(str(dev), 'dense_align', sA.cuda_stream)].buf
    ref_ws = wsA.clone()
    ref_s, ref_d = ref_s.clone(), ref_d.clone()
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    plan = m._get_plan(1, 600, 1987, 1)
    bad = 0
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1200):
        with torch.cuda.stream(sF):
            if which == 'all':
                m(l, r, info, slot=1)
            else:
                plan.fmt = 1
                from stereo_rcnn_amd import engine
                engine.PRECISION = 'f16x3'
                getattr(plan, which)()
        with torch.cuda.stream(sA):
            torch.cuda._sleep(int((k % 40) * 4e5))          # start the alignment at a different point of the forward each time
            a, b = align_parallel(calib, float(info[0, 2]), l, r, boxes, kp, poses, valid=valid)
        torch.cuda.synchronize()
        ok = valid > 0
        if not (torch.equal(a[ok], ref_s[ok]) and torch.equal(b[ok][ref_s[ok] > 0], ref_d[ok][ref_s[ok] > 0])):
            bad += 1
            j = torch.nonzero(ok & ((a != ref_s) | (b != ref_d)))[:, 0].tolist()
            print('iteration %d: objects %s dis %s vs %s' % (k, j, b[j].tolist(), ref_d[j].tolist()))
            for name, o, size in regions:
                x, y = wsA[o:o + size].view(torch.float32), ref_ws[o:o + size].view(torch.float32)
                d = torch.nonzero(x != y)[:, 0]
                if d.numel():
                    print('   %-10s %d of %d words differ, first at word %d (%.6g vs %.6g), last at %d' % (name, d.numel(), size // 4, int(d[0]), float(x[d[0]]), float(y[d[0]]), int(d[-1])))
            for obj in j:
                n1, n0 = int(wsA[regions[3][1]:regions[3][1] + 1200].view(torch.int32)[obj]), int(ref_ws[regions[3][1]:regions[3][1] + 1200].view(torch.int32)[obj])
                o = regions[2][1] + obj * MAX_PIXELS * 12
                x = wsA[o:o + n1 * 12].view(torch.float32).view(-1, 3).cpu().numpy()
                y = ref_ws[o:o + n0 * 12].view(torch.float32).view(-1, 3).cpu().numpy()
                sx, sy = set(map(tuple, x[:, :2].tolist())), set(map(tuple, y[:, :2].tolist()))
                print('   object %d: %d samples now, %d in the clean run; pose %s box %s borders %s' % (obj, n1, n0, poses[obj].tolist(), boxes[obj].tolist(), borders[obj].tolist()))
                us, vs = sorted(set(y[:, 0].tolist())), sorted(set(y[:, 1].tolist()))
                print('   clean lattice u %g..%g (%d), v %g..%g (%d)' % (us[0], us[-1], len(us), vs[0], vs[-1], len(vs)))
                for nm, st_ in (('missing', sy - sx), ('extra', sx - sy)):
                    rows = {}
                    for u, v in st_:
                        rows.setdefault(v, []).append(u)
                    print('   %s %d:' % (nm, len(st_)), '; '.join('v=%g u=%g..%g(%d)' % (v, min(u), max(u), len(u)) for v, u in sorted(rows.items())))
                common = sorted(sx & sy)
                dz1 = {(a, b): c for a, b, c in x.tolist()}; dz0 = {(a, b): c for a, b, c in y.tolist()}
                print('   common samples %d, of which dz differs in %d' % (len(common), sum(dz1[c] != dz0[c] for c in common)))
            if bad >= 3:
                break
    print('forward part %s beside dense_align: %d of %d iterations differ' % (which, bad, k + 1))
