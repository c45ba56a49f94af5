"""dev tool: do forwards in flight on several streams give the outputs of a lone forward?  (bit-exact check per slot)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval(); m.precision = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
names = ['rois_left', 'rois_right', 'cls_prob', 'bbox_pred', 'dim_orien', 'kpts_prob', 'left', 'right']
with torch.no_grad():
    ref = [t.clone() for t in m(l, r, info)[:8]]
    torch.cuda.synchronize()
    again = [t.clone() for t in m(l, r, info)[:8]]
    torch.cuda.synchronize()
    print('lone forward repeatable:', all(torch.equal(a, b) for a, b in zip(ref, again)))
    for S in (2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(S)]
        for s in range(S):                     # first touch per slot, serially
            with torch.cuda.stream(streams[s]):
                m(l, r, info, slot=s)
            torch.cuda.synchronize()
        outs = []
        for k in range(6 * S):
            with torch.cuda.stream(streams[k % S]):
                outs.append([t.clone() for t in m(l, r, info, slot=k % S)[:8]])
        torch.cuda.synchronize()
        bad = {}
        for k, o in enumerate(outs):
            for n, a, b in zip(names, ref, o):
                if not torch.equal(a, b):
                    bad.setdefault(k % S, {}).setdefault(n, 0)
                    bad[k % S][n] += 1
        print('%d in flight: mismatching (slot -> tensor -> count of %d runs): %s' % (S, 6, bad if bad else 'none'))
        # stage probe: which intermediate differs first (compare plan buffers of slot s with slot 0 after a concurrent round)
        if bad:
            plans = [m._get_plan(1, l.shape[2], l.shape[3], s) for s in range(S)]
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    m(l, r, info, slot=k)
            torch.cuda.synchronize()
            with torch.cuda.stream(streams[0]):
                m(l, r, info, slot=0)
            torch.cuda.synchronize()
            p0 = plans[0]
            lone = {n: getattr(p0, n).clone() for n in ('c1', 'p5', 'p4', 'p3', 'p2', 'probs', 'deltas', 'sem', 'h1', 'h2', 'fc', 'kp_in', 'kp_up', 'kp_logits')}
            lone.update({'c%d' % i: p0.c[i].clone() for i in range(4)})
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    m(l, r, info, slot=k)
            torch.cuda.synchronize()
            for s in range(S):
                diffs = [n for n in lone if not torch.equal(lone[n], (plans[s].c[int(n[1])] if n in ('c0', 'c1_', 'c2', 'c3') and n != 'c1' else getattr(plans[s], n, None)) if n not in ('c0', 'c2', 'c3') else plans[s].c[int(n[1])])]
                print('   slot %d buffers differing from a lone slot-0 run: %s' % (s, diffs))

# ---- the device 3-D stage under concurrency: record of every frame vs the record of a lone run
import numpy as np
from stereo_rcnn_amd import pipeline
from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
from tools.demo_pipeline import demo_calib
calib = demo_calib()
shape = (375, 1242, 3)
with torch.no_grad():
    out = m(l, r, info)
    st = pipeline.launch_3d(out, l, r, info, float(info[0, 2]), calib, shape)
    st.event.synchronize()
    ref_rec, ref_state = st.rec_host.numpy().copy(), st.state_host.numpy().copy()
    k0 = int(ref_rec[0, 0])
    FR = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for S in (3, 3, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        pend = []
        nbad = 0
        for k in range(FR * S):
            s = k % S
            if len(pend) == S:
                h = pend.pop(0)
                h.event.synchronize()
                rec = h.rec_host.numpy()
                d = rec[:k0 + 1] != ref_rec[:k0 + 1]
                if d.any():
                    nbad += 1
                    torch.cuda.synchronize()
                    rows = np.nonzero(d.any(1))[0]
                    for rr in rows:
                        print('frame %d slot %d row %d: got  %s' % (k - S, (k - S) % S, rr, np.array2string(rec[rr, 20:32], precision=4)))
                        print('%s ref  %s' % (' ' * 24, np.array2string(ref_rec[rr, 20:32], precision=4)))
                    # the stage buffers still hold this frame's alignment inputs / outputs
                    i = int(rows[0]) - 1
                    print('   stage buffers: valid %s box %s borders %s pose %s -> status %s dis %s' % (
                        float(h.valid[i]), h.boxes[i].tolist(), h.borders[i].tolist(), h.poses[i].tolist(), float(h.align_status[i]), float(h.best_dis[i])))
                    kp = torch.zeros(h.boxes.shape[0], 5, device=dev); kp[:, 3:5] = h.borders
                    a2, b2 = align_parallel(calib, float(info[0, 2]), l, r, h.boxes, kp, h.poses, valid=h.valid)
                    torch.cuda.synchronize()
                    print('   dense alignment of those very buffers again: status %s dis %s' % (float(a2[i]), float(b2[i])))
            streams[s].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[s]):
                out = m(l, r, info, slot=s)
                pend.append(pipeline.launch_3d(out, l, r, info, float(info[0, 2]), calib, shape, slot=s))
        torch.cuda.synchronize()
        print('%d in flight: %d of %d frames differ' % (S, nbad, FR * S - S), flush=True)
