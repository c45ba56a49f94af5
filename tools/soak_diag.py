import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle.dense_align import KITTI_DEMO_CALIB as calib
from stereo_rcnn_amd import fixture, pipeline, serving
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture(); m.load_state_dict(fixture.make_state_dict(3)); m.cuda().eval()
m.precision = 'f16x3'; m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
frame = (l, r, info, calib, (375, 1242, 3), float(info[0, 2]))
serving.load_shipped_plans()
solver = sys.argv[1] if len(sys.argv) > 1 else 'host'
lone = pipeline.detect_3d(m, *frame[:5], solver=solver)
list(pipeline.detect_3d_stream(m, [frame] * 6, slots=3, solver=solver))
bad = {}
for k, objs in enumerate(pipeline.detect_3d_stream(m, [frame] * 306, slots=3, solver=solver)):
    if len(objs) != len(lone):
        bad[k] = 'len %d vs %d' % (len(objs), len(lone)); continue
    for i, (a, b) in enumerate(zip(lone, objs)):
        for key, va in a.items():
            vb = b[key]
            if not (np.array_equal(va, vb) if isinstance(va, np.ndarray) else va == vb):
                bad.setdefault(k, []).append((i, key, va, vb))
print(len(bad), 'bad frames')
for k in list(bad)[:6]:
    print(k, bad[k] if isinstance(bad[k], str) else bad[k][:4])
