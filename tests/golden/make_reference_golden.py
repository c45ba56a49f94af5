"""Golden vectors produced by the REFERENCE'S OWN PYTHON, executed in this container.

    python tests/golden/make_reference_golden.py            # needs /root/reference (read-only); writes tests/golden/reference_*.npz

The reference (py2 / torch-0.3 era) is imported from /root/reference/lib under the shims of reference_shims.py
(listed there with what each one does and does not change).  These files are what pins the CPU oracle
(tests/test_reference_golden.py) and, through it or directly, the HIP path (tests/test_*_gpu.py): nothing at test
time reads /root/reference.  Inputs are regenerated from seeds by stereo_rcnn_amd/fixture.py; only outputs are stored.
"""
import hashlib
import math
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
warnings.filterwarnings('ignore')

import numpy as np   # noqa: E402
import torch         # noqa: E402

torch.set_num_threads(min(os.cpu_count() or 1, 16))
from oracle import ops as oracle_ops          # noqa: E402
import reference_shims                        # noqa: E402

reference_shims.install(oracle_ops)
from stereo_rcnn_amd import fixture           # noqa: E402

NAMES = ['rois_left', 'rois_right', 'cls_prob', 'bbox_pred', 'dim_orien_pred', 'kpts_prob', 'left_border_prob',
         'right_border_prob']


def reference_model(seed, layers=101, sd=None):
    from model.stereo_rcnn.resnet import resnet
    net = resnet(('__background__', 'Car'), layers, pretrained=False)
    net.create_architecture()
    net.load_state_dict(fixture.make_state_dict(seed) if sd is None else sd)
    net.eval()
    return net


def net_golden(net, seed, h, w, short, tag):
    l, r, info = fixture.make_inputs(seed, h, w, target_short=short)
    z, nb = torch.zeros(1, 1, 5), torch.zeros(1)
    with torch.no_grad():
        out = net(l, r, info, z, z, z, z, z, nb)      # the reference's 9-argument eval call (demo.py:137-140)
    d = {n: out[i].detach().numpy().astype(np.float32) for i, n in enumerate(NAMES)}
    d['input_shape'] = np.asarray(l.shape)
    d['spec'] = np.asarray([seed, h, w, short])
    np.savez_compressed(os.path.join(HERE, 'reference_net_%s.npz' % tag), **d)
    print('reference_net_%s.npz' % tag, {k: v.shape for k, v in d.items()})


def net_golden_cv_input_with_rpn(net, seed, h, w, tag, top=7000):
    """A frame whose network input comes from the OpenCV-resize restatement (oracle/preprocess.py) -- the input the product's own
    preprocessing kernel produces bit for bit -- so that the ragged tails are the ones a real KITTI frame of this size has
    (370x1224 -> 600x1985; fixture.make_inputs' truncating resize gives 1984).  Besides the network outputs the golden keeps what
    the reference's proposal layer was FED (captured by wrapping _ProposalLayer.forward): every anchor's foreground score and the
    deltas of the `top` best anchors (only the 6000 best can reach the NMS) -- the data of the tie audit in
    tests/test_model_gpu.py: every discrete decision (top-6000 membership, order, IoU > 0.7) that differs between the HIP run and
    the reference run must be a near-tie of the reference's own margins."""
    import hashlib
    from oracle import preprocess as opre
    lu, ru = fixture.synthetic_pair(seed, h, w)
    tl, s = opre.prepare_image(lu)
    tr, _ = opre.prepare_image(ru)
    l, r = torch.from_numpy(tl), torch.from_numpy(tr)
    info = torch.tensor([[l.shape[2], l.shape[3], s]], dtype=torch.float32)
    cap = {}
    layer = net.RCNN_rpn.RPN_proposal
    orig = layer.forward

    real_sort = torch.sort

    def recording_sort(*a, **k):                  # proposal_layer.py:96 -- torch.sort(scores, 1, True): NOT stable; its tie order is
        out = real_sort(*a, **k)                  # whatever this torch build does, so the golden keeps the order it actually used
        cap.setdefault('sorts', []).append(out[1].clone())
        return out

    def recording(inp):
        cap['probs'], cap['deltas'], cap['shapes'] = inp[0].clone(), inp[1].clone(), [list(map(int, x)) for x in inp[4]]
        torch.sort = recording_sort
        try:
            return orig(inp)
        finally:
            torch.sort = real_sort
    layer.forward = recording
    try:
        d = net_outputs(net, l, r, info)
    finally:
        layer.forward = orig
    fg = cap['probs'][0, :, 1].numpy().astype(np.float32)
    idx = np.argsort(-fg, kind='stable')[:top].astype(np.int32)
    orders = [o for o in cap.get('sorts', []) if o.numel() == fg.shape[0]]
    assert len(orders) == 1, [tuple(o.shape) for o in cap.get('sorts', [])]
    d['rpn_order'] = orders[0].view(-1)[:6000].numpy().astype(np.int32)          # the 6000 anchors the reference's NMS saw, in its order
    d.update({'input_shape': np.asarray(l.shape), 'spec': np.asarray([seed, h, w, 600]), 'im_info': info.numpy(),
              'input_sha256': np.frombuffer(hashlib.sha256(np.ascontiguousarray(tl).tobytes()).digest(), np.uint8),
              'rpn_fg': fg, 'rpn_top_idx': idx, 'rpn_top_deltas': cap['deltas'][0].numpy()[idx].astype(np.float32),
              'rpn_shapes': np.asarray(cap['shapes'], np.int32)})
    np.savez_compressed(os.path.join(HERE, 'reference_net_%s.npz' % tag), **d)
    print('reference_net_%s.npz' % tag, {k: v.shape for k, v in d.items()})


def net_outputs(net, l, r, info):
    z, nb = torch.zeros(1, 1, 5), torch.zeros(1)
    with torch.no_grad():
        out = net(l, r, info, z, z, z, z, z, nb)      # the reference's 9-argument eval call (demo.py:137-140)
    return {n: out[i].detach().numpy().astype(np.float32) for i, n in enumerate(NAMES)}


def net_golden_batch2(net):
    """BASELINE configs[2] form of the forward: two different pairs in one batch (rois carry the batch index)."""
    a = fixture.make_inputs(3, 120, 400, target_short=192)
    b = fixture.make_inputs(4, 120, 400, target_short=192)
    l, r, info = torch.cat((a[0], b[0]), 0), torch.cat((a[1], b[1]), 0), torch.cat((a[2], b[2]), 0)
    z, nb = torch.zeros(2, 1, 5), torch.zeros(2)
    with torch.no_grad():
        out = net(l, r, info, z, z, z, z, z, nb)
    d = {n: out[i].detach().numpy().astype(np.float32) for i, n in enumerate(NAMES)}
    d['input_shape'] = np.asarray(l.shape)
    np.savez_compressed(os.path.join(HERE, 'reference_net_small_b2_seeds3_4.npz'), **d)
    print('reference_net_small_b2_seeds3_4.npz', {k: v.shape for k, v in d.items()})


def net_golden_batch8(net):
    """BASELINE configs[2] batch size: EIGHT different pairs in one batch (rois carry the batch index 0..7,
    proposal_layer.py:139; roi_align_kernel.cu:33,51 reads it), small frames so that the file stays small."""
    parts = [fixture.make_inputs(3 + i, 120, 400, target_short=192) for i in range(8)]
    l, r, info = (torch.cat([p[k] for p in parts], 0) for k in range(3))
    z, nb = torch.zeros(8, 1, 5), torch.zeros(8)
    with torch.no_grad():
        out = net(l, r, info, z, z, z, z, z, nb)
    d = {n: out[i].detach().numpy().astype(np.float32) for i, n in enumerate(NAMES)}
    d['input_shape'] = np.asarray(l.shape)
    np.savez_compressed(os.path.join(HERE, 'reference_net_small_b8_seeds3_10.npz'), **d)
    print('reference_net_small_b8_seeds3_10.npz', {k: v.shape for k, v in d.items()})


def net_golden_full_b8(net):
    """BASELINE configs[2] AT THE SHAPE bench.py --config 2 runs: eight different 375x1242 pairs (seeds 3..10, bench.make_batch)
    in ONE forward at network input 600x1987 (B = 8 was pinned at 192x640 only)."""
    parts = [fixture.make_inputs(3 + i, 375, 1242) for i in range(8)]
    l, r, info = (torch.cat([p[k] for p in parts], 0) for k in range(3))
    z, nb = torch.zeros(8, 1, 5), torch.zeros(8)
    with torch.no_grad():
        out = net(l, r, info, z, z, z, z, z, nb)
    d = {n: out[i].detach().numpy().astype(np.float32) for i, n in enumerate(NAMES)}
    d['input_shape'] = np.asarray(l.shape)
    np.savez_compressed(os.path.join(HERE, 'reference_net_full_b8_seeds3_10.npz'), **d)
    print('reference_net_full_b8_seeds3_10.npz', {k: v.shape for k, v in d.items()})


def net_golden_r50_2x_b4(net):
    """BASELINE configs[4] AT THE SHAPE bench.py --config 4 runs: ResNet-50, four different 750x2484 pairs (seeds 5..8) in one
    forward at network input 1200x3974 (bench.make_batch: fixture.synthetic_pair + fixture.preprocess(short side 1200))."""
    parts = []
    for b in range(4):
        lu, ru = fixture.synthetic_pair(5 + b, 750, 2484)
        tl, sc = fixture.preprocess(lu, 1200, max_size=1 << 30)
        tr, _ = fixture.preprocess(ru, 1200, max_size=1 << 30)
        parts.append((tl, tr, torch.tensor([[tl.shape[2], tl.shape[3], sc]], dtype=torch.float32)))
    l, r, info = (torch.cat([p[k] for p in parts], 0) for k in range(3))
    z, nb = torch.zeros(4, 1, 5), torch.zeros(4)
    with torch.no_grad():
        out = net(l, r, info, z, z, z, z, z, nb)
    d = {n: out[i].detach().numpy().astype(np.float32) for i, n in enumerate(NAMES)}
    d['input_shape'] = np.asarray(l.shape)
    np.savez_compressed(os.path.join(HERE, 'reference_net_r50_2x_b4_seeds5_8.npz'), **d)
    print('reference_net_r50_2x_b4_seeds5_8.npz', {k: v.shape for k, v in d.items()})


def reference_model_r50(seed):
    """BASELINE configs[4] names a ResNet-50 trunk.  The reference ships `resnet50()` (resnet.py:188-196) next to the
    `resnet101()` its `_init_modules` hard-codes (resnet.py:229): for this golden the module-level name `resnet101` is
    pointed at the reference's own `resnet50` while the model is built -- every layer is still the reference's code."""
    import model.stereo_rcnn.resnet as ref_resnet
    saved = ref_resnet.resnet101
    ref_resnet.resnet101 = ref_resnet.resnet50
    try:
        net = ref_resnet.resnet(('__background__', 'Car'), 50, pretrained=False)
        net.create_architecture()
    finally:
        ref_resnet.resnet101 = saved
    net.load_state_dict(fixture.make_state_dict(seed, layers=fixture.R50))
    net.eval()
    return net


def anchors_golden():
    from generate_anchors import generate_anchors_all_pyramids
    from model.utils.config import cfg
    out = {}
    for tag, (h, w) in (('full', (600, 1987)), ('small', (192, 640))):
        shapes = []
        hh, ww = h, w
        # feature map sizes exactly as the network produces them: conv 7x7/2 pad 3, maxpool 3/2 ceil, then /2 three times
        hh, ww = (hh + 2 * 3 - 7) // 2 + 1, (ww + 2 * 3 - 7) // 2 + 1
        hh, ww = math.ceil((hh - 3) / 2) + 1, math.ceil((ww - 3) / 2) + 1
        shapes.append((hh, ww))
        for _ in range(3):
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
            shapes.append((hh, ww))
        shapes.append(((hh - 1) // 2 + 1, (ww - 1) // 2 + 1))          # P6: max_pool2d(p5, 1, stride=2)
        a = generate_anchors_all_pyramids(cfg.FPN_ANCHOR_SCALES, cfg.ANCHOR_RATIOS, np.asarray(shapes),
                                          cfg.FPN_FEAT_STRIDES, cfg.FPN_ANCHOR_STRIDE).astype(np.float32)
        out['anchors_%s_shapes' % tag] = np.asarray(shapes)
        out['anchors_%s_count' % tag] = np.asarray([a.shape[0]])
        out['anchors_%s_sha256' % tag] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
        out['anchors_%s_sample' % tag] = a[::499].copy()
    return out


def bbox_transform_golden():
    from bbox_transform import bbox_transform_inv, clip_boxes
    g = torch.Generator().manual_seed(5)
    boxes = torch.rand(1, 200, 4, generator=g) * 300
    boxes[:, :, 2:] += boxes[:, :, :2] + 5
    deltas = torch.randn(1, 200, 4, generator=g) * 0.3
    im_info = torch.tensor([[192.0, 640.0, 1.6]])
    dec = bbox_transform_inv(boxes, deltas, 1)
    clipped = clip_boxes(dec.clone(), im_info, 1)
    return {'bt_boxes': boxes.numpy(), 'bt_deltas': deltas.numpy(), 'bt_im_info': im_info.numpy(),
            'bt_decoded': dec.numpy(), 'bt_clipped': clipped.numpy()}


# ---------------------------------------------------------------------------- host-side geometry (A13, A14, A17)
def synthetic_cases(n=24, seed=7):
    """Observations of random 3-D boxes projected with the KITTI demo calibration (same generator as the tests):
    (alpha, dim(w,h,l), box_left, box_right, kpts(5)) per case, float64."""
    from oracle import box_estimator as obe
    from oracle.dense_align import KITTI_DEMO_CALIB as calib
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        z = rng.uniform(6, 45)
        x = rng.uniform(-0.45, 0.45) * z
        th = rng.uniform(-math.pi, math.pi)
        dim = (rng.uniform(1.5, 1.8), rng.uniform(1.4, 1.7), rng.uniform(3.5, 4.6))
        bl, br, corners = obe.project_observations(calib, (x, 1.65, z, th), dim)
        f, cx = calib.p2[0, 0], calib.p2[0, 2]
        kt = int(rng.integers(0, 4))
        sx, sz = obe._KPT_VERTS[kt]
        X, Z = corners[(sx, sz)]
        kp = f * X / Z + cx + rng.normal(0, 0.5)
        if not (bl[0] + 1 < kp < bl[2] - 1) or bl[2] - bl[0] < 12 or bl[3] - bl[1] < 12:
            continue
        alpha = th - math.pi / 2 + math.atan2(-x, z) + rng.normal(0, 0.05)
        noise = rng.normal(0, 0.4, 8)
        bl = [bl[i] + noise[i] for i in range(4)]
        br = [br[0] + noise[4], bl[1], br[2] + noise[5], bl[3]]
        cases.append((alpha, dim, bl, br, [kp, kt, 0.9, bl[0], bl[2]]))
    return cases


def solver_golden():
    """The reference's OWN cost / gradient closures (captured by intercepting scipy.optimize.minimize inside its module),
    evaluated at the start point and at perturbed points, and its solutions."""
    import box_estimator as rbe                 # the reference module (lib/model/utils on sys.path)
    import kitti_utils as rku
    import scipy.optimize
    calib = rku.read_obj_calibration('/root/reference/demo/calib.txt')
    captured = {}

    def recording_minimize(fun, x0, *a, **k):
        captured['fun'], captured['jac'], captured['x0'] = fun, k.get('jac'), np.array(x0, dtype=np.float64)
        return scipy.optimize.minimize(fun, x0, *a, **k)
    rbe.minimize = recording_minimize

    class _ScipyCompat(object):                # `scipy.array` (a numpy alias) was removed from scipy; same function
        array = staticmethod(np.array)
        optimize = scipy.optimize
    rbe.scipy = _ScipyCompat()
    rng = np.random.default_rng(11)

    def closure_rows(im_shape, cases):
        rows4, rows3, cases_out = [], [], []
        for alpha, dim, bl, br, kpts in cases:
            dimv = np.array(dim)
            captured.clear()
            status, state = rbe.solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dimv, np.array(bl), np.array(br), np.array(kpts))
            if 'x0' not in captured:            # early-out (:186-187): nothing to evaluate
                continue
            x0 = captured['x0']
            pts = [x0] + [x0 + rng.normal(0, [0.3, 0.1, 0.8, 0.1]) for _ in range(3)]
            ev = [[float(captured['fun'](p))] + list(np.asarray(captured['jac'](p), dtype=np.float64)) for p in pts]
            rows4.append(np.concatenate([[status], np.asarray(state, dtype=np.float64).ravel()[:4], np.concatenate(pts), np.asarray(ev).ravel()]))
            disp = (bl[0] + bl[2]) / 2 - (br[0] + br[2]) / 2
            st3, z = rbe.solve_x_y_theta_from_kpt(im_shape, calib, alpha, dimv, np.array(bl), disp, np.array(kpts))
            x0 = captured['x0']
            pts = [x0] + [x0 + rng.normal(0, [0.3, 0.1, 0.1]) for _ in range(3)]
            ev = [[float(captured['fun'](p))] + list(np.asarray(captured['jac'](p), dtype=np.float64)) for p in pts]
            rows3.append(np.concatenate([np.asarray(st3, dtype=np.float64), [z], np.concatenate(pts), np.asarray(ev).ravel()]))
            cases_out.append(np.concatenate([[alpha], dim, bl, br, kpts]))
        return np.asarray(cases_out), np.asarray(rows4), np.asarray(rows3)

    cases_out, rows4, rows3 = closure_rows((375, 1242, 3), synthetic_cases())
    # the per-class detections of the small network run on a 120x400 frame: boxes hugging every image border, i.e. the
    # truncation branches (alpha residual, right-box residuals, dropped top/bottom/left/right terms)
    d = decode_golden()
    det_cases = []
    for i in range(d['cls_dets_left'].shape[0]):
        do = d['cls_dim_orien'][i].astype(np.float64)
        det_cases.append((math.atan2(do[3], do[4]), do[0:3], d['cls_dets_left'][i, 0:4].astype(np.float64),
                          d['cls_dets_right'][i, 0:4].astype(np.float64), d['cls_kpts'][i].astype(np.float64)))
    tc, t4, t3 = closure_rows((120, 400, 3), det_cases)
    out = {'solver_cases': cases_out, 'solver_4dof': rows4, 'solver_3dof': rows3,
           'solver_trunc_cases': tc, 'solver_trunc_4dof': t4, 'solver_trunc_3dof': t3,
           'calib_p2': calib.p2, 'calib_p3': calib.p3, 'calib_t_cam2_cam0': calib.t_cam2_cam0}
    # discrete helpers: viewpoint classification / vertex tables / keypoint -> alpha
    al = np.linspace(-7.0, 7.0, 561)
    out['viewpoint_alpha'] = al
    out['viewpoint_class'] = np.asarray([rbe.BB2Viewpoint(a) for a in al])
    out['viewpoint_vertex'] = np.asarray([np.ravel(rbe.viewpoint2vertex(v, 1.6, 4.0)) for v in range(-1, 8)], dtype=np.float64)
    out['kpt_vertex'] = np.asarray([np.ravel(rbe.kpt2vertex(t, 1.6, 4.0)) for t in range(4)], dtype=np.float64)
    box = np.array([100.0, 50.0, 220.0, 130.0])
    out['kpt2alpha'] = np.asarray([[rbe.kpt2alpha(p, t, box) for p in np.linspace(60, 260, 21)] for t in range(4)])
    # infer_boundary on overlapping boxes; the KITTI result line
    g = np.random.default_rng(3)
    b = np.zeros((12, 4), np.float32)
    b[:, 0] = g.uniform(0, 900, 12); b[:, 2] = b[:, 0] + g.uniform(30, 300, 12)
    b[:, 1] = g.uniform(100, 200, 12); b[:, 3] = b[:, 1] + g.uniform(30, 170, 12)
    b[:, 2] = np.minimum(b[:, 2], 1241)
    out['ib_boxes'] = b
    out['ib_left_right'] = rku.infer_boundary((375, 1242, 3), b)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        rku.write_detection_results(td, '000007', calib, np.array([10.5, 20.25, 200.0, 180.125]), np.array([1.5, 1.6, 22.75]),
                                    np.array([1.62, 1.53, 3.9]), 0.37, 0.93)
        out['kitti_line'] = np.frombuffer(open(td + '/data/000007.txt', 'rb').read(), np.uint8)
    return out


# ---------------------------------------------------------------------------- dense alignment (A15, A16)
def dense_align_golden():
    from model.dense_align.dense_align import align_parallel
    import kitti_utils as rku
    from oracle import dense_align as oda           # only its box-projection helper, to place boxes around the poses
    calib = rku.read_obj_calibration('/root/reference/demo/calib.txt')
    out = {}
    for seed, n in ((2, 6), (3, 12)):
        rng = np.random.default_rng(seed)
        poses = []
        for _ in range(n):
            z = rng.uniform(7, 45)
            x = rng.uniform(-0.6, 0.6) * z * 0.8
            poses.append([x, rng.uniform(1.4, 1.8), z, 1.6 * rng.uniform(0.9, 1.1), 1.5 * rng.uniform(0.9, 1.1),
                          4.0 * rng.uniform(0.9, 1.1), rng.uniform(-np.pi, np.pi)])
        poses = torch.tensor(poses, dtype=torch.float32)
        boxes = torch.tensor([oda.project_box(oda.KITTI_DEMO_CALIB, p) for p in poses], dtype=torch.float32)
        boxes[:, 0::2].clamp_(0, 1241)
        boxes[:, 1::2].clamp_(0, 374)
        kp = torch.zeros(n, 5)
        kp[:, 3] = boxes[:, 0] + torch.from_numpy(rng.uniform(0, 3, n).astype(np.float32))
        kp[:, 4] = boxes[:, 2] - torch.from_numpy(rng.uniform(0, 3, n).astype(np.float32))
        l, r, info = fixture.make_inputs(seed, 375, 1242)
        with torch.no_grad():
            status, dis = align_parallel(calib, float(info[0, 2]), l, r, boxes.clone(), kp.clone(), poses.clone())
        t = 'da%d_' % seed
        out.update({t + 'poses': poses.numpy(), t + 'boxes': boxes.numpy(), t + 'kpts': kp.numpy(),
                    t + 'status': status.numpy().astype(np.float32), t + 'best_dis': dis.numpy().astype(np.float32)})
    return out


# ---------------------------------------------------------------------------- detection decode + per-class NMS (A11, A12)
def _demo_slice(start_marker, end_marker):
    """Source lines of /root/reference/demo.py from the line containing start_marker to the one containing end_marker
    (inclusive), de-indented -- executed in memory, never written anywhere."""
    import textwrap
    lines = open('/root/reference/demo.py').read().split('\n')
    a = next(i for i, ln in enumerate(lines) if start_marker in ln)
    b = next(i for i, ln in enumerate(lines) if end_marker in ln and i > a)
    return textwrap.dedent('\n'.join(lines[a:b + 1]))


def decode_golden(g=None, info=None):
    """demo.py is a script, so its decode block (:143-224) and the per-class filter / sort / NMS block (:231-251) are
    sliced out of the file by content markers and exec'd on the reference network's own outputs (default: the small
    seeded case; `g` / `info`: another forward's outputs and its im_info)."""
    from bbox_transform import bbox_transform_inv, kpts_transform_inv, border_transform_inv, clip_boxes
    from model.utils.config import cfg
    from model.nms.nms_wrapper import nms
    if g is None:
        g = np.load(os.path.join(HERE, 'reference_net_small_r101_seed3.npz'))
        seed, h, w, short = [int(v) for v in g['spec']]
        _, _, info = fixture.make_inputs(seed, h, w, target_short=short)
    ns = {'torch': torch, 'np': np, 'cfg': cfg, 'kitti_classes': np.asarray(['__background__', 'Car']), 'xrange': range,
          'bbox_transform_inv': bbox_transform_inv, 'kpts_transform_inv': kpts_transform_inv,
          'border_transform_inv': border_transform_inv, 'clip_boxes': clip_boxes, 'nms': nms, 'eval_thresh': 0.05,
          'im_info': info, 'cls_prob': torch.from_numpy(g['cls_prob']), 'rois_left': torch.from_numpy(g['rois_left']),
          'rois_right': torch.from_numpy(g['rois_right']), 'bbox_pred': torch.from_numpy(g['bbox_pred']),
          'bbox_pred_dim': torch.from_numpy(g['dim_orien_pred']), 'kpts_prob': torch.from_numpy(g['kpts_prob']),
          'left_prob': torch.from_numpy(g['left_border_prob']), 'right_prob': torch.from_numpy(g['right_border_prob'])}
    exec(compile(_demo_slice('scores = cls_prob.data', 'dim_orien = dim_orien.squeeze()'), 'demo.py[decode]', 'exec'), ns)
    out = {'dec_scores': ns['scores'].numpy(), 'dec_boxes_left': ns['pred_boxes_left'].numpy(),
           'dec_boxes_right': ns['pred_boxes_right'].numpy(), 'dec_kpts': ns['pred_kpts'].numpy(),
           'dec_dim_orien': ns['dim_orien'].numpy()}
    exec(compile(_demo_slice('for j in xrange(1, len(kitti_classes)):', 'cls_kpts = cls_kpts[keep]'), 'demo.py[class loop]', 'exec'), ns)
    out.update({'cls_dets_left': ns['cls_dets_left'].numpy(), 'cls_dets_right': ns['cls_dets_right'].numpy(),
                'cls_dim_orien': ns['cls_dim_orien'].numpy(), 'cls_kpts': ns['cls_kpts'].numpy(),
                'cls_keep': ns['keep'].numpy().astype(np.int64)})
    return out


# ---------------------------------------------------------------------------- the post-network flow of demo.py (:259-326)
def pipeline_golden(d=None, inputs=None, hw=None):
    """Border replacement, 4-DoF solve, dense alignment and 3-DoF rectification exactly as demo.py strings them together:
    the three blocks are sliced out of the script and exec'd on the per-class detections of decode_golden() (default: the
    small seeded case; d / inputs=(l, r, info) / hw=(h, w) of the original image: another frame)."""
    import types
    import box_estimator as rbe
    import kitti_utils as rku
    from model.dense_align import dense_align as rda
    import math as m
    if d is None:
        d = decode_golden()
        g = np.load(os.path.join(HERE, 'reference_net_small_r101_seed3.npz'))
        seed, h, w, short = [int(v) for v in g['spec']]
        l, r, info = fixture.make_inputs(seed, h, w, target_short=short)
    else:
        (l, r, info), (h, w) = inputs, hw

    class _Scalar03(object):              # torch 0.3: indexing a tensor down to one element gave a Python number
        def __init__(self, t):
            self.t = t

        @property
        def data(self):
            return self

        def __getitem__(self, idx):
            v = self.t[idx]
            return float(v) if (not torch.is_tensor(v) or v.dim() == 0) else v

    class _Img(object):
        shape = (h, w, 3)
    solved = []
    vis = types.SimpleNamespace(vis_box_in_bev=lambda im, xyz, dim, theta, width=0: solved.append(
        np.concatenate([np.asarray(xyz, dtype=np.float64), [float(theta)]])) or im,
        vis_single_box_in_img=lambda im, calib, xyz, dim, theta: im)
    ns = {'torch': torch, 'np': np, 'm': m, 'time': __import__('time'), 'kitti_utils': rku, 'box_estimator': rbe,
          'dense_align': rda, 'vis_utils': vis, 'vis_detections': lambda im, *a, **k: im, 'kitti_classes': ['__background__', 'Car'],
          'j': 1, 'eval_thresh': 0.05, 'vis_thresh': -1.0, 'calib': rku.read_obj_calibration('/root/reference/demo/calib.txt'),
          'im2show_left': _Img(), 'im2show_right': _Img(), 'im_box': None, 'im_info': _Scalar03(info),
          'im_left_data': l, 'im_right_data': r,
          'cls_dets_left': torch.from_numpy(d['cls_dets_left']), 'cls_dets_right': torch.from_numpy(d['cls_dets_right']),
          'cls_dim_orien': torch.from_numpy(d['cls_dim_orien']), 'cls_kpts': torch.from_numpy(d['cls_kpts']).clone()}

    class _ScipyCompat(object):
        array = staticmethod(np.array)
    rbe.scipy = _ScipyCompat()
    import scipy.optimize
    rbe.minimize = scipy.optimize.minimize
    # torch 0.3 indexing rule for this block: a tensor indexed down to ONE element is a Python number, so that e.g. the
    # keypoint terms of the solver run in double as they did for the reference's authors (under torch 2.x they would be
    # float32 0-dim tensor arithmetic, and scipy's Newton-CG amplifies that 1e-7 difference to metres on ill-posed boxes)
    _getitem = torch.Tensor.__getitem__

    def getitem03(self, idx):
        v = _getitem(self, idx)
        return v.item() if v.dim() == 0 else v
    torch.Tensor.__getitem__ = getitem03
    try:
        return _pipeline_blocks(ns, solved)
    finally:
        torch.Tensor.__getitem__ = _getitem


def _pipeline_blocks(ns, solved):
    exec(compile(_demo_slice('infered_kpts = kitti_utils.infer_boundary(', 'cls_kpts[detect_idx,3:5] = infered_kpts[detect_idx]'),
                 'demo.py[borders]', 'exec'), ns)
    out = {'pipe_kpts_after_borders': ns['cls_kpts'].numpy().copy()}
    exec(compile(_demo_slice('# read intrinsic', 'poses_all = torch.cat((poses_all,poses.unsqueeze(0)),0)'), 'demo.py[solve]', 'exec'), ns)
    out.update({'pipe_boxes_all': ns['boxes_all'].numpy(), 'pipe_kpts_all': ns['kpts_all'].numpy(), 'pipe_poses_all': ns['poses_all'].numpy()})
    exec(compile(_demo_slice('if boxes_all.dim() > 0:', 'im2show_left = vis_utils.vis_single_box_in_img('), 'demo.py[align+rectify]', 'exec'), ns)
    out.update({'pipe_succ': ns['succ'].numpy().astype(np.float32), 'pipe_dis_final': ns['dis_final'].numpy().astype(np.float32),
                'pipe_rectified': np.asarray(solved, dtype=np.float64).reshape(-1, 4)})
    return out


# ---------------------------------------------------------------------------- BASELINE configs[0]: the demo pair
def demo_pair_golden(net):
    """demo.py:100-326 on the reference's own demo/left.png + right.png + calib.txt (a NATURAL image: other score / tie
    statistics, ROI level mix and dense-alignment cost landscape than the smooth synthetic fixtures), seeded weights
    fixture.demo_state_dict (the trained checkpoint is an external download; see there for the 1/8 objectness scale).  PNGs are decoded with PIL (scipy.misc.imread did the same through PIL)
    and the decoded uint8 arrays are committed (demo_pair_u8.npz) so that no test needs /root/reference or a decoder;
    preprocessing = demo.py:107-124 with cv2.resize replaced by its restatement oracle/preprocess.py (cv2 is absent)."""
    from PIL import Image
    from oracle import preprocess as opre
    left = np.asarray(Image.open('/root/reference/demo/left.png').convert('RGB'))
    right = np.asarray(Image.open('/root/reference/demo/right.png').convert('RGB'))
    assert left.shape == (375, 1242, 3) and left.dtype == np.uint8
    np.savez_compressed(os.path.join(HERE, 'demo_pair_u8.npz'), left=left, right=right,
                        calib=np.frombuffer(open('/root/reference/demo/calib.txt', 'rb').read(), np.uint8))
    tl, s = opre.prepare_image(left)
    tr, _ = opre.prepare_image(right)
    l, r = torch.from_numpy(tl), torch.from_numpy(tr)
    info = torch.tensor([[l.shape[2], l.shape[3], s]], dtype=torch.float32)        # demo.py:122-123
    d = net_outputs(net, l, r, info)
    d['input_shape'] = np.asarray(l.shape)
    d['input_sha256'] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(tl).tobytes()).digest(), np.uint8)
    dec = decode_golden(d, info)
    d.update(dec)
    d.update(pipeline_golden(dec, (l, r, info), (375, 1242)))
    np.savez_compressed(os.path.join(HERE, 'reference_demo_pair_r101_seed3.npz'), **d)
    print('reference_demo_pair_r101_seed3.npz', {k: v.shape for k, v in d.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['net', 'misc']
    if 'net' in which:
        net = reference_model(3)
        net_golden(net, 3, 120, 400, 192, 'small_r101_seed3')
        net_golden(net, 3, 375, 1242, 600, 'full_r101_seed3')
        net_golden_batch2(net)
    if 'b8' in which:
        net_golden_batch8(reference_model(3))
    if 'r50' in which:
        net_golden(reference_model_r50(5), 5, 375, 1242, 600, 'full_r50_seed5')
    if 'full_b8' in which:
        net_golden_full_b8(reference_model(3))
    if 'r50_2x' in which:
        net_golden_r50_2x_b4(reference_model_r50(5))
    if 'kitti370' in which:       # KITTI's other common frame size: 370x1224 -> 600x1985 (other ragged tails in every layer)
        net_golden(reference_model(3), 4, 370, 1224, 600, 'full_370x1224_r101_seed4')
    if 'kitti370cv' in which:     # ... the same frame size through the OpenCV-resize restatement (600x1985), with the RPN data of the tie audit
        net_golden_cv_input_with_rpn(reference_model(3), 4, 370, 1224, 'cv_370x1224_r101_seed4')
    if 'demo' in which:
        demo_pair_golden(reference_model(3, sd=fixture.demo_state_dict(3)))
    if 'misc' in which:
        d = {}
        d.update(anchors_golden())
        d.update(bbox_transform_golden())
        d.update(solver_golden())
        d.update(dense_align_golden())
        d.update(decode_golden())
        d.update(pipeline_golden())
        np.savez_compressed(os.path.join(HERE, 'reference_misc.npz'), **d)
        print('reference_misc.npz', sorted(d))
