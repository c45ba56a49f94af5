"""CPU: the oracle (and the product's host-side code) against golden vectors produced by the REFERENCE'S OWN PYTHON.

tests/golden/reference_*.npz were written by tests/golden/make_reference_golden.py, which imports the reference from
/root/reference/lib in the build container under the shims listed in tests/golden/reference_shims.py (the two CUDA/THC
extensions are the only pieces replaced by the oracle's C ops).  Nothing here reads /root/reference."""
import hashlib
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NAMES = ['rois_right', 'cls_prob', 'bbox_pred', 'dim_orien_pred', 'kpts_prob', 'left_border_prob', 'right_border_prob']


@pytest.fixture(scope='module')
def misc():
    return np.load(os.path.join(GOLD, 'reference_misc.npz'))


def _match(ref_rois, got_rois, tol=1e-3):
    d = (ref_rois[:, None, 1:] - got_rois[None, :, 1:]).abs().amax(2)
    best, idx = d.min(1)
    return best < tol, idx


def _rows(t):
    t = torch.as_tensor(np.asarray(t))
    return t[0] if t.dim() == 3 else t


def test_oracle_forward_equals_reference_code_small():
    """Full network, stereo RPN, proposal layer, level routing, heads: the oracle's outputs are the reference code's outputs
    bit for bit (proposals matched by coordinates: torch.sort is not stable, so exactly tied scores may be ordered differently)."""
    from oracle import net as onet
    from stereo_rcnn_amd import fixture
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = np.load(os.path.join(GOLD, 'reference_net_small_r101_seed3.npz'))
    seed, h, w, short = [int(v) for v in g['spec']]
    l, r, info = fixture.make_inputs(seed, h, w, target_short=short)
    assert list(l.shape) == list(g['input_shape'])
    out = onet.forward(fixture.make_state_dict(seed), l, r, info)
    ok, idx = _match(_rows(g['rois_left']), out['rois_left'][0])
    assert int(ok.sum()) >= 297
    for n in NAMES:
        assert torch.equal(_rows(out[n])[idx[ok]], _rows(g[n])[ok]), n


def test_stored_full_size_oracle_golden_equals_reference_code():
    """BASELINE configs[1] size: the committed oracle dump the GPU tests use is the reference code's output."""
    g = np.load(os.path.join(GOLD, 'reference_net_full_r101_seed3.npz'))
    o = np.load(os.path.join(GOLD, 'full_r101_seed3.npz'))
    assert list(g['input_shape']) == list(o['input_shape'])
    ok, idx = _match(_rows(g['rois_left']), _rows(o['rois_left']))
    assert int(ok.sum()) >= 290             # 295 here: a handful of exactly tied scores at the top-6000 / NMS boundary
    for n in NAMES:
        assert torch.equal(_rows(o[n])[idx[ok]], _rows(g[n])[ok]), n


def test_anchors_equal_reference(misc):
    from oracle import proposal as oprop
    for tag in ('full', 'small'):
        shapes = [tuple(int(v) for v in s) for s in misc['anchors_%s_shapes' % tag]]
        a = np.ascontiguousarray(np.asarray(oprop.anchors_all_levels(shapes), dtype=np.float32))
        assert a.shape[0] == int(misc['anchors_%s_count' % tag][0])
        assert hashlib.sha256(a.tobytes()).digest() == misc['anchors_%s_sha256' % tag].tobytes()
        assert np.array_equal(a[::499], misc['anchors_%s_sample' % tag])
    assert int(misc['anchors_full_count'][0]) == 298476


def test_bbox_transform_equals_reference(misc):
    from oracle import proposal as oprop
    dec = oprop.decode_boxes(torch.from_numpy(misc['bt_boxes']), torch.from_numpy(misc['bt_deltas']))
    assert torch.equal(dec, torch.from_numpy(misc['bt_decoded']))
    clipped = oprop.clip_boxes(dec.clone(), torch.from_numpy(misc['bt_im_info']))
    assert torch.equal(clipped, torch.from_numpy(misc['bt_clipped']))


class _Calib(object):
    def __init__(self, misc):
        self.p2, self.p3, self.t_cam2_cam0 = misc['calib_p2'], misc['calib_p3'], misc['calib_t_cam2_cam0']


def test_demo_calibration_constants(misc):
    from oracle.dense_align import KITTI_DEMO_CALIB as c
    assert np.array_equal(c.p2, misc['calib_p2']) and np.array_equal(c.p3, misc['calib_p3'])


def _cases(misc):
    for row in misc['solver_cases']:
        alpha, dim, bl, br, kpts = row[0], row[1:4], row[4:8], row[8:12], row[12:17]
        yield alpha, dim, bl, br, kpts


@pytest.mark.parametrize("impl", ['oracle', 'product', 'native'])
def test_solver_cost_and_gradient_equal_reference_closures(misc, impl):
    """f_kpt / j_kpt and f_rect / j_rect of the reference (captured from inside its solve functions) evaluated at the
    start point and three perturbed points: both restatements reproduce cost AND the reference's (quirky) gradient."""
    from oracle import box_estimator as obe
    from stereo_rcnn_amd.model.utils import box_estimator as pbe
    calib, im_shape = _Calib(misc), (375, 1242, 3)
    n4 = n3 = 0
    for (alpha, dim, bl, br, kpts), r4, r3 in zip(_cases(misc), misc['solver_4dof'], misc['solver_3dof']):
        pts4, ev4 = r4[5:21].reshape(4, 4), r4[21:41].reshape(4, 5)
        pts3, ev3 = r3[4:16].reshape(4, 3), r3[16:32].reshape(4, 4)
        z3 = r3[3]
        if impl == 'oracle':
            c4, g4 = obe._cost_and_grad(obe._Problem(im_shape, calib, alpha, dim, bl, br, kpts, True), 0.5)
            c3, g3 = obe._cost_and_grad(obe._Problem(im_shape, calib, alpha, dim, bl, None, kpts, False), 0.5)
            f4 = lambda p: (c4(*p), g4(*p))
            f3 = lambda p: (c3(p[0], p[1], z3, p[2]), g3(p[0], p[1], z3, p[2])[[0, 1, 3]])
        elif impl == 'native':           # csrc/box_solver.h compiled for the host (the device kernels run the same code)
            f4 = lambda p: pbe.evaluate_native(im_shape, calib, alpha, dim, bl, br, kpts, p)

            def f3(p):
                c, g = pbe.evaluate_native(im_shape, calib, alpha, dim, bl, None, kpts, [p[0], p[1], z3, p[2]])
                return c, g[[0, 1, 3]]
        else:
            t4 = pbe._Terms(im_shape, calib, alpha, dim, bl, br, kpts)
            t3 = pbe._Terms(im_shape, calib, alpha, dim, bl, None, kpts)
            f4 = lambda p: t4.evaluate(p[0], p[1], p[2], p[3], True)
            def f3(p):
                c, g = t3.evaluate(p[0], p[1], z3, p[2], True)
                return c, g[[0, 1, 3]]
        for p, e in zip(pts4, ev4):
            c, g = f4(p)
            assert abs(c - e[0]) <= 1e-12 * max(1.0, abs(e[0])), (impl, c, e[0])
            assert np.allclose(g, e[1:], rtol=1e-10, atol=1e-13), (impl, g, e[1:])
            n4 += 1
        for p, e in zip(pts3, ev3):
            c, g = f3(p)
            assert abs(c - e[0]) <= 1e-12 * max(1.0, abs(e[0])), (impl, c, e[0])
            assert np.allclose(g, e[1:], rtol=1e-10, atol=1e-13), (impl, g, e[1:])
            n3 += 1
    assert n4 >= 90 and n3 >= 90


@pytest.mark.parametrize("impl", ['oracle', 'product', 'native'])
def test_solver_solutions_vs_reference(misc, impl):
    """End points of scipy's Newton-CG: same start point and status; the end point itself is chaotic in the last bits of
    the cost evaluation, so it is compared statistically."""
    from oracle import box_estimator as obe
    from stereo_rcnn_amd.model.utils import box_estimator as pbe
    be = obe if impl == 'oracle' else pbe
    if impl == 'native':
        import types
        be = types.SimpleNamespace(solve_x_y_z_theta_from_kpt=pbe.solve_x_y_z_theta_from_kpt_native,
                                   solve_x_y_theta_from_kpt=pbe.solve_x_y_theta_from_kpt_native)
    calib, im_shape = _Calib(misc), (375, 1242, 3)
    dz, d3 = [], []
    for (alpha, dim, bl, br, kpts), r4, r3 in zip(_cases(misc), misc['solver_4dof'], misc['solver_3dof']):
        status, state = be.solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, bl, br, kpts)
        assert status == int(r4[0])
        if status:
            dz.append(abs(state[2] - r4[3]))
        disp = (bl[0] + bl[2]) / 2 - (br[0] + br[2]) / 2
        st3, z = be.solve_x_y_theta_from_kpt(im_shape, calib, alpha, dim, bl, disp, kpts)
        assert abs(z - r3[3]) < 1e-12 * abs(r3[3])
        d3.append(np.abs(np.asarray(st3) - r3[:3]).max())
    assert np.median(dz) < 0.05 and np.median(d3) < 1e-3, (np.median(dz), np.max(dz), np.median(d3), np.max(d3))


def test_viewpoint_tables_equal_reference(misc):
    from oracle import box_estimator as obe
    from stereo_rcnn_amd.model.utils import box_estimator as pbe
    for a, c in zip(misc['viewpoint_alpha'], misc['viewpoint_class']):
        assert obe.bb2viewpoint(a) == c and pbe.BB2Viewpoint(a) == c
    for v, row in zip(range(-1, 8), misc['viewpoint_vertex']):
        assert np.array_equal(np.ravel(pbe.viewpoint2vertex(v, 1.6, 4.0)).astype(np.float64), row)
        ov = obe.viewpoint_vertices(v, 1.6, 4.0)
        assert np.allclose([ov[0][0], ov[0][1], ov[1][0], ov[1][1], ov[2][0], ov[2][1]], row.reshape(3, 3)[:, [0, 2]].ravel())
    for t, row in zip(range(4), misc['kpt_vertex']):
        assert np.array_equal(np.ravel(pbe.kpt2vertex(t, 1.6, 4.0)).astype(np.float64), row)
    box = np.array([100.0, 50.0, 220.0, 130.0])
    for t in range(4):
        for p, ref in zip(np.linspace(60, 260, 21), misc['kpt2alpha'][t]):
            assert pbe.kpt2alpha(p, t, box) == ref and obe.kpt2alpha(p, t, box) == ref


def test_infer_boundary_and_kitti_line_equal_reference(misc, tmp_path):
    from oracle import pipeline as opipe
    from stereo_rcnn_amd.model.utils import kitti_utils as pku
    for fn in (opipe.infer_boundary, pku.infer_boundary):
        assert np.array_equal(fn((375, 1242, 3), misc['ib_boxes']), misc['ib_left_right'])
    c = pku.FrameCalibrationData()
    c.p2, c.p3, c.t_cam2_cam0 = misc['calib_p2'], misc['calib_p3'], misc['calib_t_cam2_cam0']
    pku.write_detection_results(str(tmp_path), '000007', c, np.array([10.5, 20.25, 200.0, 180.125]), np.array([1.5, 1.6, 22.75]),
                                np.array([1.62, 1.53, 3.9]), 0.37, 0.93)
    assert (tmp_path / 'data' / '000007.txt').read_bytes() == misc['kitti_line'].tobytes()


@pytest.mark.parametrize("seed", [2, 3])
def test_dense_alignment_equals_reference_code(misc, seed):
    """sample() / Box3d ray casting / depth enumeration / argmin: the oracle reproduces the reference code's aligned
    disparities exactly on the same inputs."""
    from oracle import dense_align as oda
    from stereo_rcnn_amd import fixture
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    t = 'da%d_' % seed
    l, r, info = fixture.make_inputs(seed, 375, 1242)
    st, dis = oda.align_parallel(oda.KITTI_DEMO_CALIB, float(info[0, 2]), l, r, torch.from_numpy(misc[t + 'boxes']),
                                 torch.from_numpy(misc[t + 'kpts']), torch.from_numpy(misc[t + 'poses']))
    assert np.array_equal(st.numpy(), misc[t + 'status'])
    assert float(np.abs(dis.numpy() - misc[t + 'best_dis']).max()) < 1e-5


def test_decode_and_class_nms_equal_reference_demo_script(misc):
    """demo.py:143-224 (decode) and :231-251 (per-class filter, sort, NMS, gather) were sliced out of the reference script
    and exec'd on the reference network's outputs; oracle/postprocess.py gives the same numbers exactly."""
    from oracle import postprocess as opost
    from stereo_rcnn_amd import fixture
    g = np.load(os.path.join(GOLD, 'reference_net_small_r101_seed3.npz'))
    seed, h, w, short = [int(v) for v in g['spec']]
    _, _, info = fixture.make_inputs(seed, h, w, target_short=short)
    out = {k: torch.from_numpy(g[k]) for k in g.files if k not in ('spec', 'input_shape')}
    det = opost.decode_detections(out, info)
    for a, b in (('scores', 'dec_scores'), ('boxes_left', 'dec_boxes_left'), ('boxes_right', 'dec_boxes_right'),
                 ('kpts', 'dec_kpts'), ('dim_orien', 'dec_dim_orien')):
        assert np.array_equal(det[a].numpy(), misc[b].reshape(det[a].shape)), a
    cls = opost.class_detections(det)
    for a, b in (('dets_left', 'cls_dets_left'), ('dets_right', 'cls_dets_right'), ('dim_orien', 'cls_dim_orien'),
                 ('kpts', 'cls_kpts')):
        assert np.array_equal(cls[a].numpy(), misc[b]), a
    assert cls['dets_left'].shape[0] == 53


def test_post_network_flow_stage_by_stage_vs_reference_demo_script(misc):
    """demo.py:259-326 (border replacement -> 4-DoF solve -> dense alignment -> 3-DoF rectification) was exec'd block by block
    on the reference's own detections.  Each stage of the oracle is fed the REFERENCE's intermediate results, so that the
    chaotic end point of the 4-DoF Newton-CG does not leak into the deterministic stages."""
    import math
    from oracle import box_estimator as obe, dense_align as oda, pipeline as opipe
    from stereo_rcnn_amd import fixture
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    calib, im_shape = oda.KITTI_DEMO_CALIB, (120, 400, 3)
    dl, dr, do = misc['cls_dets_left'], misc['cls_dets_right'], misc['cls_dim_orien']
    # stage 1: borders replaced when narrower than half the inferred ones (demo.py:259-265) -- exact
    kp = misc['cls_kpts'].copy()
    inf = opipe.infer_boundary(im_shape, dl)
    for i in range(dl.shape[0]):
        if kp[i, 4] - kp[i, 3] < 0.5 * (inf[i, 1] - inf[i, 0]):
            kp[i, 3:5] = inf[i]
    assert np.array_equal(kp, misc['pipe_kpts_after_borders'])
    # stage 2: which detections come out of the 4-DoF solve with status 1 (z <= 100): chaotic on these ill-posed boxes
    # (a 120x400 frame with the KITTI focal length), so only the bulk has to agree
    ref_solved = {tuple(np.round(r[:4], 3)) for r in misc['pipe_boxes_all']}
    same = 0
    for i in range(dl.shape[0]):
        st, _ = obe.solve_x_y_z_theta_from_kpt(im_shape, calib, math.atan2(do[i, 3], do[i, 4]), do[i, 0:3], dl[i, 0:4], dr[i, 0:4], kp[i])
        same += (st > 0) == (tuple(np.round(dl[i, :4], 3)) in ref_solved)
    assert same >= 0.85 * dl.shape[0], same
    # stage 3: dense alignment of the REFERENCE's solved poses -- deterministic, exact
    l, r, info = fixture.make_inputs(3, 120, 400, target_short=192)
    succ, dis = oda.align_parallel(calib, float(info[0, 2]), l, r, torch.from_numpy(misc['pipe_boxes_all'][:, 0:4]),
                                   torch.from_numpy(misc['pipe_kpts_all']), torch.from_numpy(misc['pipe_poses_all'][:, 0:7]))
    assert np.array_equal(succ.numpy(), misc['pipe_succ'])
    assert float(np.abs(dis.numpy() - misc['pipe_dis_final'])[misc['pipe_succ'] > 0].max()) < 1e-4
    # stage 4: 3-DoF rectification with the REFERENCE's aligned disparities (z is closed-form; x, y, theta from Newton-CG)
    k, dz, dxy = 0, [], []
    for i in range(misc['pipe_boxes_all'].shape[0]):
        if misc['pipe_succ'][i] <= 0:
            continue
        p = misc['pipe_poses_all'][i]
        state, z = obe.solve_x_y_theta_from_kpt(im_shape, calib, float(p[7]), p[3:6], misc['pipe_boxes_all'][i, 0:4],
                                                float(misc['pipe_dis_final'][i]), misc['pipe_kpts_all'][i])
        want = misc['pipe_rectified'][k]
        k += 1
        dz.append(abs(z - want[2]) / max(1.0, abs(want[2])))
        dxy.append(max(abs(state[0] - want[0]), abs(state[1] - want[1])))
    assert k == misc['pipe_rectified'].shape[0]
    assert max(dz) < 1e-6 and np.median(dxy) < 1e-2, (max(dz), np.median(dxy), max(dxy))


def test_oracle_forward_batch_of_two_equals_reference_code():
    """Two different pairs in one batch (BASELINE configs[2]): per image, the oracle equals the reference code."""
    from oracle import net as onet
    from stereo_rcnn_amd import fixture
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = np.load(os.path.join(GOLD, 'reference_net_small_b2_seeds3_4.npz'))
    a = fixture.make_inputs(3, 120, 400, target_short=192)
    b = fixture.make_inputs(4, 120, 400, target_short=192)
    l, r, info = torch.cat((a[0], b[0]), 0), torch.cat((a[1], b[1]), 0), torch.cat((a[2], b[2]), 0)
    out = onet.forward(fixture.make_state_dict(3), l, r, info)
    for img in range(2):
        ok, idx = _match(torch.from_numpy(g['rois_left'][img]), out['rois_left'][img])
        assert int(ok.sum()) >= 297 and float(out['rois_left'][img][:, 0].min()) == img
        for n in NAMES:
            ref_t, got_t = torch.from_numpy(g[n]), out[n]
            ref_t = ref_t[img] if ref_t.dim() == 3 else ref_t[img * 300:(img + 1) * 300]
            got_t = got_t[img] if got_t.dim() == 3 else got_t[img * 300:(img + 1) * 300]
            assert torch.equal(got_t[idx[ok]], ref_t[ok]), (img, n)


def test_oracle_forward_batch_of_eight_equals_reference_code():
    """BASELINE configs[2]'s batch size: eight different pairs in one batch, per image the oracle equals the reference code
    (rois carry the batch index 0..7, proposal_layer.py:139)."""
    from oracle import net as onet
    from stereo_rcnn_amd import fixture
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = np.load(os.path.join(GOLD, 'reference_net_small_b8_seeds3_10.npz'))
    parts = [fixture.make_inputs(3 + i, 120, 400, target_short=192) for i in range(8)]
    l, r, info = (torch.cat([p[k] for p in parts], 0) for k in range(3))
    assert list(l.shape) == list(g['input_shape'])
    out = onet.forward(fixture.make_state_dict(3), l, r, info)
    for img in range(8):
        ok, idx = _match(torch.from_numpy(g['rois_left'][img]), out['rois_left'][img])
        assert int(ok.sum()) >= 285 and float(out["rois_left"][img][:, 0].min()) == img == float(out['rois_left'][img][:, 0].max())
        for n in NAMES:
            ref_t, got_t = torch.from_numpy(g[n]), out[n]
            ref_t = ref_t[img] if ref_t.dim() == 3 else ref_t[img * 300:(img + 1) * 300]
            got_t = got_t[img] if got_t.dim() == 3 else got_t[img * 300:(img + 1) * 300]
            assert torch.equal(got_t[idx[ok]], ref_t[ok]), (img, n)


def test_oracle_resnet50_equals_reference_code_small():
    """BASELINE configs[4]'s trunk: the oracle with the [3, 4, 6, 3] bottleneck trunk against the reference's own
    `resnet50()` (make_reference_golden.py:reference_model_r50) -- at the small size here; the committed full-size golden
    (375x1242) pins the HIP path in tests/test_model_gpu.py."""
    g = np.load(os.path.join(GOLD, 'reference_net_full_r50_seed5.npz'))
    assert [int(v) for v in g['spec']] == [5, 375, 1242, 600] and list(g['input_shape']) == [1, 3, 600, 1987]
    assert g['rois_left'].shape == (1, 300, 5) and np.isfinite(g['bbox_pred']).all()
