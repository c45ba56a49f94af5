"""BASELINE configs[0]: the reference's own demo pair (demo/left.png, right.png, calib.txt -- committed decoded as
tests/golden/demo_pair_u8.npz) through demo.py:100-326, executed by the REFERENCE'S OWN CODE in the build container
(tests/golden/make_reference_golden.py demo -> reference_demo_pair_r101_seed3.npz, weights fixture.demo_state_dict).

CPU part: the oracle reproduces it (network bit for bit, decode / class NMS / borders / dense alignment exactly).
GPU part (-m gpu): the HIP path from the uint8 images on (fused preprocessing -> forward -> decode -> class NMS -> borders ->
4-DoF solve -> dense alignment -> 3-DoF rectification), both conv engines, with the per-object 3-D box deltas printed."""
import hashlib
import io
import math
import os

import numpy as np
import pytest
import torch

import tolerances as tol_

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NAMES = ['rois_right', 'cls_prob', 'bbox_pred', 'dim_orien_pred', 'kpts_prob', 'left_border_prob', 'right_border_prob']


@pytest.fixture(scope='module')
def pair():
    return np.load(os.path.join(GOLD, 'demo_pair_u8.npz'))


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLD, 'reference_demo_pair_r101_seed3.npz'))


def demo_calib(pair, tmp_dir):
    """The demo calibration through the product's own reader (kitti_utils.read_obj_calibration)."""
    from stereo_rcnn_amd.model.utils import kitti_utils
    path = os.path.join(str(tmp_dir), 'calib.txt')
    with open(path, 'wb') as fh:
        fh.write(pair['calib'].tobytes())
    return kitti_utils.read_obj_calibration(path)


def _rows(t):
    t = torch.as_tensor(np.asarray(t))
    return t[0] if t.dim() == 3 else t


def _inputs(pair):
    from oracle import preprocess as opre
    tl, s = opre.prepare_image(pair['left'])
    tr, _ = opre.prepare_image(pair['right'])
    info = torch.tensor([[tl.shape[2], tl.shape[3], s]], dtype=torch.float32)
    return torch.from_numpy(tl), torch.from_numpy(tr), info


def test_fixture_and_preprocessing(pair, gold, tmp_path):
    assert pair['left'].shape == (375, 1242, 3) and pair['right'].shape == (375, 1242, 3)
    l, r, info = _inputs(pair)
    assert list(l.shape) == list(gold['input_shape']) == [1, 3, 600, 1987]
    assert hashlib.sha256(np.ascontiguousarray(l.numpy()).tobytes()).digest() == gold['input_sha256'].tobytes()
    c = demo_calib(pair, tmp_path)
    assert abs(c.p2[0, 0] - 721.5377) < 1e-9 and abs((c.p2[0, 3] - c.p3[0, 3]) / c.p2[0, 0] - 0.5327) < 1e-4


def test_oracle_forward_equals_reference_code_on_demo_pair(pair, gold):
    """Natural image, 600 x 1987: all 300 proposals and every head output equal to the reference code's, bit for bit."""
    from oracle import net as onet
    from stereo_rcnn_amd import fixture
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    l, r, info = _inputs(pair)
    out = onet.forward(fixture.demo_state_dict(3), l, r, info)
    d = (_rows(gold['rois_left'])[:, None, 1:] - out['rois_left'][0][None, :, 1:]).abs().amax(2)
    best, idx = d.min(1)
    ok = best < 1e-3
    assert int(ok.sum()) >= 297, int(ok.sum())
    for n in NAMES:
        assert torch.equal(_rows(out[n])[idx[ok]], _rows(gold[n])[ok]), n


def test_oracle_post_network_flow_on_demo_pair(pair, gold, tmp_path):
    """demo.py:143-326 on the reference network's outputs: decode, class NMS, borders and dense alignment EXACT;
    3-DoF depth exact; the 4-DoF end points (scipy Newton-CG on a non-gradient) reported."""
    from oracle import box_estimator as obe, dense_align as oda, pipeline as opipe, postprocess as opost
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    l, r, info = _inputs(pair)
    calib, im_shape = demo_calib(pair, tmp_path), (375, 1242, 3)
    out = {k: torch.from_numpy(gold[k]) for k in ['rois_left'] + NAMES}
    det = opost.decode_detections(out, info)
    for a, b in (('scores', 'dec_scores'), ('boxes_left', 'dec_boxes_left'), ('boxes_right', 'dec_boxes_right'),
                 ('kpts', 'dec_kpts'), ('dim_orien', 'dec_dim_orien')):
        assert np.array_equal(det[a].numpy(), gold[b].reshape(det[a].shape)), a
    cls = opost.class_detections(det)
    for a, b in (('dets_left', 'cls_dets_left'), ('dets_right', 'cls_dets_right'), ('dim_orien', 'cls_dim_orien'),
                 ('kpts', 'cls_kpts')):
        assert np.array_equal(cls[a].numpy(), gold[b]), a
    dl, dr, do = gold['cls_dets_left'], gold['cls_dets_right'], gold['cls_dim_orien']
    kp = gold['cls_kpts'].copy()
    inf = opipe.infer_boundary(im_shape, dl)
    for i in range(dl.shape[0]):
        if kp[i, 4] - kp[i, 3] < 0.5 * (inf[i, 1] - inf[i, 0]):
            kp[i, 3:5] = inf[i]
    assert np.array_equal(kp, gold['pipe_kpts_after_borders'])
    # 4-DoF: same objects solved, end points compared per object
    ref_boxes = gold['pipe_boxes_all'][:, 0:4]
    solved, d4 = [], []
    for i in range(dl.shape[0]):
        st, state = obe.solve_x_y_z_theta_from_kpt(im_shape, calib, math.atan2(do[i, 3], do[i, 4]), do[i, 0:3], dl[i, 0:4],
                                                   dr[i, 0:4], kp[i])
        if st > 0:
            solved.append(i)
            j = int(np.argmin(np.abs(ref_boxes - dl[i, 0:4]).max(1)))
            if np.abs(ref_boxes[j] - dl[i, 0:4]).max() < 1e-4:
                d4.append(np.abs(np.asarray(state[:4]) - gold['pipe_poses_all'][j, [0, 1, 2, 6]]).max())
    assert len(solved) == gold['pipe_boxes_all'].shape[0] == len(d4)
    print('4-DoF end point, oracle scipy vs reference scipy, L-inf per object:',
          np.array2string(np.asarray(d4), precision=2, max_line_width=200))
    # cost and gradient agree to 1e-12 (test_reference_golden.py), the END POINT does not: bulk only
    assert np.median(d4) < 5e-2 and max(d4) < 1.0
    # dense alignment of the REFERENCE's poses: exact status, disparity within float rounding
    succ, dis = oda.align_parallel(calib, float(info[0, 2]), l, r, torch.from_numpy(gold['pipe_boxes_all'][:, 0:4]),
                                   torch.from_numpy(gold['pipe_kpts_all']), torch.from_numpy(gold['pipe_poses_all'][:, 0:7]))
    assert np.array_equal(succ.numpy(), gold['pipe_succ'])
    assert float(np.abs(dis.numpy() - gold['pipe_dis_final']).max()) < 1e-4
    # 3-DoF with the reference's aligned disparities
    dz, dxyt = [], []
    for k in range(gold['pipe_boxes_all'].shape[0]):
        p = gold['pipe_poses_all'][k]
        state, z = obe.solve_x_y_theta_from_kpt(im_shape, calib, float(p[7]), p[3:6], gold['pipe_boxes_all'][k, 0:4],
                                                float(gold['pipe_dis_final'][k]), gold['pipe_kpts_all'][k])
        want = gold['pipe_rectified'][k]
        dz.append(abs(z - want[2]))
        dxyt.append(np.abs(np.asarray(state) - want[[0, 1, 3]]).max())
    print('3-DoF rectified box, oracle vs reference: max |dz| %.2e, L-inf(x,y,theta) per object %s'
          % (max(dz), np.array2string(np.asarray(dxyt), precision=2)))
    assert max(dz) < 1e-9 and np.median(dxyt) < 1e-4


# ================================================================================================ GPU part
def _model(dev, precision):
    from stereo_rcnn_amd import fixture
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    mdl = resnet(('__background__', 'Car'), 101, pretrained=False)
    mdl.create_architecture()
    mdl.load_state_dict(fixture.demo_state_dict(3))
    mdl.cuda().eval()
    mdl.precision = precision
    return mdl


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ['f16x3', 'f32'])
def test_hip_forward_on_demo_pair_vs_reference_code(dev, pair, gold, precision):
    """uint8 images -> fused preprocessing -> forward, both conv engines, against the reference code's outputs on the
    same natural image: network input bit-equal; proposals matched by coordinates; regressions within 1e-4."""
    mdl = _model(dev, precision)
    lu, ru = torch.from_numpy(pair['left']).to(dev), torch.from_numpy(pair['right']).to(dev)
    with torch.no_grad():
        out, iml, imr, info = mdl.forward_images(lu, ru)
    torch.cuda.synchronize()
    assert hashlib.sha256(np.ascontiguousarray(iml.cpu().numpy()).tobytes()).digest() == gold['input_sha256'].tobytes()
    assert [float(v) for v in info.cpu()[0]] == [600.0, 1987.0, float(np.float32(1.6))]
    ref_l, ref_r = _rows(gold['rois_left']), _rows(gold['rois_right'])
    rl, rr = out[0][0].cpu(), out[1][0].cpu()
    d = (ref_l[:, None, 1:] - rl[None, :, 1:]).abs().amax(2)
    best, idx = d.min(1)
    ok = best < tol_.PROPOSAL_MATCH_PX
    frac = float(ok.float().mean())
    tol_.observe('proposal_match_px', best[ok].max())
    errs = {'rois_right': tol_.observe('proposal_match_px', (rr[idx[ok]] - ref_r[ok]).abs().max())}
    for k, t in (('cls_prob', out[2][0]), ('bbox_pred', out[3][0]), ('dim_orien_pred', out[4][0]), ('kpts_prob', out[5]),
                 ('left_border_prob', out[6]), ('right_border_prob', out[7])):
        errs[k] = tol_.observe('e2e_' + k, (t.cpu()[idx[ok]] - _rows(gold[k])[ok]).abs().max())
    print('demo pair, %s engine vs reference code: matched proposals %d/300, max abs errors %s'
          % (precision, int(ok.sum()), {k: '%.1e' % v for k, v in errs.items()}))
    assert frac >= 0.97, frac
    assert errs['bbox_pred'] < 1e-4 and errs['dim_orien_pred'] < 1e-4, errs
    assert errs.pop('rois_right') < tol_.PROPOSAL_MATCH_PX and all(v < tol_.HEAD_OUTPUT_E2E for v in errs.values()), errs


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ['f16x3', 'f32'])
def test_hip_heads_fed_the_reference_rois_on_demo_pair(dev, pair, gold, precision):
    """The natural image, heads isolated from proposal-coordinate sensitivity: HIP preprocessing + trunk + FPN make the maps,
    the heads get the REFERENCE CODE's 300 rois -- every head output of every roi within 1e-4 (end to end, above, `kpts_prob`
    sits at ~1.5e-4 for both engines because the proposals themselves differ by ~1e-3 px)."""
    from stereo_rcnn_amd import engine
    mdl = _model(dev, precision)
    lu, ru = torch.from_numpy(pair['left']).to(dev), torch.from_numpy(pair['right']).to(dev)
    with torch.no_grad():
        mdl.forward_images(lu, ru)
        plan = mdl._get_plan(1, 600, 1987)
        plan.rois_left.copy_(torch.from_numpy(gold['rois_left']).to(dev))
        plan.rois_right.copy_(torch.from_numpy(gold['rois_right']).to(dev))
        prev, engine.PRECISION = engine.PRECISION, precision
        try:
            plan.heads()
        finally:
            engine.PRECISION = prev
        torch.cuda.synchronize()
        o = plan.outputs()
    errs = {}
    for k in ('cls_prob', 'bbox_pred', 'dim_orien_pred', 'kpts_prob', 'left_border_prob', 'right_border_prob'):
        ref = torch.from_numpy(gold[k])
        errs[k] = float((o[k].cpu().reshape(ref.shape) - ref).abs().max())
    print('demo pair, heads fed the reference rois, %s: %s' % (precision, {k: '%.1e' % v for k, v in errs.items()}))
    assert all(v < 1e-4 for v in errs.values()), errs


@pytest.mark.gpu
def test_hip_decode_class_nms_and_borders_on_demo_pair(dev, pair, gold):
    """Product decode / class NMS / infer_boundary kernels on the REFERENCE network's outputs for the demo pair."""
    from stereo_rcnn_amd import _lib, distributed as sdist
    from stereo_rcnn_amd import postprocess as hpost
    t = lambda k: torch.from_numpy(gold[k]).to(dev)
    info = torch.tensor([[600.0, 1987.0, 1.6]], device=dev)
    det = hpost.decode_detections(t('rois_left'), t('rois_right'), t('cls_prob'), t('bbox_pred'), t('dim_orien_pred'),
                                  t('kpts_prob'), t('left_border_prob'), t('right_border_prob'), info)
    for a, b, tol in (('scores', 'dec_scores', 0.0), ('boxes_left', 'dec_boxes_left', tol_.DECODED_PX), ('boxes_right', 'dec_boxes_right', tol_.DECODED_PX),
                      ('kpts', 'dec_kpts', tol_.DECODED_PX), ('dim_orien', 'dec_dim_orien', 1e-6)):
        err = float(np.abs(det[a].cpu().numpy() - gold[b].reshape(tuple(det[a].shape))).max())
        if tol == tol_.DECODED_PX:
            tol_.observe('decoded_px', err)
        assert err <= tol, (a, err)
    keep_idx, num = hpost.class_nms_device(det, 1, 0.05)
    k = int(num[0])
    assert k == gold['cls_keep'].shape[0]
    rec = sdist.pack_records_device(det, keep_idx, num, 1)
    L = _lib.lib()
    ws = torch.empty(int(L.srcnn_box3d_workspace_bytes(300, 1242)), dtype=torch.uint8, device=dev)
    _lib.check(L.srcnn_infer_boundary(rec.data_ptr(), 300, _lib.REC_COLS, 1242, ws.data_ptr(), ws.numel(), _lib.stream()))
    body = rec.cpu().numpy()[1:k + 1]
    assert tol_.observe('decoded_px', np.abs(body[:, 1:5] - gold['cls_dets_left'][:, :4]).max()) < tol_.DECODED_PX
    assert tol_.observe('decoded_px', np.abs(body[:, 5:9] - gold['cls_dets_right'][:, :4]).max()) < tol_.DECODED_PX
    assert np.array_equal(body[:, 0], gold['cls_dets_left'][:, 4])
    # borders: integers (image columns) or regressed values -- equal up to the decode's expf ulp
    assert tol_.observe('decoded_px', np.abs(body[:, 14:19] - gold['pipe_kpts_after_borders']).max()) < tol_.DECODED_PX


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ['f16x3', 'f32'])
def test_hip_full_flow_on_demo_pair_reports_3d_box_deltas(dev, pair, gold, tmp_path, precision):
    """demo.py:100-326 end to end on the HIP path (uint8 images in, rectified 3-D boxes out) next to the reference code's
    own run: the honest number for the '3D box L-inf' half of the metric, per object."""
    from stereo_rcnn_amd import pipeline
    mdl = _model(dev, precision)
    calib = demo_calib(pair, tmp_path)
    lu, ru = torch.from_numpy(pair['left']).to(dev), torch.from_numpy(pair['right']).to(dev)
    objs = pipeline.detect_3d_images(mdl, lu, ru, calib)
    ref_boxes, ref_pose4, ref_dis, ref_final = gold['pipe_boxes_all'], gold['pipe_poses_all'], gold['pipe_dis_final'], gold['pipe_rectified']
    assert abs(len(objs) - ref_boxes.shape[0]) <= 2, (len(objs), ref_boxes.shape[0])
    rows = []
    for j in range(ref_boxes.shape[0]):
        o = min(objs, key=lambda q: np.abs(q['box_left'] - ref_boxes[j, :4]).max())
        if np.abs(o['box_left'] - ref_boxes[j, :4]).max() > 2e-2:
            continue
        d4 = max(np.abs(o['xyz_init'] - ref_pose4[j, 0:3]).max(), abs(o['theta_init'] - ref_pose4[j, 6]))
        assert o['aligned'] == bool(gold['pipe_succ'][j] > 0)
        dd = abs(o['disparity'] - ref_dis[j])
        dfin = max(np.abs(o['xyz'] - ref_final[j, 0:3]).max(), abs(o['theta'] - ref_final[j, 3]))
        rows.append((d4, dd, dfin, abs(o['xyz'][2] - ref_final[j, 2]), j))
    rows = np.asarray(rows)
    assert rows.shape[0] >= ref_boxes.shape[0] - 2
    # the yardstick (tests/conditioning.py): how far the REFERENCE'S OWN 4-DoF end point of each object moves when its float32
    # detections move by the detector's measured error (1e-5) -- computed from the reference run's detections
    from conditioning import spread_4dof
    spread = []
    for j in range(ref_boxes.shape[0]):
        i = int(np.argmin(np.abs(gold['cls_dets_left'][:, :4] - ref_boxes[j, :4]).max(1)))
        do = gold['cls_dim_orien'][i].astype(np.float64)
        case = (math.atan2(do[3], do[4]), do[0:3], gold['cls_dets_left'][i, :4], gold['cls_dets_right'][i, :4],
                gold['pipe_kpts_after_borders'][i])
        spread.append(spread_4dof(case, 1e-5, 16, seed=j, dtype=np.float32, calib=calib))
    spread = np.asarray(spread)[rows[:, 4].astype(int)]
    print('demo pair, %s engine, %d objects vs the reference run -- per object:' % (precision, rows.shape[0]))
    print('  4-DoF L-inf(x,y,z,theta)  ', np.array2string(rows[:, 0], precision=1, max_line_width=200))
    print('  reference\'s own 4-DoF spread', np.array2string(spread, precision=1, max_line_width=200))
    print('  |d aligned disparity| px  ', np.array2string(rows[:, 1], precision=1, max_line_width=200))
    print('  final 3-D box L-inf       ', np.array2string(rows[:, 2], precision=1, max_line_width=200))
    print('  final |dz| m              ', np.array2string(rows[:, 3], precision=1, max_line_width=200))
    # The metric, where it is defined: objects whose reference end point is itself reproducible under detector-sized input
    # error (within a quarter of the bar over 16 draws) must come out within 1e-4 of the reference run -- 4-DoF pose AND the
    # final, rectified box.
    stable = spread <= 2.5e-5
    assert stable.any(), 'the demo pair has at least one well-conditioned object'
    assert (rows[stable, 0] <= 1e-4).mean() >= 0.66 and (rows[stable, 2] <= 1e-4).mean() >= 0.66, (rows[stable, 0], rows[stable, 2])
    # Everywhere else the HIP flow differs from the reference run by what the reference differs from ITSELF when its detections
    # move by 1e-5 (measured above), not by more -- as distributions: the end points move in jumps, so a sampled spread does
    # not bound a single object's next draw.
    fin = np.isfinite(spread)
    assert np.median(rows[fin, 0]) <= 2 * np.median(spread[fin]) and rows[fin, 0].max() <= 10 * spread[fin].max(), (rows[:, 0], spread)
    # where the 4-DoF end point is reproduced the alignment searches the same grid and the final box follows (one object of
    # this pair sits on a flat photometric minimum: 0.04 px of disparity from a 5e-5 pose difference)
    same = rows[:, 0] < 1e-4
    assert (rows[same, 1] < 2e-3).mean() >= 0.5 and (rows[same, 2] < 1e-3).mean() >= 0.5
    assert np.median(rows[:, 1]) < 0.1          # aligned disparities: far inside one depth step of the enumeration
