"""CPU: the host-side rows of the path (A13 infer_boundary, A14/A17 3-D solvers, KITTI I/O):
product implementation vs the oracle's independent restatement."""
import math
import os

import numpy as np

from oracle import box_estimator as obe
from oracle import pipeline as opipe
from oracle.dense_align import KITTI_DEMO_CALIB
from stereo_rcnn_amd.model.utils import box_estimator as pbe
from stereo_rcnn_amd.model.utils import kitti_utils

IM_SHAPE = (375, 1242, 3)


def _case(rng):
    calib = KITTI_DEMO_CALIB
    z = rng.uniform(8, 40); x = rng.uniform(-0.55, 0.55) * z; y = rng.uniform(1.4, 1.8); th = rng.uniform(-math.pi, math.pi)
    dim = (1.6 * rng.uniform(0.9, 1.1), 1.5 * rng.uniform(0.9, 1.1), 4.0 * rng.uniform(0.9, 1.1))
    bl, br, corners = obe.project_observations(calib, (x, y, z, th), dim)
    types = {(-1, -1): 0, (-1, 1): 1, (1, 1): 2, (1, -1): 3}
    k = min(corners, key=lambda c: corners[c][1])
    ku = calib.p2[0, 0] * corners[k][0] / corners[k][1] + calib.p2[0, 2]
    clip = lambda b: [max(0, min(1241, b[0])), max(0, b[1]), max(0, min(1241, b[2])), min(374, b[3])]
    bl, br = clip(bl), clip(br)
    kp = [ku, float(types[k]), 0.9, bl[0], bl[2]]
    alpha = th - math.pi / 2 + math.atan2(-x, z)
    return calib, (x, y, z, th), dim, bl, br, kp, alpha


def test_cost_and_reference_gradient_agree_to_rounding():
    """The well-defined parity for this row: cost and the reference's (quirky) gradient, evaluated at the
    same points by the two independent restatements, agree to the last bits."""
    rng = np.random.default_rng(1)
    for _ in range(60):
        calib, pose, dim, bl, br, kp, alpha = _case(rng)
        p = obe._Problem(IM_SHAPE, calib, alpha, dim, bl, br, kp, True)
        cost, grad = obe._cost_and_grad(p, 0.5)
        t = pbe._Terms(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        pt = np.array(pose) + rng.normal(0, [0.3, 0.05, 1.0, 0.05])
        c2, g2 = t.evaluate(pt[0], pt[1], pt[2], pt[3], True)
        assert abs(cost(*pt) - c2) < 1e-12 * max(1.0, c2)
        assert np.abs(grad(*pt) - g2).max() < 1e-12
        assert p.trunc == t.truncation and abs(p.alpha - t.alpha) < 1e-15


def test_keypoint_gradient_quirk_is_reproduced():
    """box_estimator.py:264 doubles the keypoint residual but :311-316 differentiate it without the 2."""
    rng = np.random.default_rng(2)
    calib, pose, dim, bl, br, kp, alpha = _case(rng)
    while obe._Problem(IM_SHAPE, calib, alpha, dim, bl, br, kp, True).trunc:
        calib, pose, dim, bl, br, kp, alpha = _case(rng)
    t = pbe._Terms(IM_SHAPE, calib, alpha, dim, bl, br, kp)
    pt = np.array(pose) + 0.1
    g = t.evaluate(*pt, True)[1]
    eps = 1e-6
    num = np.array([(t.evaluate(*(pt + eps * np.eye(4)[i]), False)[0] - t.evaluate(*(pt - eps * np.eye(4)[i]), False)[0]) / (2 * eps)
                    for i in range(4)])
    assert np.abs(g - num).max() > 1e-6          # NOT the true gradient ...
    t.active['uk'] = False                       # ... but exact once the keypoint term is removed
    g2 = t.evaluate(*pt, True)[1]
    num2 = np.array([(t.evaluate(*(pt + eps * np.eye(4)[i]), False)[0] - t.evaluate(*(pt - eps * np.eye(4)[i]), False)[0]) / (2 * eps)
                     for i in range(4)])
    assert np.abs(g2 - num2).max() < 1e-6


def test_solvers_recover_pose_and_agree_statistically():
    """scipy's Newton-CG stops where its line search gives up (see box_estimator.py docstring): end points of
    two bit-different but equivalent evaluations scatter, so agreement is statistical, not pointwise."""
    rng = np.random.default_rng(3)
    d4, d3, err = [], [], []
    for _ in range(40):
        calib, pose, dim, bl, br, kp, alpha = _case(rng)
        s1, a = obe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        s2, b = pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        assert s1 == s2
        if not s1:
            continue
        d4.append(np.abs(np.array(a) - np.array(b)).max())
        err.append(abs(b[2] - pose[2]) / pose[2])
        disp = calib.p2[0, 0] * ((calib.p2[0, 3] - calib.p3[0, 3]) / calib.p2[0, 0]) / pose[2]
        r1, z1 = obe.solve_x_y_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, disp, kp)
        r2, z2 = pbe.solve_x_y_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, disp, kp)
        assert z1 == z2
        d3.append(np.abs(r1 - r2).max())
    assert np.median(d3) < 1e-4 and np.median(d4) < 5e-3
    assert np.median(err) < 0.02                 # depth recovered to ~2 % on clean synthetic observations


def test_native_newton_cg_is_scipys_iteration_bit_for_bit():
    """csrc/box_solver.h in its host build against the Python path as shipped (scipy.optimize's Newton-CG on the reference's
    cost / gradient): END POINTS BIT-IDENTICAL on every case, 4-DoF and 3-DoF.  I.e. the optimiser -- CG loop,
    finite-difference Hessian products, MINPACK-2 dcsrch, the wolfe2 / zoom fall-back, every stopping rule -- is restated
    exactly; np.dot is matched as the fused multiply-add chain OpenBLAS' ddot runs on x86, and `v ** 2` as libm's
    pow(v, 2.0) (not always the correctly rounded v * v).  The device build of the same header squares exactly and uses
    ocml's cos / sin: its end points are compared on the GPU (tests/test_box3d_gpu.py)."""
    rng = np.random.default_rng(3)
    n4 = n3 = 0
    for _ in range(120):
        calib, pose, dim, bl, br, kp, alpha = _case(rng)
        s_ref, b = pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        s_nat, a = pbe.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        assert s_ref == s_nat
        if s_ref == 0 and np.ndim(b) == 0:
            continue
        assert np.array_equal(np.asarray(a), np.asarray(b)), (a, b)
        n4 += 1
        disp = calib.p2[0, 0] * ((calib.p2[0, 3] - calib.p3[0, 3]) / calib.p2[0, 0]) / pose[2]
        r_ref, z_ref = pbe.solve_x_y_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, disp, kp)
        r_nat, z_nat = pbe.solve_x_y_theta_from_kpt_native(IM_SHAPE, calib, alpha, dim, bl, disp, kp)
        assert z_ref == z_nat and np.array_equal(r_nat, r_ref), (r_nat, r_ref)
        n3 += 1
    assert n4 >= 100 and n3 >= 100


def test_native_solver_vs_scipy_path_as_shipped():
    """Perturbed evaluation points: cost / gradient of the native code equal the Python path's to the last bit, and so do the
    end points of both solvers on every case."""
    rng = np.random.default_rng(5)
    same4, same3, d4, dc, dg = [], [], [], [], []
    for _ in range(150):
        calib, pose, dim, bl, br, kp, alpha = _case(rng)
        pt = np.array(pose) + rng.normal(0, [0.3, 0.05, 1.0, 0.05])
        c_py, g_py = pbe._Terms(IM_SHAPE, calib, alpha, dim, bl, br, kp).evaluate(pt[0], pt[1], pt[2], pt[3], True)
        c_nat, g_nat = pbe.evaluate_native(IM_SHAPE, calib, alpha, dim, bl, br, kp, pt)
        dc.append(abs(c_nat - c_py) / max(1.0, abs(c_py)))
        dg.append(np.abs(g_nat - g_py).max())
        s_ref, b = pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        s_nat, a, newton = pbe.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib, alpha, dim, bl, br, kp, return_status=True)
        assert s_ref == s_nat and newton in (0, 1, 2, 3)
        if not s_ref:
            continue
        same4.append(np.array_equal(np.asarray(a), np.asarray(b)))
        d4.append(np.abs(np.asarray(a) - np.asarray(b)).max())
        disp = calib.p2[0, 0] * ((calib.p2[0, 3] - calib.p3[0, 3]) / calib.p2[0, 0]) / pose[2]
        r_ref, _ = pbe.solve_x_y_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, disp, kp)
        r_nat, _ = pbe.solve_x_y_theta_from_kpt_native(IM_SHAPE, calib, alpha, dim, bl, disp, kp)
        same3.append(np.array_equal(r_nat, r_ref))
    assert max(dc) == 0 and max(dg) == 0
    print('native vs scipy path: bit-identical 4-DoF %.3f, 3-DoF %.3f; 4-DoF L-inf median %.1e max %.1e'
          % (np.mean(same4), np.mean(same3), np.median(d4), np.max(d4)))
    assert np.mean(same4) == 1.0 and np.mean(same3) == 1.0


def test_native_solver_with_float32_rows_as_the_pipeline_passes_them():
    """demo.py:284-293 hands box_left / box_right / dim to the solver as numpy float32 rows: numpy then evaluates the start
    disparity (and the box-size tests) in float32.  The native solver mirrors that (boxes_are_float32), so that on the rows
    the detector produces it starts from the same point as the scipy path -- bit-identical end points for the bulk."""
    rng = np.random.default_rng(9)
    same, n = 0, 0
    for _ in range(120):
        calib, pose, dim, bl, br, kp, alpha = _case(rng)
        f = lambda v: np.asarray(v, np.float32)
        bl, br, dim, kp = f(bl), f(br), f(dim), f(kp)
        s_ref, b = pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        s_nat, a = pbe.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib, alpha, dim, bl, br, kp)
        assert s_ref == s_nat
        if np.ndim(b) == 0:
            continue
        n += 1
        same += np.array_equal(np.asarray(a), np.asarray(b))
        # the float64 form of the same rows starts elsewhere (by a float32 rounding of the disparity)
    assert n >= 100 and same >= 0.9 * n, (same, n)


def test_native_early_outs_and_status():
    calib = KITTI_DEMO_CALIB
    assert pbe.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib, 0.0, (1.6, 1.5, 4.0), [100, 100, 105, 160], [90, 100, 95, 160],
                                                 [102, 0, 1, 100, 105]) == (0, 0)
    assert pbe.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib, 0.0, (1.6, 1.5, 4.0), [100, 100, 200, 160], [90, 100, 190, 160],
                                                 [150, 0, 1, 150, 152]) == (0, 0)
    # a far object: solved, but z > 100 -> status 0 with the state returned (box_estimator.py:383-384)
    st, state = pbe.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib, 0.3, (1.6, 1.5, 4.0), [600, 160, 615, 172], [598, 160, 613, 172],
                                                      [607, 0, 1, 600, 615])
    ref = pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, 0.3, (1.6, 1.5, 4.0), [600, 160, 615, 172], [598, 160, 613, 172],
                                         [607, 0, 1, 600, 615])
    assert st == ref[0] == 0 and state[2] > 100 and abs(state[2] - ref[1][2]) < 1e-3 * ref[1][2]


def test_early_outs():
    calib = KITTI_DEMO_CALIB
    assert pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, 0.0, (1.6, 1.5, 4.0), [100, 100, 105, 160], [90, 100, 95, 160],
                                          [102, 0, 1, 100, 105]) == (0, 0)          # box narrower than 10 px
    assert pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, 0.0, (1.6, 1.5, 4.0), [100, 100, 200, 160], [90, 100, 190, 160],
                                          [150, 0, 1, 150, 152]) == (0, 0)          # borders closer than 3 px
    for a in (-3.2, -1.6, -0.7, 0.0, 0.8, 1.57, 2.4, 3.1):
        assert pbe.BB2Viewpoint(a) == obe.bb2viewpoint(a)


def test_infer_boundary_matches_oracle():
    rng = np.random.default_rng(4)
    for _ in range(20):
        n = rng.integers(1, 9)
        x1 = rng.uniform(0, 1100, n); w = rng.uniform(20, 300, n); y2 = rng.uniform(150, 374, n)
        boxes = np.stack([x1, y2 - rng.uniform(20, 120, n), np.minimum(x1 + w, 1241), y2, rng.uniform(0, 1, n)], 1).astype(np.float32)
        a = kitti_utils.infer_boundary(IM_SHAPE, boxes)
        b = opipe.infer_boundary(IM_SHAPE, boxes)
        assert np.array_equal(a, b)
    # an occluder in front hides the right part of the farther box
    boxes = np.array([[100, 100, 300, 200, 0.9], [250, 100, 500, 300, 0.8]], np.float32)
    lr = kitti_utils.infer_boundary(IM_SHAPE, boxes)
    assert lr[0, 0] == 100 and lr[0, 1] < 300 and lr[1, 0] == 250 and lr[1, 1] == 500


def test_calibration_reader_and_result_writer(tmp_path):
    calib_txt = tmp_path / 'calib.txt'
    rows = {'P0': [721.5377, 0, 609.5593, 0, 0, 721.5377, 172.854, 0, 0, 0, 1, 0],
            'P1': [721.5377, 0, 609.5593, -387.5744, 0, 721.5377, 172.854, 0, 0, 0, 1, 0],
            'P2': [721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884],
            'P3': [721.5377, 0, 609.5593, -339.5242, 0, 721.5377, 172.854, 2.199936, 0, 0, 1, 0.002729905],
            'R0_rect': [1, 0, 0, 0, 1, 0, 0, 0, 1], 'Tr_velo_to_cam': list(range(12)), 'Tr_imu_to_velo': list(range(12))}
    calib_txt.write_text('\n'.join('%s: %s' % (k, ' '.join('%.12e' % v for v in vals)) for k, vals in rows.items()) + '\n')
    c = kitti_utils.read_obj_calibration(str(calib_txt))
    assert c.p2.shape == (3, 4) and abs(c.p2[0, 3] - 44.85728) < 1e-9 and abs(c.p3[0, 3] + 339.5242) < 1e-9
    assert abs(c.t_cam2_cam0[0] - 44.85728 / 721.5377) < 1e-12 and c.p2_3[0, 3] == c.p3[0, 3] - c.p2[0, 3]
    assert np.allclose(c.p2, KITTI_DEMO_CALIB.p2) and np.allclose(c.p3, KITTI_DEMO_CALIB.p3)
    kitti_utils.write_detection_results(str(tmp_path / 'res'), '000012', c, [10, 20, 110, 90], [1.0, 1.6, 20.0],
                                        [1.6, 1.5, 4.0], 0.3, 0.91)
    line = (tmp_path / 'res' / 'data' / '000012.txt').read_text().split()
    assert line[0] == 'Car' and len(line) == 16
    assert abs(float(line[11]) - (1.0 - c.t_cam2_cam0[0])) < 1e-5          # cam2 -> cam0 shift on x
    assert abs(float(line[14]) - (0.3 - 1.57)) < 1e-5 and abs(float(line[15]) - 0.91) < 1e-6
    assert abs(float(line[3]) - (0.3 - math.pi / 2 + math.atan2(-1.0, 20.0))) < 1e-5
    kitti_utils.write_detection_results(None, 'x', c, [0] * 4, [0, 0, 1], [1, 1, 1], 0, 0)    # result_dir None: no-op


def test_solver_pool_matches_serial():
    """pipeline.SolverPool fans the per-object scipy solves out to worker processes; same function, same inputs ->
    bitwise the same answers as the serial path."""
    import os
    import numpy as np
    from stereo_rcnn_amd import pipeline
    m = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_misc.npz'))
    tasks = [(4, (375, 1242, 3), m['calib_p2'], m['calib_p3'], (r[0], r[1:4], r[4:8], r[8:12], r[12:17]))
             for r in m['solver_cases'][:8]]
    tasks += [(3, (375, 1242, 3), m['calib_p2'], m['calib_p3'], (r[0], r[1:4], r[4:8], 25.0, r[12:17]))
              for r in m['solver_cases'][:8]]
    serial = [pipeline._solve_task(t) for t in tasks]
    with pipeline.SolverPool(2) as pool:
        par = pool.map(tasks)
    for a, b in zip(serial, par):
        assert np.array_equal(np.asarray(a[0]), np.asarray(b[0])) and np.array_equal(np.asarray(a[1]), np.asarray(b[1]))


def test_record_forms_on_the_host_equal_the_scipy_path_bit_for_bit():
    """srcnn_solve_4dof_records_host / srcnn_solve_3dof_records_host (what pipeline solver='host' runs between the device
    stages) on a synthetic image record: every row's status and end point equal to the Python scipy path called as demo.py
    calls it (float32 rows for the 4-DoF solve; float32 alpha / dim, double boxes for the 3-DoF one), for any thread count."""
    import ctypes
    from stereo_rcnn_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(21)
    calib = KITTI_DEMO_CALIB
    n, k = 64, 40
    rec = np.zeros((n + 1, _lib.REC_COLS), np.float32)
    rec[0, 0] = k
    cases = []
    for i in range(k):
        _, pose, dim, bl, br, kp, alpha = _case(rng)
        row = rec[1 + i]
        row[0] = 0.9 if i % 7 else 0.01                      # one in seven below the threshold
        row[1:5], row[5:9], row[9:12] = bl, br, dim
        row[12], row[13] = math.sin(alpha), math.cos(alpha)
        row[14:19] = kp
        cases.append(pose)
    cal = (float(calib.p2[0, 0]), float(calib.p2[0, 2]), float(calib.p2[1, 2]), float(calib.p2[0, 3] - calib.p3[0, 3]))
    outs = []
    for threads in (1, 5):
        r, st = rec.copy(), np.full((2, n, 4), 7.0)
        assert L.srcnn_solve_4dof_records_host(r.ctypes.data, n, _lib.REC_COLS, 375, 1242, *cal, 0.05, st[0].ctypes.data, threads) == 0
        status = (r[1:k + 1, 20] > 0).astype(np.float32)
        status[::3] = 0                                       # alignment "failed" on a third of them
        dis = np.zeros(n, np.float32)
        dis[:k] = [cal[0] * (cal[3] / cal[0]) / max(p[2], 1.0) * 1.01 for p in cases]
        full = np.zeros(n, np.float32); full[:k] = status
        assert L.srcnn_solve_3dof_records_host(r.ctypes.data, n, _lib.REC_COLS, 375, 1242, *cal, full.ctypes.data, dis.ctypes.data,
                                               st[1].ctypes.data, threads) == 0
        outs.append((r, st.copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1][0], outs[1][1][0])
    r, st = outs[0]
    assert (st[0, k:] == 0).all()
    n4 = n3 = 0
    for i in range(k):
        row = rec[1 + i]
        if not row[0] > 0.05:
            assert r[1 + i, 20] == 0 and (st[0, i] == 0).all()
            continue
        alpha = math.atan2(row[12], row[13])
        s_ref, b = pbe.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, row[9:12], row[1:5], row[5:9], row[14:19])
        assert int(r[1 + i, 20]) == int(s_ref)
        if not s_ref:
            continue
        assert np.array_equal(st[0, i], np.asarray(b, np.float64)), (i, st[0, i], b)
        assert np.array_equal(r[1 + i, 21:25], np.asarray(b, np.float32))
        n4 += 1
        if not (full[i] > 0):
            assert r[1 + i, 25] == 0
            continue
        f64 = lambda v: np.asarray(v, np.float64)
        state, z = pbe.solve_x_y_theta_from_kpt(IM_SHAPE, calib, float(np.float32(alpha)), f64(row[9:12]), f64(row[1:5]),
                                                float(dis[i]), f64(row[14:19]))
        assert np.array_equal(st[1, i], np.array([state[0], state[1], z, state[2]])), (i, st[1, i], state, z)
        assert r[1 + i, 26] == dis[i] and r[1 + i, 25] == 1
        n3 += 1
    assert n4 >= 25 and n3 >= 12
