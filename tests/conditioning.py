"""Shared helpers of the 3-D-box conditioning tests (test infrastructure; imports the oracle only to PROJECT boxes).

The metric's second half is "3D box L-inf vs reference <= 1e-4".  The reference's boxes come out of scipy's Newton-CG with a
gradient that is not the cost's gradient and scipy's default step tolerance (box_estimator.py:169-385): the iteration stops
1e-3..1e-2 short of the optimum, at a point that depends on how many iterations it took.  These helpers measure how far the
REFERENCE'S OWN end point moves when its inputs move by the detector's measured error (1e-5): the yardstick against which a
re-implementation's 3-D deltas have to be read.  The solver used is the library's host build, which is bit-identical to the
reference's scipy path (tests/test_solvers_cpu.py), so the spread is the reference's."""
import math

import numpy as np

IM_SHAPE = (375, 1242, 3)


def _wrap(d):
    d = np.array(d, dtype=np.float64)
    d[3] = (d[3] + math.pi) % (2 * math.pi) - math.pi
    return d


def well_posed_cases(n=48, seed=11):
    """Detections synthesised by projecting known cars (KITTI demo calibration) that the solver's own model explains
    EXACTLY (cost at the planted pose < 1e-12: consistent keypoint vertex, no truncation, box and keypoint inside the image),
    6..40 m away.  Returns [(alpha, dim(3), box_left(4), box_right(4), kpts(5)), planted (x, y, z, theta)]."""
    from oracle import box_estimator as obe
    from oracle.dense_align import KITTI_DEMO_CALIB as calib
    from stereo_rcnn_amd.model.utils import box_estimator as be
    rng = np.random.default_rng(seed)
    f, cx = calib.p2[0, 0], calib.p2[0, 2]
    out = []
    while len(out) < n:
        z = rng.uniform(6, 40)
        x = rng.uniform(-0.4, 0.4) * z
        th = rng.uniform(-math.pi, math.pi)
        dim = np.array([rng.uniform(1.5, 1.8), rng.uniform(1.4, 1.7), rng.uniform(3.5, 4.6)])
        bl, br, corners = obe.project_observations(calib, (x, 1.65, z, th), tuple(dim))
        kt = int(rng.integers(0, 4))
        sx, sz = obe._KPT_VERTS[kt]
        X, Z = corners[(sx, sz)]
        kp = f * X / Z + cx
        if not (bl[0] + 2 < kp < bl[2] - 2) or bl[0] < 5 or bl[2] > 1236 or bl[1] < 5 or bl[3] > 369 or br[0] < 5:
            continue
        alpha = th - math.pi / 2 + math.atan2(-x, z)
        bl = np.array(bl, np.float64)
        br = np.array([br[0], bl[1], br[2], bl[3]], np.float64)
        kpts = np.array([kp, kt, 0.9, bl[0], bl[2]], np.float64)
        planted = np.array([x, 1.65, z, th])
        cost, _ = be.evaluate_native(IM_SHAPE, calib, alpha, dim, bl, br, kpts, planted)
        if cost > 1e-12:
            continue
        out.append(((alpha, dim, bl, br, kpts), planted))
    return out


def perturb(case, eps, rng, dtype=np.float64):
    """The case with every measured quantity moved by a uniform error in [-eps, eps] (the keypoint TYPE is discrete)."""
    alpha, dim, bl, br, kpts = case
    e = lambda a: (np.asarray(a, np.float64) + rng.uniform(-eps, eps, np.shape(a))).astype(dtype)
    k2 = np.array(kpts, dtype=np.float64)
    k2[[0, 3, 4]] = e(k2[[0, 3, 4]])
    return float(alpha + rng.uniform(-eps, eps)), e(dim).astype(np.float64), e(bl), e(br), k2.astype(dtype)


def solve4(case, calib=None):
    from oracle.dense_align import KITTI_DEMO_CALIB
    from stereo_rcnn_amd.model.utils import box_estimator as be
    alpha, dim, bl, br, kpts = case
    return be.solve_x_y_z_theta_from_kpt_native(IM_SHAPE, calib or KITTI_DEMO_CALIB, alpha, dim, bl, br, kpts, return_status=True)


def spread_4dof(case, eps=1e-5, draws=8, seed=0, dtype=np.float64, calib=None):
    """max over `draws` perturbations of L-inf(x, y, z, theta) between the reference solver's end point on the perturbed and
    on the unperturbed inputs (theta compared modulo 2 pi); inf if a perturbation flips the solver's success status."""
    rng = np.random.default_rng(seed)
    st0, x0, _ = solve4(case, calib)
    if not st0:
        return float('inf')
    worst = 0.0
    for _ in range(draws):
        st1, x1, _ = solve4(perturb(case, eps, rng, dtype), calib)
        if not st1:
            return float('inf')
        worst = max(worst, float(np.abs(_wrap(x1 - x0)).max()))
    return worst
