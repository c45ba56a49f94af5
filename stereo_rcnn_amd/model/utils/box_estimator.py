"""3-D box solvers - reference lib/model/utils/box_estimator.py:169-385 (4-DoF) and :387-545 (3-DoF).

Both reference solvers hand a sum of squared re-projection residuals and a hand-written gradient to
`scipy.optimize.minimize(method='Newton-CG')`.  That gradient is not the gradient of the cost (the doubled keypoint
residual, :264, is differentiated without its factor 2, :311-316), so the point scipy stops at is decided by its line
search giving up, not by a stationarity condition; the only way to return the reference's 3-D boxes is to run the
same iteration.  Three forms of it live here / behind this module:

  * `solve_x_y_z_theta_from_kpt` / `solve_x_y_theta_from_kpt` -- the reference's own arrangement (host numpy + scipy per
    object); `pipeline.detect_3d(solver='scipy')` uses them; they are the comparison baseline of the tests.
  * `*_native` -- the same Newton-CG (scipy 1.15's `_minimize_newtoncg`, MINPACK-2 line search, restated once in
    csrc/box_solver.h) called per object through the C ABI (`srcnn_solve_4dof_host` / `srcnn_solve_3dof_host`): host build,
    BIT-IDENTICAL to the scipy path (tests/test_solvers_cpu.py, test_box3d_gpu.py).  The pipeline's DEFAULT
    (`solver='host'`) runs that host build over the whole detection record in C threads between the device stages.
  * the device build of the same source (`solver='device'`, csrc/box3d.hip): same iteration with ocml's libm -- numerically
    equivalent, not reference-identical.

Interface = the reference's: `solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, box_left,
box_right, kpts) -> (status, state)` and `solve_x_y_theta_from_kpt(im_shape, calib, alpha, dim,
box_left, disparity, kpts) -> (state, z)`.
"""
import math as m

import numpy as np
from scipy.optimize import minimize

TRUNCATE_BORDER = 10


def BB2Viewpoint(alpha):
    """Discrete viewpoint (0..7, -1 = none) of a viewpoint angle (:15-41)."""
    alpha = alpha * 180.0 / m.pi
    if alpha > 360:
        alpha = alpha - 360
    elif alpha < -360:
        alpha = alpha + 360
    thr = 4.0
    bands = ((-90.0 - thr, -90.0 + thr, 0), (-180.0 + thr, -90.0 - thr, 1), (90.0 + thr, 180.0 - thr, 3),
             (90.0 - thr, 90.0 + thr, 4), (0.0 + thr, 90.0 - thr, 5), (0.0 - thr, 0.0 + thr, 6),
             (-90.0 + thr, 0.0 - thr, 7))
    if bands[0][0] <= alpha <= bands[0][1]:
        return 0
    if bands[1][0] <= alpha <= bands[1][1]:
        return 1
    if alpha >= 180.0 - thr or alpha <= -180.0 + thr:
        return 2
    for lo, hi, v in bands[2:]:
        if lo <= alpha <= hi:
            return v
    return -1


_SIDE_VERTS = (   # (left, right, bottom) vertex as (sign_w, sign_l) for viewpoints 0..7 (:92-122)
    ((-1, -1), (1, -1), (1, -1)), ((-1, 1), (1, -1), (-1, -1)), ((-1, 1), (-1, -1), (-1, -1)),
    ((1, 1), (-1, -1), (-1, 1)), ((1, 1), (-1, 1), (-1, 1)), ((1, -1), (-1, 1), (1, 1)),
    ((1, -1), (1, 1), (1, 1)), ((-1, -1), (1, 1), (1, -1)))
_KPT_VERTS = ((-1, -1), (-1, 1), (1, 1), (1, -1))   # keypoint types 0..3 (:138-146)


def viewpoint2vertex(view_point, w, l):
    """3-D vertices (object frame) seen at the left / right / bottom side of the 2-D box (:43-124)."""
    signs = _SIDE_VERTS[view_point] if 0 <= view_point <= 7 else _SIDE_VERTS[7]
    return tuple(np.array([sw * w, 0, sl * l]) / 2 for sw, sl in signs)


def kpt2vertex(kpt_type, w, l):
    sw, sl = _KPT_VERTS[kpt_type]
    return np.array([sw * w, 0, sl * l]) / 2


def kpt2alpha(kpt_pos, kpt_type, box):
    """Approximate viewpoint angle from the perspective keypoint (:150-167)."""
    ratio = max(min(1, (kpt_pos - box[0]) / (box[2] - box[0])), -1)
    return (-m.pi / 2, m.pi, m.pi / 2, 0.0)[kpt_type] - m.asin(ratio)


class _Terms(object):
    """Observation set-up shared by the two solvers and the residual/gradient evaluation."""

    def __init__(self, im_shape, calib, alpha, dim, box_left, box_right, kpts):
        h_max, w_max = im_shape[0], im_shape[1]
        self.h = float(dim[1])
        w, l = float(dim[0]), float(dim[2])
        ul, vt, ur, vb = (float(v) for v in box_left[:4])
        f = calib.p2[0, 0]
        cx, cy = calib.p2[0, 2], calib.p2[1, 2]
        self.f = f
        self.bl = (calib.p2[0, 3] - calib.p3[0, 3]) / f
        kpt_pos, kpt_type = float(kpts[0]), int(kpts[1])
        self.obs = {'ul': (ul - cx) / f, 'ur': (ur - cx) / f, 'vt': (vt - cy) / f, 'vb': (vb - cy) / f,
                    'uk': (kpt_pos - cx) / f}
        self.truncation = ul < 2.0 * TRUNCATE_BORDER or ur > w_max - 2.0 * TRUNCATE_BORDER
        if not self.truncation:          # in the truncation case the regressed alpha replaces the keypoint
            alpha = kpt2alpha(kpt_pos, kpt_type, box_left)
        self.alpha = alpha
        lv, rv, bv = viewpoint2vertex(BB2Viewpoint(alpha), w, l)
        kv = kpt2vertex(kpt_type, w, l)
        self.vert = {'ul': (lv[0], lv[2]), 'ur': (rv[0], rv[2]), 'uk': (kv[0], kv[2]), 'b': (bv[0], bv[2])}
        # residuals that the reference zeroes (:254-276, :464-480)
        self.active = {'ul': not ul < 2.0 * TRUNCATE_BORDER, 'ur': not ur > w_max - 2.0 * TRUNCATE_BORDER,
                       'uk': not self.truncation, 'alpha': self.truncation,
                       'vt': not vt < TRUNCATE_BORDER, 'vb': not vb > h_max - TRUNCATE_BORDER,
                       'ul_r': False, 'ur_r': False}
        if box_right is not None:
            ul_r, ur_r = float(box_right[0]), float(box_right[2])
            self.obs['ul_r'], self.obs['ur_r'] = (ul_r - cx) / f, (ur_r - cx) / f
            self.active['ul_r'] = self.truncation and not ul_r < 2.0 * TRUNCATE_BORDER
            self.active['ur_r'] = self.truncation and not ur_r > w_max - 2.0 * TRUNCATE_BORDER

    def evaluate(self, x, y, z, theta, want_grad):
        """Cost (sum of squares) and, if asked, the REFERENCE's gradient (x, y, z, theta)."""
        ct, st = np.cos(theta), np.sin(theta)
        cost = 0.0
        g = np.zeros(4)
        for name, vkey, shift, scale in (('ul', 'ul', 0.0, 1.0), ('ur', 'ur', 0.0, 1.0), ('uk', 'uk', 0.0, 2.0),
                                         ('ul_r', 'ul', self.bl, 1.0), ('ur_r', 'ur', self.bl, 1.0)):
            if not self.active[name]:
                continue
            vw, vl = self.vert[vkey]
            num = x - shift + ct * vw + st * vl
            den = z - st * vw + ct * vl
            res = scale * (num / den - self.obs[name])            # res_uk = 2*res_uk (:264)
            cost += res ** 2
            if want_grad:                                         # d/dx..dth of res**2 as the reference writes it
                g[0] += 2.0 * res / den
                g[2] += -2.0 * res * num / den ** 2
                g[3] += 2.0 * res * ((vl * ct - vw * st) / den + (vw * ct + vl * st) * num / den ** 2)
        bw, bl_ = self.vert['b']
        if self.active['vb']:
            den = z - st * bw + ct * bl_
            res = y / den - self.obs['vb']
            cost += res ** 2
            if want_grad:
                g[1] += 2.0 * res / den
                g[2] += -2.0 * res * y / den ** 2
                g[3] += 2.0 * res * (y * (bw * ct + bl_ * st)) / den ** 2
        if self.active['vt']:
            den = z + st * bw - ct * bl_
            res = (y - self.h) / den - self.obs['vt']
            cost += res ** 2
            if want_grad:
                g[1] += 2.0 * res / den
                g[2] += 2.0 * res * (self.h - y) / den ** 2
                g[3] += 2.0 * res * ((self.h - y) * (bw * ct + bl_ * st)) / den ** 2
        if self.active['alpha']:
            res = theta - m.pi / 2 + m.atan2(-x, z) - self.alpha
            cost += res ** 2
            if want_grad:
                q = 1.0 + (-x / z) ** 2
                g[0] += 2.0 * res / q * (-1.0 / z)
                g[2] += 2.0 * res / q * (x / (z * z))
                g[3] += 2.0 * res
        return cost, g


def solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, box_left, box_right, kpts):
    """Initial 3-D box from the 2-D boxes and the keypoint / alpha (:169-385).
    Returns (status, state): status 0 = failed, 1 = normal; state = (x, y, z, theta)."""
    if kpts[4] - kpts[3] < 3 or box_left[2] - box_left[0] < 10 or box_left[3] - box_left[1] < 10:
        return 0, 0
    t = _Terms(im_shape, calib, alpha, dim, box_left, box_right, kpts)
    disparity = (box_left[0] + box_left[2]) / 2 - (box_right[0] + box_right[2]) / 2
    init_z = t.f * t.bl / disparity
    init_x = init_z * (t.obs['ul'] + t.obs['ur']) / 2.0
    init_y = init_z * (t.obs['vb'] + t.obs['vt']) / 2.0 + t.h / 2.0
    init_theta = t.alpha + m.pi / 2 - m.atan2(-init_x, init_z)
    res = minimize(lambda s: t.evaluate(s[0], s[1], s[2], s[3], False)[0], [init_x, init_y, init_z, init_theta],
                   method='Newton-CG', jac=lambda s: t.evaluate(s[0], s[1], s[2], s[3], True)[1],
                   options={'disp': False})
    if res.x[2] > 100:
        return 0, res.x
    return 1, res.x


def solve_x_y_theta_from_kpt(im_shape, calib, alpha, dim, box_left, disparity, kpts):
    """3-D rectification with the depth fixed by the aligned disparity (:387-545).
    Returns (state, z) with state = (x, y, theta)."""
    t = _Terms(im_shape, calib, alpha, dim, box_left, None, kpts)
    z = t.f * t.bl / float(disparity)
    init_x = z * (t.obs['ul'] + t.obs['ur']) / 2.0
    init_y = z * (t.obs['vb'] + t.obs['vt']) / 2.0 + t.h / 2.0
    init_theta = t.alpha + m.pi / 2 - m.atan2(-init_x, z)
    res = minimize(lambda s: t.evaluate(s[0], s[1], z, s[2], False)[0], [init_x, init_y, init_theta],
                   method='Newton-CG', jac=lambda s: t.evaluate(s[0], s[1], z, s[2], True)[1][[0, 1, 3]],
                   options={'disp': False})
    return res.x, z


# ------------------------------------------------------------------------------------------------ native solvers
# The same two functions without scipy / Python in the loop: csrc/box_solver.h restates scipy's Newton-CG (its CG loop, the
# MINPACK-2 dcsrch line search and the wolfe2 fall-back) in double precision; the device pipeline runs it as kernels
# (srcnn_solve_4dof / srcnn_solve_3dof on the detection record), these two call the identical code compiled for the host.
def _calib_args(calib):
    return float(calib.p2[0, 0]), float(calib.p2[0, 2]), float(calib.p2[1, 2]), float(calib.p2[0, 3] - calib.p3[0, 3])


def _dbl(values, n):
    import ctypes
    return (ctypes.c_double * n)(*[float(v) for v in list(values)[:n]])


def solve_x_y_z_theta_from_kpt_native(im_shape, calib, alpha, dim, box_left, box_right, kpts, return_status=False):
    """`solve_x_y_z_theta_from_kpt` on the native solver (host build).  Same arguments and (status, state) result."""
    import ctypes
    from ... import _lib
    state, ns = (ctypes.c_double * 4)(), ctypes.c_int(0)
    # numpy float32 box rows (what demo.py:284-285 hands over) get numpy's float32 arithmetic for the box-size tests and
    # the start disparity, exactly as `solve_x_y_z_theta_from_kpt` above would evaluate them
    f32 = int(getattr(box_left, 'dtype', None) == np.float32 and getattr(box_right, 'dtype', None) == np.float32)
    status = _lib.lib().srcnn_solve_4dof_host(int(im_shape[0]), int(im_shape[1]), *_calib_args(calib), float(alpha), _dbl(dim, 3),
                                              _dbl(box_left, 4), _dbl(box_right, 4), _dbl(kpts, 5), state, ctypes.byref(ns), f32)
    if ns.value == -1:                      # the early-out of :186-187 returns (0, 0)
        return (0, 0, -1) if return_status else (0, 0)
    out = np.array(list(state), dtype=np.float64)
    return (status, out, ns.value) if return_status else (status, out)


def solve_x_y_theta_from_kpt_native(im_shape, calib, alpha, dim, box_left, disparity, kpts, return_status=False):
    """`solve_x_y_theta_from_kpt` on the native solver (host build).  Same arguments and (state, z) result."""
    import ctypes
    from ... import _lib
    state, z, ns = (ctypes.c_double * 3)(), ctypes.c_double(0), ctypes.c_int(0)
    _lib.check(_lib.lib().srcnn_solve_3dof_host(int(im_shape[0]), int(im_shape[1]), *_calib_args(calib), float(alpha),
                                                _dbl(dim, 3), _dbl(box_left, 4), float(disparity), _dbl(kpts, 5), state,
                                                ctypes.byref(z), ctypes.byref(ns)), "srcnn_solve_3dof_host")
    out = np.array(list(state), dtype=np.float64)
    return (out, z.value, ns.value) if return_status else (out, z.value)


def evaluate_native(im_shape, calib, alpha, dim, box_left, box_right, kpts, xyzt):
    """(cost, reference-gradient (4)) of the native restatement at (x, y, z, theta); box_right None = the 3-DoF terms."""
    import ctypes
    from ... import _lib
    cost, g = ctypes.c_double(0), (ctypes.c_double * 4)()
    _lib.check(_lib.lib().srcnn_solver_evaluate_host(int(im_shape[0]), int(im_shape[1]), *_calib_args(calib), float(alpha),
                                                     _dbl(dim, 3), _dbl(box_left, 4),
                                                     None if box_right is None else _dbl(box_right, 4), _dbl(kpts, 5),
                                                     _dbl(xyzt, 4), ctypes.byref(cost), g), "srcnn_solver_evaluate_host")
    return cost.value, np.array(list(g), dtype=np.float64)
