"""CPU oracle: dense photometric alignment (TEST INFRASTRUCTURE ONLY).

Restates, with torch-CPU float32 tensors and Python doubles exactly where the reference
(PyTorch 0.3: indexing a tensor with ints yields a Python float) has them:
  align_parallel      lib/model/dense_align/dense_align.py:240-300
  sample              lib/model/dense_align/dense_align.py:13-69
  enumeration_depth   lib/model/dense_align/dense_align.py:175-238
  Box3d / BoxRayInsec lib/model/dense_align/box_3d.py:12-106
PyTorch-0.3 semantics spelled out: F.upsample(bilinear) and F.grid_sample are
align_corners=True; torch.cat of a rank-deficient operand at box_3d.py:97 is
`ones[..., None]`; argmin takes the first minimum.
Note: the cost is SAD (L1), dense_align.py:231.
Parity status: PINNED -- `align_parallel` reproduces the reference's own align_parallel (run in the build container under
tests/golden/reference_shims.py) exactly: same status, max |d disparity| = 0 on 18 objects (tests/test_reference_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PLANE_GROUP = [[0, 3, 4], [2, 3, 4], [1, 2, 4], [0, 1, 4], [0, 3, 5], [2, 3, 5], [1, 2, 5], [0, 1, 5]]   # box_3d.py:88-96
DOUBLE_EPS = 0.01


class Calib(object):
    """The two fields of kitti_utils.FrameCalibrationData the path reads (kitti_utils.py:97-159)."""

    def __init__(self, p2, p3):
        self.p2 = np.asarray(p2, np.float64).reshape(3, 4)
        self.p3 = np.asarray(p3, np.float64).reshape(3, 4)


KITTI_DEMO_CALIB = Calib(   # demo/calib.txt P2 / P3 (values quoted in SURVEY 8(c))
    [721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884],
    [721.5377, 0, 609.5593, -339.5242, 0, 721.5377, 172.854, 2.199936, 0, 0, 1, 0.002729905])


def _f32(x):
    return torch.tensor(float(x), dtype=torch.float32)


class Box3d(object):
    """box_3d.py:12-60."""

    def __init__(self, pose):
        p = [float(v) for v in pose]          # Python floats, as tensor[int] gave in torch 0.3
        self.T = pose[0:3].clone()
        sx, sy, sz = p[3], p[4], p[5]
        c, s = math.cos(p[6]), math.sin(p[6])
        self.R = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=torch.float32)
        self.P_o = torch.tensor([[-sx / 2, 0, -sz / 2.0], [-sx / 2, 0, sz / 2.0], [sx / 2, 0, sz / 2.0],
                                 [sx / 2, 0, -sz / 2.0], [-sx / 2, -sy, -sz / 2.0], [-sx / 2, -sy, sz / 2.0],
                                 [sx / 2, -sy, sz / 2.0], [sx / 2, -sy, -sz / 2.0]], dtype=torch.float32)
        P_c = torch.stack([torch.mv(self.R, self.P_o[i]) + self.T for i in range(8)])

        def plane(p1, p2, p3):
            n = torch.linalg.cross(p2 - p1, p3 - p1)
            return torch.stack((n[0], n[1], n[2], -n[0] * p1[0] - n[1] * p1[1] - n[2] * p1[2]))

        self.planes = torch.stack([plane(P_c[0], P_c[3], P_c[4]), plane(P_c[2], P_c[3], P_c[6]),
                                   plane(P_c[1], P_c[2], P_c[5]), plane(P_c[0], P_c[1], P_c[4]),
                                   plane(P_c[0], P_c[1], P_c[2]), plane(P_c[4], P_c[5], P_c[6])])
        best, self.nearest = 100000000.0, 0
        for i in range(8):                     # strict '<': the first nearest vertex wins (box_3d.py:55-60)
            d = float(torch.norm(P_c[i]))
            if d < best:
                best, self.nearest = d, i

    def ray_intersect(self, norm_uv):
        """BoxRayInsec + mask_out_box (box_3d.py:62-106).  norm_uv (nr, nc, 2) -> (nr, nc, 4)."""
        homo = torch.cat((norm_uv, torch.ones_like(norm_uv[:, :, :1])), 2)
        out = homo.new_zeros(homo.shape[0], homo.shape[1], 4)
        Rt = self.R.t()
        lo = [_f32(float(self.P_o[4, k]) - DOUBLE_EPS) for k in range(3)]
        hi = [_f32(float(self.P_o[2, k]) + DOUBLE_EPS) for k in range(3)]
        for i in range(3):
            pl = self.planes[PLANE_GROUP[self.nearest][i]]
            t = homo[:, :, 0] * pl[0] + homo[:, :, 1] * pl[1] + homo[:, :, 2] * pl[2]
            t = -t.reciprocal() * pl[3]
            ic = homo * t.unsqueeze(2) - self.T
            io = torch.stack([Rt[k, 0] * ic[:, :, 0] + Rt[k, 1] * ic[:, :, 1] + Rt[k, 2] * ic[:, :, 2] for k in range(3)], 2)
            mask = torch.ones_like(t, dtype=torch.bool)
            for k in range(3):
                mask &= (io[:, :, k] >= lo[k]) & (io[:, :, k] <= hi[k])
            todo = out[:, :, 3] == 0
            out[:, :, 0:3][todo] = ic[todo]
            out[:, :, 3][todo] = mask.float()[todo]
        return out


def sample(calib, scale, f_h, f_w, box_left, poses, borders):
    """dense_align.py:13-69 -> all_uvz (R, P, 3), all_weight (R, P)."""
    f = calib.p2[0, 0] * scale
    cx, cy = calib.p2[0, 2] * scale, calib.p2[1, 2] * scale
    us = torch.arange(f_w, dtype=torch.float32)
    vs = torch.arange(f_h, dtype=torch.float32)
    uvz_list, max_pixels = [], 0
    for i in range(box_left.shape[0]):
        b = [float(v) for v in box_left[i]]
        bl, br = float(borders[i, 0]), float(borders[i, 1])
        # torch-0.3 scalars: float32 tensor arithmetic happens first for borders[i,1]-borders[i,0] (both Python
        # floats there) -> the reference computes these in double on float32-valued inputs
        width = max(int((br - bl) / 56.0), 1)
        height = max(int((b[3] - b[1]) / 56.0), 1)
        rows = vs[slice(int((b[1] + b[3]) / 2.0 + 0.5), int(b[3] - (b[3] - b[1]) * 0.1 + 0.5), height)]
        cols = us[slice(int(bl + 0.5), int(br + 0.5), width)]
        if rows.numel() == 0 or cols.numel() == 0:
            uvz_list.append(torch.zeros(0, 3))
            continue
        u = cols.view(1, -1).expand(rows.numel(), -1)
        v = rows.view(-1, 1).expand(-1, cols.numel())
        norm = torch.stack(((u - cx) / f, (v - cy) / f), 2)
        ins = Box3d(poses[i]).ray_intersect(norm)
        ok = ins[:, :, 3] == 1
        uvz = torch.stack((u[ok], v[ok], ins[:, :, 2][ok]), 1)
        max_pixels = max(max_pixels, uvz.shape[0])
        uvz_list.append(uvz)
    R = box_left.shape[0]
    all_uvz = torch.zeros(R, max_pixels, 3)
    all_w = torch.zeros(R, max_pixels)
    for i, uvz in enumerate(uvz_list):
        all_uvz[i, :uvz.shape[0]] = uvz
        all_w[i, :uvz.shape[0]] = 1.0
    return all_uvz, all_w


def enumeration_depth(im_left, im_right, all_uvz, all_weight, depth_enum, fb):
    """dense_align.py:175-238: depth_enum (iters, R) -> best depth per roi (R)."""
    iters, R, P = depth_enum.shape[0], all_uvz.shape[0], all_uvz.shape[1]
    f_h, f_w = float(im_left.shape[2]) - 1, float(im_left.shape[3]) - 1
    gl = all_uvz.new_zeros(1, R, P, 2)
    gl[0, :, :, 0] = (all_uvz[:, :, 0] - f_w / 2) / (f_w / 2)
    gl[0, :, :, 1] = (all_uvz[:, :, 1] - f_h / 2) / (f_h / 2)
    gl = gl.expand(iters, -1, -1, -1).contiguous().view(1, -1, P, 2)
    gr = gl.clone()
    w = all_weight.unsqueeze(1).expand(-1, 3, -1).unsqueeze(0).expand(iters, -1, -1, -1).contiguous().view(-1, 3, P)
    de = depth_enum.view(-1).unsqueeze(1).expand(-1, P)
    dis_enum = de.reciprocal() * fb
    ldd = all_uvz[:, :, 2].unsqueeze(0).expand(iters, -1, -1).contiguous().view(-1, P)
    gdd = (ldd / fb + dis_enum.reciprocal()).reciprocal()
    all_u = all_uvz[:, :, 0].unsqueeze(0).expand(iters, -1, -1).contiguous().view(-1, P)
    gr[0, :, :, 0] = (all_u - gdd - f_w / 2) / (f_w / 2)
    err = F.grid_sample(im_left, gl, mode='bilinear', padding_mode='border', align_corners=True) \
        - F.grid_sample(im_right, gr, mode='bilinear', padding_mode='border', align_corners=True)
    err = err.squeeze(0).permute(1, 0, 2).contiguous() * w
    err_sum = torch.sum(torch.abs(err.view(iters, R, -1)), 2)             # L1 (dense_align.py:231)
    idx = torch.from_numpy(np.argmin(err_sum.numpy(), axis=0))            # first minimum
    return depth_enum[idx, torch.arange(R)], err_sum


def align_parallel(calib, scale, im_left, im_right, box_left, keypoints, poses, return_extra=False):
    """dense_align.py:240-300.  scale: Python float (im_info[0,2]); im_*: (1,3,H,W) network inputs;
    box_left (R,4) and keypoints (R,5) in original-image pixels; poses (R,7).
    Returns (solve_status (R), best_dis (R))."""
    scale = scale * 2
    up = lambda t: F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=True)
    im_left, im_right = up(im_left), up(im_right)
    f = calib.p2[0, 0] * scale
    bl = (calib.p2[0, 3] - calib.p3[0, 3]) * scale / f
    box_left = box_left * scale
    keypoints = keypoints * scale
    dis_init = f * bl / poses[:, 2]
    all_uvz, all_weight = sample(calib, scale, im_left.shape[2], im_left.shape[3], box_left, poses, keypoints[:, 3:5])
    status = box_left.new_zeros(box_left.shape[0])
    if float(torch.sum(all_weight)) == 0:
        return (status, dis_init, {}) if return_extra else (status, dis_init)
    status = status + 1.0
    status[torch.sum(all_weight, 1) == 0] = 0
    iters, interval = 50, 0.5
    depth_enum = torch.stack([dis_init.reciprocal() * f * bl - iters * interval / 2 + interval * i for i in range(iters)])
    depth_enum[depth_enum < 1.5] = 1.5
    best, cost_c = enumeration_depth(im_left, im_right, all_uvz, all_weight, depth_enum, f * bl)
    tune, tint = 20, interval * 2.0 / 20
    tune_enum = torch.stack([best - tune * tint / 2 + tint * i for i in range(tune)])
    best2, cost_f = enumeration_depth(im_left, im_right, all_uvz, all_weight, tune_enum, f * bl)
    best_dis = f * bl / (best2 * scale) + 0.5
    if return_extra:
        return status, best_dis, {'uvz': all_uvz, 'weight': all_weight, 'coarse_cost': cost_c, 'fine_cost': cost_f,
                                  'coarse_depth': best, 'fine_depth': best2, 'depth_enum': depth_enum}
    return status, best_dis


def project_box(calib, pose):
    """Helper for synthetic test cases: 2-D bounding box (original pixels) of a 3-D box pose."""
    x, y, z, w, h, l, th = [float(v) for v in pose]
    c, s = math.cos(th), math.sin(th)
    pts = []
    for dx in (-w / 2, w / 2):
        for dy in (0, -h):
            for dz in (-l / 2, l / 2):
                pts.append((c * dx + s * dz + x, dy + y, -s * dx + c * dz + z))
    f, cx, cy = calib.p2[0, 0], calib.p2[0, 2], calib.p2[1, 2]
    us = [f * p[0] / p[2] + cx for p in pts]
    vs = [f * p[1] / p[2] + cy for p in pts]
    return min(us), min(vs), max(us), max(vs)
