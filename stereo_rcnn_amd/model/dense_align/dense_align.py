"""`align_parallel` - reference lib/model/dense_align/dense_align.py:240-300, one native call."""
import torch

from ... import _lib

# per-object sample bound.  The reference's lattice (dense_align.py:42-45) has at most 113 columns (a border span w gives
# w // max(w // 56, 1) + 1 <= 113 points) and 0.4 * 113 + 1 = 46 rows, i.e. <= 5198 points, so 8192 cannot overflow; if a
# caller lowers it, an object whose lattice does not fit gets status -1 and `check_status` raises (never a silent truncation).
MAX_PIXELS = 8192


def check_status(status_host):
    """Raise if the native call reported a lattice overflow (status -1) for any object."""
    if (status_host < 0).any():
        raise RuntimeError("srcnn_dense_align: sample lattice larger than MAX_PIXELS=%d" % MAX_PIXELS)
    return status_host


def align_parallel(calib, scale, im_left, im_right, box_left, keypoints, poses, valid=None, return_search=False):
    """Dense alignment for multiple objects, depth enumeration in parallel.

    Inputs (as the reference):
        calib: object with `.p2`, `.p3` (3x4, kitti_utils.read_obj_calibration)
        scale: H_im_left / H_origin_img (im_info[0,2])
        im_left, im_right: 1 x 3 x H x W network inputs (device)
        box_left: rois x 4 in the origin image
        keypoints: rois x 5 (kpt, kpt_type, prob, left_border, right_border in the origin image)
        poses: rois x 7 (x, y, z, w, h, l, theta)
        valid (extension): optional rois float32 mask; rows <= 0 are skipped (status 0) -- lets a fixed-size batch
               straight from the device-side 4-DoF solve be aligned without compacting it on the host
        return_search (extension): also return the depth search itself (copies of the call's workspace,
               srcnn_dense_align_workspace_layout): {'count' (rois) valid lattice pixels, 'coarse_depth' / 'coarse_cost' (50, rois),
               'fine_depth' / 'fine_cost' (20, rois), 'coarse_best' / 'fine_best' (rois)} -- for auditing the discrete argmin
    Returns:
        solve_status: 1 = success, 0 = failed (no valid pixel), -1 = lattice overflow (see check_status)   (rois)
        best_dis: aligned disparity in the origin image          (rois)
    """
    assert im_left.is_cuda, "device tensors required (no CPU path)"
    dev = im_left.device
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    im_l, im_r = f(im_left), f(im_right)
    _, _, H, W = im_l.shape
    boxes = f(box_left)
    borders = f(keypoints[:, 3:5])
    poses = f(poses[:, 0:7])
    R = int(boxes.shape[0])
    status = torch.zeros((R,), dtype=torch.float32, device=dev)
    best_dis = torch.zeros((R,), dtype=torch.float32, device=dev)
    if R == 0:
        return status, best_dis
    L = _lib.lib()
    ws = _lib.workspace(L.srcnn_dense_align_workspace_bytes(H, W, R, MAX_PIXELS), dev, "dense_align")
    _lib.check(L.srcnn_dense_align(im_l.data_ptr(), im_r.data_ptr(), H, W, float(scale),
                                   float(calib.p2[0, 0]), float(calib.p2[0, 2]), float(calib.p2[1, 2]),
                                   float(calib.p2[0, 3] - calib.p3[0, 3]), boxes.data_ptr(), borders.data_ptr(),
                                   poses.data_ptr(), valid.data_ptr() if valid is not None else None, R, MAX_PIXELS,
                                   status.data_ptr(), best_dis.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _lib.stream()), "srcnn_dense_align")
    if return_search:
        import ctypes
        off = (ctypes.c_size_t * 7)()
        _lib.check(L.srcnn_dense_align_workspace_layout(H, W, R, MAX_PIXELS, off, 7), "srcnn_dense_align_workspace_layout")
        raw = ws.view(torch.uint8)
        f32 = lambda o, n: raw[o:o + 4 * n].view(torch.float32).clone()
        search = {'count': raw[off[0]:off[0] + 4 * R].view(torch.int32).clone(),
                  'coarse_depth': f32(off[1], 50 * R).view(50, R), 'coarse_cost': f32(off[2], 50 * R).view(50, R), 'coarse_best': f32(off[3], R),
                  'fine_depth': f32(off[4], 20 * R).view(20, R), 'fine_cost': f32(off[5], 20 * R).view(20, R), 'fine_best': f32(off[6], R)}
        return status, best_dis, search
    return status, best_dis
