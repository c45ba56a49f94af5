"""Stereo 3-D detection of one pair, end to end - the flow of the reference's demo.py:137-326
(= test_net.py:131-331): network forward, decode, per-class NMS, border inference, 4-DoF box solve,
dense alignment, 3-DoF rectification.  Device work goes through the HIP library; the two scipy
solvers and `infer_boundary` are host code, as in the reference (see model/utils/box_estimator.py
for why they cannot be anything else and still return the reference's boxes)."""
import math as m

import numpy as np
import torch

from . import postprocess
from .model.dense_align.dense_align import align_parallel
from .model.utils import box_estimator, kitti_utils
from .model.utils.config import cfg


def detect_3d(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh=0.05, class_index=1,
              dense_align=True):
    """Returns a list of dicts (one per solved object, descending score):
    box_left (4), box_right (4), score, dim (w,h,l), alpha, xyz (3), theta, aligned (bool)."""
    with torch.no_grad():
        out = model(im_left_data, im_right_data, im_info)
        det = postprocess.decode_detections(*out[:8], im_info)
        cls = postprocess.class_detections(det, class_index, eval_thresh, cfg.TEST.NMS)
    dets_left = cls['dets_left'].cpu().numpy()
    if dets_left.shape[0] == 0:
        return []
    dets_right = cls['dets_right'].cpu().numpy()
    dim_orien = cls['dim_orien'].cpu().numpy()
    kpts = cls['kpts'].cpu().numpy().copy()
    # demo.py:259-265: replace the regressed borders when they are narrower than half the inferred ones
    inferred = kitti_utils.infer_boundary(im_shape, dets_left)
    for i in range(dets_left.shape[0]):
        if kpts[i, 4] - kpts[i, 3] < 0.5 * (inferred[i, 1] - inferred[i, 0]):
            kpts[i, 3:5] = inferred[i]
    solved = []
    for i in range(dets_left.shape[0]):                       # demo.py:282-302
        if not dets_left[i, -1] > eval_thresh:
            continue
        dim = dim_orien[i, 0:3]
        alpha = m.atan2(dim_orien[i, 3], dim_orien[i, 4])
        status, state = box_estimator.solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, dets_left[i, 0:4],
                                                                 dets_right[i, 0:4], kpts[i])
        if status > 0:
            solved.append({'box_left': dets_left[i, 0:4].copy(), 'box_right': dets_right[i, 0:4].copy(),
                           'score': float(dets_left[i, 4]), 'dim': dim.astype(np.float64), 'alpha': alpha,
                           'xyz': np.array(state[0:3], dtype=np.float64), 'theta': float(state[3]),
                           'kpts': kpts[i].copy(), 'aligned': False,
                           'xyz_init': np.array(state[0:3], dtype=np.float64)})   # 4-DoF solve, before alignment
    if not solved or not dense_align:
        return solved
    dev = im_left_data.device
    f32 = lambda rows: torch.tensor(np.asarray(rows), dtype=torch.float32, device=dev)
    boxes = f32([o['box_left'] for o in solved])
    kp = f32([o['kpts'] for o in solved])
    poses = f32([[o['xyz'][0], o['xyz'][1], o['xyz'][2], o['dim'][0], o['dim'][1], o['dim'][2], o['theta']]
                 for o in solved])
    succ, dis_final = align_parallel(calib, float(im_info.view(-1, 3)[0, 2]), im_left_data, im_right_data, boxes, kp,
                                     poses)                   # demo.py:306-308
    succ, dis_final = succ.cpu().numpy(), dis_final.cpu().numpy()
    for i, o in enumerate(solved):                            # demo.py:311-319
        if succ[i] > 0:
            state, z = box_estimator.solve_x_y_theta_from_kpt(im_shape, calib, o['alpha'], o['dim'], o['box_left'],
                                                              float(dis_final[i]), o['kpts'])
            o['xyz'] = np.array([state[0], state[1], z], dtype=np.float64)
            o['theta'] = float(state[2])
            o['aligned'] = True
            o['disparity'] = float(dis_final[i])
    return solved


def write_kitti_results(result_dir, file_number, calib, objects):
    """test_net.py:329-330: one KITTI line per object."""
    for o in objects:
        kitti_utils.write_detection_results(result_dir, file_number, calib, o['box_left'], o['xyz'], o['dim'],
                                            o['theta'], o['score'])
