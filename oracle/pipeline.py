"""CPU oracle of the end-to-end flow demo.py:137-326 (TEST INFRASTRUCTURE ONLY): oracle forward,
decode, per-class NMS, `infer_boundary` (kitti_utils.py:398-437, restated below), the scipy solvers
(oracle/box_estimator.py) and dense alignment (oracle/dense_align.py)."""
import math

import numpy as np
import torch

from . import box_estimator, config as C, dense_align, net, postprocess


def infer_boundary(im_shape, boxes):
    """kitti_utils.py:398-437."""
    boxes = np.asarray(boxes, dtype=np.float64)
    out = np.zeros((boxes.shape[0], 2), np.float32)
    line = np.zeros(im_shape[1] + 1)
    span = lambda b: range(int(b[0]), int(b[2]) + 1)
    for b in boxes:
        d = 1050.0 / b[3]
        for c in span(b):
            if line[c] == 0.0:
                line[c] = d
            elif d < line[c]:
                line[c] = (d + line[c]) / 2.0
    for i, b in enumerate(boxes):
        d = 1050.0 / b[3]
        out[i] = (b[0], b[2])
        lv, rv = not line[int(b[0])] < d, not line[int(b[2])] < d
        if not lv and not rv:
            out[i, 1] = b[0]
        for c in span(b):
            if lv and line[c] >= d:
                out[i, 1] = c
            elif rv and line[c] < d:
                out[i, 0] = c
    return out


def detect_3d(sd, im_left, im_right, im_info, calib, im_shape, eval_thresh=C.EVAL_THRESH, dense=True):
    out = net.forward(sd, im_left, im_right, im_info)
    det = postprocess.decode_detections(out, im_info)
    cls = postprocess.class_detections(det, 1, eval_thresh, C.TEST_NMS)
    dl, dr = cls['dets_left'].numpy(), cls['dets_right'].numpy()
    if dl.shape[0] == 0:
        return []
    do, kp = cls['dim_orien'].numpy(), cls['kpts'].numpy().copy()
    inf = infer_boundary(im_shape, dl)
    for i in range(dl.shape[0]):
        if kp[i, 4] - kp[i, 3] < 0.5 * (inf[i, 1] - inf[i, 0]):
            kp[i, 3:5] = inf[i]
    objs = []
    for i in range(dl.shape[0]):
        if not dl[i, -1] > eval_thresh:
            continue
        alpha = math.atan2(do[i, 3], do[i, 4])
        st, state = box_estimator.solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, do[i, 0:3], dl[i, 0:4], dr[i, 0:4], kp[i])
        if st > 0:
            objs.append({'box_left': dl[i, 0:4].copy(), 'box_right': dr[i, 0:4].copy(), 'score': float(dl[i, 4]),
                         'dim': do[i, 0:3].astype(np.float64), 'alpha': alpha, 'xyz': np.array(state[0:3]),
                         'theta': float(state[3]), 'kpts': kp[i].copy(), 'aligned': False,
                         'xyz_init': np.array(state[0:3])})
    if not objs or not dense:
        return objs
    t = lambda rows: torch.tensor(np.asarray(rows), dtype=torch.float32)
    poses = t([[o['xyz'][0], o['xyz'][1], o['xyz'][2], o['dim'][0], o['dim'][1], o['dim'][2], o['theta']] for o in objs])
    succ, dis = dense_align.align_parallel(calib, float(im_info[0, 2]), im_left, im_right,
                                           t([o['box_left'] for o in objs]), t([o['kpts'] for o in objs]), poses)
    for i, o in enumerate(objs):
        if succ[i] > 0:
            state, z = box_estimator.solve_x_y_theta_from_kpt(im_shape, calib, o['alpha'], o['dim'], o['box_left'],
                                                              float(dis[i]), o['kpts'])
            o['xyz'] = np.array([state[0], state[1], z])
            o['theta'] = float(state[2])
            o['aligned'] = True
            o['disparity'] = float(dis[i])
    return objs
