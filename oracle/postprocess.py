"""CPU oracle: detection decode and per-class filter + NMS (TEST INFRASTRUCTURE ONLY).

Reference lines followed (relative to /root/reference):
  de-interleave + denormalise + decode + clip + /scale   demo.py:144-218 (= test_net.py:138-212)
  keypoint / border decode                              lib/model/rpn/bbox_transform.py:133-155
  per-class threshold, sort, NMS, gather                demo.py:231-257
Parity status: PINNED -- demo.py is a script, so its decode block (:143-224) and per-class filter/sort/NMS block (:231-251)
are sliced out by content markers and exec'd on the reference network's outputs (tests/golden/make_reference_golden.py:
decode_golden); this file reproduces those numbers exactly (tests/test_reference_golden.py).
"""
import numpy as np
import torch

from . import config as C
from . import ops
from .proposal import decode_boxes, clip_boxes


def decode_detections(out, im_info, n_classes=2):
    """out: dict from oracle.net.forward (B == 1).  Returns dict of (300, .) tensors:
    scores (300, n_cls), boxes_left/right (300, 4*n_cls) in ORIGINAL image pixels,
    dim_orien (300, 5*n_cls), kpts (300, 5) = (u, type, prob, left_border, right_border)."""
    scores = out['cls_prob']
    boxes_l = out['rois_left'][:, :, 1:5]
    boxes_r = out['rois_right'][:, :, 1:5]
    bp = out['bbox_pred'][0]                                   # (300, 6*n_cls)
    n = bp.shape[0]
    v = bp.view(n, n_classes, 6)
    d_left = v[:, :, [0, 1, 2, 3]].reshape(-1, 4)               # demo.py:153-156
    d_right = v[:, :, [4, 1, 5, 3]].reshape(-1, 4)              # demo.py:158-161
    stds = torch.tensor(C.BBOX_NORMALIZE_STDS, dtype=torch.float32)
    means = torch.tensor(C.BBOX_NORMALIZE_MEANS, dtype=torch.float32)
    d_left = (d_left * stds + means).view(1, -1, 4 * n_classes)
    d_right = (d_right * stds + means).view(1, -1, 4 * n_classes)
    dim = out['dim_orien_pred'].reshape(-1, 5)
    dim = dim * torch.tensor(C.DIM_NORMALIZE_STDS, dtype=torch.float32) \
        + torch.tensor(C.DIM_NORMALIZE_MEANS, dtype=torch.float32)
    dim = dim.view(-1, 5 * n_classes)

    g = C.KPTS_GRID
    max_prob, kpts_delta = torch.max(out['kpts_prob'].view(-1, 4 * g), 1)
    left_delta = torch.max(out['left_border_prob'].view(-1, g), 1)[1]
    right_delta = torch.max(out['right_border_prob'].view(-1, g), 1)[1]

    pred_l = clip_boxes(decode_boxes(boxes_l, d_left), im_info)
    pred_r = clip_boxes(decode_boxes(boxes_r, d_right), im_info)
    widths = boxes_l[0, :, 2] - boxes_l[0, :, 0] + 1.0          # bbox_transform.py:135 (proposal width)
    x1 = boxes_l[0, :, 0]
    dk = kpts_delta.float()
    kpts_type = dk / g                                          # float division (bbox_transform.py:139)
    pred_kpts = (dk % g) * widths / g + x1
    pred_lb = left_delta.float() * widths / g + x1
    pred_rb = right_delta.float() * widths / g + x1
    s = im_info[0, 2]
    pred_l = pred_l / s
    pred_r = pred_r / s
    kpts = torch.stack((pred_kpts / s, kpts_type, max_prob, pred_lb / s, pred_rb / s), 1)
    return {'scores': scores[0], 'boxes_left': pred_l[0], 'boxes_right': pred_r[0],
            'dim_orien': dim, 'kpts': kpts}


def class_detections(det, j=1, thresh=C.EVAL_THRESH, nms_thresh=C.TEST_NMS):
    """demo.py:231-257 for class j.  Returns dict with dets_left/right (k,5), dim_orien (k,5),
    kpts (k,5) and the index chain (inds, order, keep) for bit-exact index tests."""
    scores = det['scores']
    inds = torch.nonzero(scores[:, j] > thresh).view(-1)
    if inds.numel() == 0:
        z = scores.new_zeros(0, 5)
        return {'dets_left': z, 'dets_right': z, 'dim_orien': z, 'kpts': z,
                'inds': inds, 'order': inds, 'keep': np.zeros((0,), np.int32)}
    cls_scores = scores[:, j][inds]
    order = torch.sort(cls_scores, dim=0, descending=True, stable=True)[1]
    bl = det['boxes_left'][inds][:, j * 4:(j + 1) * 4]
    br = det['boxes_right'][inds][:, j * 4:(j + 1) * 4]
    do = det['dim_orien'][inds][:, j * 5:(j + 1) * 5]
    kp = det['kpts'][inds]
    dl = torch.cat((bl, cls_scores.unsqueeze(1)), 1)[order]
    dr = torch.cat((br, cls_scores.unsqueeze(1)), 1)[order]
    do, kp = do[order], kp[order]
    keep = ops.nms(dl.numpy(), nms_thresh)
    kt = torch.from_numpy(keep.astype(np.int64))
    return {'dets_left': dl[kt], 'dets_right': dr[kt], 'dim_orien': do[kt], 'kpts': kp[kt],
            'inds': inds, 'order': order, 'keep': keep}
