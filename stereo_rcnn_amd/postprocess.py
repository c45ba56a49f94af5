"""Detection decode and per-class NMS on the device.

The reference does this inline in its scripts (demo.py:144-257 == test_net.py:138-257) with a
300-iteration Python loop that issues ~2400 tiny GPU ops (demo.py:152-161).  Here it is two
native calls: srcnn_decode_detections and srcnn_class_nms.
"""
import torch

from . import _lib
from .model.utils.config import cfg


def decode_detections(rois_left, rois_right, cls_prob, bbox_pred, dim_orien_pred, kpts_prob,
                      left_border_prob, right_border_prob, im_info):
    """Network outputs (batch of 1, as the reference scripts assume: demo.py:153,212) ->
    dict(scores (n, n_cls), boxes_left/right (n, 4*n_cls) in ORIGINAL image pixels,
    dim_orien (n, 5*n_cls), kpts (n, 5) = (u, type, prob, left_border, right_border))."""
    assert rois_left.shape[0] == 1, "decode works on one image at a time (as the reference does)"
    dev = rois_left.device
    n = int(rois_left.shape[1])
    n_cls = int(cls_prob.shape[2])
    G = cfg.KPTS_GRID
    f = lambda t: t.contiguous().float()
    rl, rr, bp, dp = f(rois_left[0]), f(rois_right[0]), f(bbox_pred[0]), f(dim_orien_pred[0])
    if kpts_prob is None:      # lazy keypoint head (pipeline): the `kpts` rows of the kept detections are filled in after class NMS
        kp, lp, rp = _zero_probs(n, G, dev)
    else:
        kp, lp, rp = f(kpts_prob), f(left_border_prob), f(right_border_prob)
    info = f(im_info.view(-1, 3)[0].to(dev))
    boxes_l = torch.empty((n, 4 * n_cls), device=dev)
    boxes_r = torch.empty((n, 4 * n_cls), device=dev)
    dim = torch.empty((n, 5 * n_cls), device=dev)
    kpts = torch.empty((n, 5), device=dev)
    _lib.check(_lib.lib().srcnn_decode_detections(rl.data_ptr(), rr.data_ptr(), bp.data_ptr(), dp.data_ptr(),
                                                  kp.data_ptr(), lp.data_ptr(), rp.data_ptr(), info.data_ptr(),
                                                  n, n_cls, G, boxes_l.data_ptr(), boxes_r.data_ptr(),
                                                  dim.data_ptr(), kpts.data_ptr(), _lib.stream()),
               "srcnn_decode_detections")
    return {'scores': f(cls_prob[0]), 'boxes_left': boxes_l, 'boxes_right': boxes_r, 'dim_orien': dim, 'kpts': kpts}


_zeros = {}


def _zero_probs(n, G, dev):
    key = (n, G, str(dev))
    if key not in _zeros:
        _zeros[key] = (torch.zeros((n, 4 * G), device=dev), torch.zeros((n, G), device=dev), torch.zeros((n, G), device=dev))
    return _zeros[key]


def class_nms_device(det, j=1, thresh=0.05, nms_thresh=None):
    """Device-only part of demo.py:231-257: returns (keep_idx (n) int32, -1 padded, descending score
    order; num (1) int32) without any host synchronisation."""
    if nms_thresh is None:
        nms_thresh = cfg.TEST.NMS
    scores = det['scores']
    n, n_cls = int(scores.shape[0]), int(scores.shape[1])
    dev = scores.device
    keep_idx = torch.empty((n,), dtype=torch.int32, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = _lib.workspace(L.srcnn_class_nms_workspace_bytes(n), dev, "class_nms")
    _lib.check(L.srcnn_class_nms(scores.data_ptr(), n, n_cls, j, det['boxes_left'].data_ptr(), float(thresh),
                                 float(nms_thresh), keep_idx.data_ptr(), num.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _lib.stream()), "srcnn_class_nms")
    return keep_idx, num


def class_detections(det, j=1, thresh=0.05, nms_thresh=None):
    """demo.py:231-257 for class j: score > thresh, stable descending sort, NMS on the LEFT boxes,
    gather.  Returns dict(dets_left (k,5), dets_right (k,5), dim_orien (k,5), kpts (k,5), keep_idx (k,))
    where keep_idx are row indices into the 300 rois, in descending score order."""
    keep_idx, num = class_nms_device(det, j, thresh, nms_thresh)
    scores = det['scores']
    k = int(num[0])                                 # the one host sync, same place as nms_gpu.py:11
    idx = keep_idx[:k].long()
    sc = scores[idx, j].unsqueeze(1)
    return {'dets_left': torch.cat((det['boxes_left'][idx, 4 * j:4 * j + 4], sc), 1),
            'dets_right': torch.cat((det['boxes_right'][idx, 4 * j:4 * j + 4], sc), 1),
            'dim_orien': det['dim_orien'][idx, 5 * j:5 * j + 5], 'kpts': det['kpts'][idx], 'keep_idx': keep_idx[:k]}
