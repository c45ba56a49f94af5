"""CPU oracle: FPN anchors, box decode/clip and the stereo proposal layer
(TEST INFRASTRUCTURE ONLY).

Reference lines followed (relative to /root/reference/lib/model/rpn):
  anchors        generate_anchors.py:112-173 (float64 numpy, level->row->col->ratio)
  decode / clip  bbox_transform.py:79-104, 177-185
  proposal layer proposal_layer.py:60-145
Sort: the reference's torch.sort tie order is unspecified (proposal_layer.py:96);
the oracle and the HIP path both use a STABLE descending sort (ties -> lower index).
Parity status: PINNED -- anchors equal generate_anchors_all_pyramids (sha256 over all 298 476 rows), decode/clip equal
bbox_transform_inv / clip_boxes bit for bit, and the layer as a whole is exercised by the network goldens
(tests/test_reference_golden.py).
"""
import numpy as np
import torch

from . import config as C
from . import ops


def anchors_single_level(scale, ratios, shape, feature_stride, anchor_stride=1):
    """generate_anchors.py:112-154 (restated with broadcasting instead of meshgrids)."""
    ratios = np.asarray(ratios, np.float64)
    heights = scale / np.sqrt(ratios)
    widths = scale * np.sqrt(ratios)
    ys = np.arange(0, shape[0], anchor_stride, dtype=np.float64) * feature_stride
    xs = np.arange(0, shape[1], anchor_stride, dtype=np.float64) * feature_stride
    cx = np.broadcast_to(xs[None, :, None], (len(ys), len(xs), len(ratios)))
    cy = np.broadcast_to(ys[:, None, None], (len(ys), len(xs), len(ratios)))
    w = np.broadcast_to(widths[None, None, :], cx.shape)
    h = np.broadcast_to(heights[None, None, :], cx.shape)
    boxes = np.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], axis=-1)
    return boxes.reshape(-1, 4)


def anchors_all_levels(feat_shapes, scales=C.FPN_ANCHOR_SCALES, ratios=C.ANCHOR_RATIOS,
                       strides=C.FPN_FEAT_STRIDES, anchor_stride=C.FPN_ANCHOR_STRIDE):
    """generate_anchors.py:157-173 -> (A, 4) float64."""
    return np.concatenate([anchors_single_level(scales[i], ratios, feat_shapes[i], strides[i], anchor_stride)
                           for i in range(len(feat_shapes))], axis=0)


def decode_boxes(boxes, deltas):
    """bbox_transform_inv, 3-D branch (bbox_transform.py:80-104). boxes/deltas (B, A, 4) float32."""
    widths = boxes[:, :, 2] - boxes[:, :, 0] + 1.0
    heights = boxes[:, :, 3] - boxes[:, :, 1] + 1.0
    ctr_x = boxes[:, :, 0] + 0.5 * widths
    ctr_y = boxes[:, :, 1] + 0.5 * heights
    dx, dy, dw, dh = deltas[:, :, 0::4], deltas[:, :, 1::4], deltas[:, :, 2::4], deltas[:, :, 3::4]
    pcx = dx * widths.unsqueeze(2) + ctr_x.unsqueeze(2)
    pcy = dy * heights.unsqueeze(2) + ctr_y.unsqueeze(2)
    pw = torch.exp(dw) * widths.unsqueeze(2)
    ph = torch.exp(dh) * heights.unsqueeze(2)
    out = deltas.clone()
    out[:, :, 0::4] = pcx - 0.5 * pw
    out[:, :, 1::4] = pcy - 0.5 * ph
    out[:, :, 2::4] = pcx + 0.5 * pw
    out[:, :, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_info):
    """bbox_transform.py:177-185: clamp to [0, W-1] / [0, H-1]."""
    for i in range(boxes.shape[0]):
        wmax = float(im_info[i, 1]) - 1
        hmax = float(im_info[i, 0]) - 1
        boxes[i, :, 0::4].clamp_(0, wmax)
        boxes[i, :, 1::4].clamp_(0, hmax)
        boxes[i, :, 2::4].clamp_(0, wmax)
        boxes[i, :, 3::4].clamp_(0, hmax)
    return boxes


def proposal_layer(probs, deltas, im_info, feat_shapes,
                   pre_nms_top_n=C.RPN_PRE_NMS_TOP_N, post_nms_top_n=C.RPN_POST_NMS_TOP_N,
                   nms_thresh=C.RPN_NMS_THRESH, order=None):
    """proposal_layer.py:42-145 (TEST cfg).  probs (B,A,2), deltas (B,A,6) ->
    rois_left, rois_right (B, post, 5) and a dict of intermediates for stage tests.
    order: (B, >= pre_nms_top_n) anchor indices to use INSTEAD of the stable sort -- the order a recorded reference run's
    (unstable) torch.sort actually produced (tests/tie_audit.py)."""
    scores = probs[:, :, 1]
    d_left = deltas[:, :, :4].clone()
    d_right = deltas[:, :, :4].clone()
    d_right[:, :, 0] = deltas[:, :, 4]
    d_right[:, :, 2] = deltas[:, :, 5]
    bsz = deltas.shape[0]
    anchors = torch.from_numpy(anchors_all_levels(feat_shapes)).to(scores.dtype)   # float64 -> float32
    anchors = anchors.view(1, -1, 4).expand(bsz, -1, 4)
    prop_l = clip_boxes(decode_boxes(anchors, d_left), im_info)
    prop_r = clip_boxes(decode_boxes(anchors, d_right), im_info)
    if order is None:
        order = torch.sort(scores, dim=1, descending=True, stable=True)[1]
    out_l = scores.new_zeros(bsz, post_nms_top_n, 5)
    out_r = scores.new_zeros(bsz, post_nms_top_n, 5)
    extra = {'order': [], 'keep_left': [], 'keep_right': [], 'keep': [], 'dets_left': [], 'dets_right': []}
    for i in range(bsz):
        o = order[i]
        if 0 < pre_nms_top_n < scores.numel():      # (sic) numel of the whole batch, proposal_layer.py:111
            o = o[:pre_nms_top_n]
        pl, pr = prop_l[i][o], prop_r[i][o]
        sc = scores[i][o].view(-1, 1)
        dets_l = torch.cat((pl, sc), 1)
        dets_r = torch.cat((pr, sc), 1)
        keep_l = ops.nms(dets_l.numpy(), nms_thresh)
        keep_r = ops.nms(dets_r.numpy(), nms_thresh)
        keep = np.intersect1d(keep_l, keep_r)          # ascending index == descending score
        if post_nms_top_n > 0:
            keep = keep[:post_nms_top_n]
        kt = torch.from_numpy(keep.astype(np.int64))
        n = kt.numel()
        out_l[i, :, 0] = i
        out_l[i, :n, 1:] = pl[kt]
        out_r[i, :, 0] = i
        out_r[i, :n, 1:] = pr[kt]
        extra['order'].append(o)
        extra['keep_left'].append(keep_l)
        extra['keep_right'].append(keep_r)
        extra['keep'].append(keep)
        extra['dets_left'].append(dets_l)
        extra['dets_right'].append(dets_r)
    return out_l, out_r, extra
