"""`_ProposalLayer` - reference lib/model/rpn/proposal_layer.py:27-145, one native call."""
import ctypes

import torch
import torch.nn as nn

from ... import _lib
from ..utils.config import cfg


class _ProposalLayer(nn.Module):
    """Stereo proposals from RPN scores/deltas.

    forward(input) with input = (rpn_cls_prob (B,A,2), rpn_bbox_pred_left_right (B,A,6),
    im_info (B,3), cfg_key, feat_shapes [[H,W]...]) -> (rois_left, rois_right) each
    (B, post_nms_topN, 5) [batch_idx, x1, y1, x2, y2], zero padded (proposal_layer.py:98-99,139-143).
    Anchors, decode, clip, stable top-K sort, both NMS passes, the sorted intersection and
    the padding all run on the device inside srcnn_proposal_layer (no host round trip).
    """

    def __init__(self, feat_stride, ratios):
        super(_ProposalLayer, self).__init__()
        self._anchor_ratios = ratios
        self._feat_stride = feat_stride
        if list(ratios) != [0.5, 1, 2] or list(cfg.FPN_ANCHOR_SCALES) != [32, 64, 128, 256, 512] \
                or list(cfg.FPN_FEAT_STRIDES) != [4, 8, 16, 32, 64] or cfg.FPN_ANCHOR_STRIDE != 1:
            raise NotImplementedError("the HIP proposal kernel is specialised to the reference's FPN anchor config")
        self.last_num_valid = None

    def forward(self, input):
        probs, deltas, im_info, cfg_key, feat_shapes = input
        if cfg_key != 'TEST':
            raise NotImplementedError("training-time proposals are out of scope (inference path only)")
        return self.run(probs, deltas, im_info, feat_shapes, cfg[cfg_key].RPN_PRE_NMS_TOP_N,
                        cfg[cfg_key].RPN_POST_NMS_TOP_N, cfg[cfg_key].RPN_NMS_THRESH)

    def run(self, probs, deltas, im_info, feat_shapes, pre_nms, post_nms, nms_thresh, out=None):
        assert probs.is_cuda and deltas.is_cuda, "device tensors required"
        probs = probs.contiguous()
        deltas = deltas.contiguous()
        im_info = im_info.to(device=probs.device, dtype=torch.float32).contiguous()
        B, A = int(probs.shape[0]), int(probs.shape[1])
        nl = len(feat_shapes)
        hw = (ctypes.c_int * (2 * nl))(*[int(v) for s in feat_shapes for v in s])
        if out is None:
            rois_l = torch.empty((B, post_nms, 5), dtype=torch.float32, device=probs.device)
            rois_r = torch.empty((B, post_nms, 5), dtype=torch.float32, device=probs.device)
            num_valid = torch.empty((B,), dtype=torch.int32, device=probs.device)
        else:
            rois_l, rois_r, num_valid = out
        L = _lib.lib()
        ws = _lib.workspace(L.srcnn_proposal_workspace_bytes(B, A, pre_nms, post_nms), probs.device, "proposal")
        _lib.check(L.srcnn_proposal_layer(probs.data_ptr(), deltas.data_ptr(), B, A, hw, nl, im_info.data_ptr(),
                                          int(pre_nms), int(post_nms), float(nms_thresh), rois_l.data_ptr(),
                                          rois_r.data_ptr(), num_valid.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _lib.stream()), "srcnn_proposal_layer")
        self.last_num_valid = num_valid
        return rois_l, rois_r
