"""`_Stereo_RPN` - reference lib/model/rpn/stereo_rpn.py:22-138 (inference branch)."""
import torch
import torch.nn as nn

from ... import _lib, engine
from ..utils.config import cfg
from .proposal_layer import _ProposalLayer


class _Stereo_RPN(nn.Module):
    """Stereo region proposal network.

    Parameter containers keep the reference's names (RPN_Conv, RPN_cls_score,
    RPN_bbox_pred_left_right: stereo_rpn.py:32-40) so checkpoints load unchanged; the
    arithmetic runs in the HIP library:
      * RPN_Conv 3x3 on left and right maps writes straight into the [left 512 | right 512]
        channel halves of one NHWC buffer (the torch.cat of :77-78 never materialises);
      * RPN_cls_score and RPN_bbox_pred_left_right are one fused 24-channel 1x1 GEMM;
      * srcnn_rpn_score applies the (c, c+3) pair softmax and the NHWC flatten (:81-91);
      * _ProposalLayer is a single native call.
    """

    def __init__(self, din):
        super(_Stereo_RPN, self).__init__()
        self.din = din
        self.anchor_ratios = cfg.ANCHOR_RATIOS
        self.feat_stride = cfg.FEAT_STRIDE[0]
        self.RPN_Conv = nn.Conv2d(self.din, 512, 3, 1, 1, bias=True)
        self.nc_score_out = 1 * len(self.anchor_ratios) * 2
        self.RPN_cls_score = nn.Conv2d(512 * 2, self.nc_score_out, 1, 1, 0)
        self.nc_bbox_out = 1 * len(self.anchor_ratios) * 6
        self.RPN_bbox_pred_left_right = nn.Conv2d(512 * 2, self.nc_bbox_out, 1, 1, 0)
        self.RPN_proposal = _ProposalLayer(self.feat_stride, self.anchor_ratios)
        self.rpn_loss_cls = 0
        self.rpn_loss_box_left_right = 0
        self._prepared = None

    def prepare(self, device):
        """Re-layout the weights for the conv engine (cached until parameters change)."""
        key = (str(device), self.RPN_Conv.weight._version, self.RPN_cls_score.weight._version,
               self.RPN_bbox_pred_left_right.weight._version, self.RPN_Conv.weight.data_ptr())
        if self._prepared is None or self._prepared[0] != key:
            conv = engine.prep_conv(self.RPN_Conv.weight.detach().cpu(), self.RPN_Conv.bias.detach().cpu(), 1, 1, True,
                                    device=device)
            w = torch.cat((self.RPN_cls_score.weight.detach().cpu(), self.RPN_bbox_pred_left_right.weight.detach().cpu()), 0)
            b = torch.cat((self.RPN_cls_score.bias.detach().cpu(), self.RPN_bbox_pred_left_right.bias.detach().cpu()), 0)
            head = engine.prep_conv(w, b, 1, 0, False, device=device)
            self._prepared = (key, conv, head)
        return self._prepared[1], self._prepared[2]

    def forward_nhwc(self, feats, shapes, B, im_info, bufs=None):
        """feats[l]: NHWC buffer (2B, H_l, W_l, din) holding the B left images then the B right
        images.  Returns (rois_left, rois_right, probs, deltas)."""
        dev = feats[0].device
        conv, head = self.prepare(dev)
        A = sum(3 * h * w for h, w in shapes)
        if bufs is None:
            bufs = {}
        probs = bufs.get('probs')
        if probs is None:
            probs = torch.empty((B, A, 2), device=dev)
            deltas = torch.empty((B, A, 6), device=dev)
        else:
            deltas = bufs['deltas']
        L = _lib.lib()
        off = 0
        for l, (h, w) in enumerate(shapes):
            cat = bufs.get('rpn_cat%d' % l)
            if cat is None:
                cat = torch.empty((B, h, w, 1024), device=dev)
            hd = bufs.get('rpn_head%d' % l)
            if hd is None:
                hd = torch.empty((B, h, w, 24), device=dev)
            per_img = h * w * self.din
            engine.conv2d(conv, feats[l], B, h, w, cat, h, w, y_cstride=1024, y_coffset=0)
            engine.conv2d(conv, feats[l], B, h, w, cat, h, w, y_cstride=1024, y_coffset=512,
                          x_offset_elems=B * per_img)
            engine.conv2d(head, cat, B, h, w, hd, h, w)
            _lib.check(L.srcnn_rpn_score(hd.data_ptr(), B, h * w, 24, probs.data_ptr(), deltas.data_ptr(), off, A,
                                         _lib.stream()), "srcnn_rpn_score")
            off += 3 * h * w
        out = None
        if 'rois_left' in bufs:
            out = (bufs['rois_left'], bufs['rois_right'], bufs['num_valid'])
        rois_l, rois_r = self.RPN_proposal.run(probs, deltas, im_info, [list(s) for s in shapes],
                                               cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N,
                                               cfg.TEST.RPN_NMS_THRESH, out=out)
        return rois_l, rois_r, probs, deltas

    def forward(self, rpn_feature_maps_left, rpn_feature_maps_right, im_info,
                gt_boxes_left=None, gt_boxes_right=None, gt_boxes_merge=None, num_boxes=None):
        """Reference signature (stereo_rpn.py:62-63): NCHW feature map lists in,
        (rois_left, rois_right, rpn_loss_cls, rpn_loss_box_left_right) out (losses are 0 in eval)."""
        if self.training:
            raise NotImplementedError("training branch (anchor targets / losses) is out of scope")
        B = int(rpn_feature_maps_left[0].shape[0])
        feats, shapes = [], []
        for fl, fr in zip(rpn_feature_maps_left, rpn_feature_maps_right):
            both = torch.cat((fl, fr), 0).contiguous().float()
            feats.append(engine.nchw_to_nhwc(both))
            shapes.append((int(fl.shape[2]), int(fl.shape[3])))
        rois_l, rois_r, _, _ = self.forward_nhwc(feats, shapes, B, im_info)
        self.rpn_loss_cls = 0
        self.rpn_loss_box_left_right = 0
        return rois_l, rois_r, self.rpn_loss_cls, self.rpn_loss_box_left_right
