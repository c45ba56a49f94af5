"""CPU oracle of the Stereo R-CNN inference forward (TEST INFRASTRUCTURE ONLY).

Functional torch-CPU fp32 restatement of `_StereoRCNN.forward` in eval mode,
driven directly by a reference-schema state_dict.  Reference lines followed
(paths relative to /root/reference/lib/model):
  trunk            stereo_rcnn/resnet.py:66-102 (Bottleneck, stride on conv1), :105-146, :236-240
  FPN              stereo_rcnn/stereo_rcnn.py:91-108 (_upsample_add), :161-168
  stereo RPN head  rpn/stereo_rpn.py:51-60 (reshape), :73-95
  proposals        oracle/proposal.py
  pyramid ROI feat stereo_rcnn/stereo_rcnn.py:110-139
  heads            stereo_rcnn/resnet.py:256-286,345-348; stereo_rcnn/stereo_rcnn.py:248-271
PyTorch-0.3 semantics that must be spelled out under torch 2.x: bilinear
`F.upsample` == align_corners=True (stereo_rcnn.py:108); torch.round is
half-away-from-zero (stereo_rcnn.py:117).

Parity status: PINNED -- `forward` reproduces, bit for bit on every coordinate-matched proposal, the outputs of the
reference's own `_StereoRCNN.forward` run in the build container with the same seeded weights and inputs
(tests/golden/reference_net_*.npz, tests/test_reference_golden.py; shims: tests/golden/reference_shims.py).
"""
import torch
import torch.nn.functional as F

from . import config as C
from . import ops
from . import proposal

BN_EPS = 1e-5


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, BN_EPS)


def _bottleneck(sd, p, x, stride, has_down):
    out = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], None, stride)))
    out = F.relu(_bn(sd, p + '.bn2', F.conv2d(out, sd[p + '.conv2.weight'], None, 1, 1)))
    out = _bn(sd, p + '.bn3', F.conv2d(out, sd[p + '.conv3.weight']))
    res = x
    if has_down:
        res = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride))
    return F.relu(out + res)


def count_blocks(sd):
    layers = []
    for li in (1, 2, 3, 4):
        n = 0
        while 'RCNN_layer%d.0.%d.conv1.weight' % (li, n) in sd:
            n += 1
        layers.append(n)
    return tuple(layers)


def trunk(sd, im):
    """RCNN_layer0..4 -> (c2, c3, c4, c5)."""
    x = F.conv2d(im, sd['RCNN_layer0.0.weight'], None, 2, 3)
    x = F.relu(_bn(sd, 'RCNN_layer0.1', x))
    x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)          # resnet.py:113
    feats = []
    for li, nblk in zip((1, 2, 3, 4), count_blocks(sd)):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            x = _bottleneck(sd, 'RCNN_layer%d.0.%d' % (li, b), x, stride, b == 0)
        feats.append(x)
    return feats


def _convb(sd, name, x, stride=1, pad=0):
    return F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], stride, pad)


def _upsample_add(x, y):
    return F.interpolate(x, size=y.shape[2:], mode='bilinear', align_corners=True) + y


def fpn(sd, c2, c3, c4, c5):
    """-> [p2, p3, p4, p5, p6] (stereo_rcnn.py:161-168)."""
    p5 = _convb(sd, 'RCNN_toplayer', c5)
    p4 = _convb(sd, 'RCNN_smooth1', _upsample_add(p5, _convb(sd, 'RCNN_latlayer1', c4)), 1, 1)
    p3 = _convb(sd, 'RCNN_smooth2', _upsample_add(p4, _convb(sd, 'RCNN_latlayer2', c3)), 1, 1)
    p2 = _convb(sd, 'RCNN_smooth3', _upsample_add(p3, _convb(sd, 'RCNN_latlayer3', c2)), 1, 1)
    p6 = p5[:, :, ::2, ::2]                              # MaxPool2d(1, stride=2) (stereo_rcnn.py:39,168)
    return [p2, p3, p4, p5, p6]


def rpn_head(sd, feats_l, feats_r):
    """-> probs (B, A, 2), deltas (B, A, 6), shapes [[H,W]...] (stereo_rpn.py:73-95)."""
    probs, deltas, shapes = [], [], []
    for fl, fr in zip(feats_l, feats_r):
        b = fl.shape[0]
        x = torch.cat((F.relu(_convb(sd, 'RCNN_rpn.RPN_Conv', fl, 1, 1)),
                       F.relu(_convb(sd, 'RCNN_rpn.RPN_Conv', fr, 1, 1))), 1)
        score = _convb(sd, 'RCNN_rpn.RPN_cls_score', x)                   # (B, 6, H, W)
        h, w = score.shape[2:]
        prob = F.softmax(score.view(b, 2, 3 * h, w), 1).view(b, 6, h, w)  # the (c, c+3) pairing quirk
        delta = _convb(sd, 'RCNN_rpn.RPN_bbox_pred_left_right', x)        # (B, 18, H, W)
        shapes.append([h, w])
        probs.append(prob.permute(0, 2, 3, 1).contiguous().view(b, -1, 2))
        deltas.append(delta.permute(0, 2, 3, 1).contiguous().view(b, -1, 6))
    return torch.cat(probs, 1), torch.cat(deltas, 1), shapes


def round_half_away(x):
    return torch.sign(x) * torch.floor(torch.abs(x) + 0.5)


def roi_levels(rois):
    """stereo_rcnn.py:113-119.  rois (n,5) -> float levels in {2,3,4,5}."""
    h = rois[:, 4] - rois[:, 2] + 1
    w = rois[:, 3] - rois[:, 1] + 1
    lvl = round_half_away(torch.log(torch.sqrt(h * w) / 224.0) + 4)
    return lvl.clamp(2, 5)


def pyramid_roi_feat(feat_maps, rois, im_info, kpts=False):
    """stereo_rcnn.py:110-139 with the native op replaced by the C oracle."""
    lvl = roi_levels(rois)
    a = C.POOLING_SIZE * 2 if kpts else C.POOLING_SIZE
    out = torch.zeros(rois.shape[0], feat_maps[0].shape[1], a, a)
    for i, l in enumerate(range(2, 6)):
        idx = (lvl == l).nonzero().view(-1)
        if idx.numel() == 0:
            continue
        scale = feat_maps[i].shape[2] / float(im_info[0][0])       # python float, stereo_rcnn.py:128
        pooled = ops.roi_align_avg(feat_maps[i].numpy(), rois[idx].numpy(), a, a, scale)
        out[idx] = torch.from_numpy(pooled)                        # == cat + sort-permute of :134-137
    return out


def box_head(sd, feat):
    """_head_to_tail + the three FC heads (resnet.py:345-348, stereo_rcnn.py:252-257)."""
    x = F.relu(_convb(sd, 'RCNN_top.0', feat, C.POOLING_SIZE))
    x = F.relu(_convb(sd, 'RCNN_top.3', x))
    fc7 = x.mean(3).mean(2)
    bbox_pred = F.linear(fc7, sd['RCNN_bbox_pred.weight'], sd['RCNN_bbox_pred.bias'])
    dim_orien = F.linear(fc7, sd['RCNN_dim_orien_pred.weight'], sd['RCNN_dim_orien_pred.bias'])
    cls_score = F.linear(fc7, sd['RCNN_cls_score.weight'], sd['RCNN_cls_score.bias'])
    return bbox_pred, dim_orien, F.softmax(cls_score, 1)


def kpts_head(sd, feat):
    """RCNN_kpts + kpts_class + sum over H + softmaxes (stereo_rcnn.py:260-271)."""
    x = feat
    for i in (0, 2, 4, 6, 8, 10):
        x = F.relu(_convb(sd, 'RCNN_kpts.%d' % i, x, 1, 1))
    x = F.relu(F.conv_transpose2d(x, sd['RCNN_kpts.12.weight'], sd['RCNN_kpts.12.bias'], 2))
    allp = _convb(sd, 'kpts_class', x).sum(2)                    # (n, 6, 28)
    g = C.KPTS_GRID
    kpts_prob = F.softmax(allp[:, :4, :].contiguous().view(-1, 4 * g), 1)
    left_prob = F.softmax(allp[:, 4, :].contiguous().view(-1, g), 1)
    right_prob = F.softmax(allp[:, 5, :].contiguous().view(-1, g), 1)
    return kpts_prob, left_prob, right_prob


def forward(sd, im_left, im_right, im_info, keep=False):
    """Eval-mode `_StereoRCNN.forward` (stereo_rcnn.py:141-324).

    Returns a dict with the reference's 8 inference outputs (rois_left, rois_right,
    cls_prob, bbox_pred, dim_orien_pred, kpts_prob, left_border_prob,
    right_border_prob) plus, if keep=True, stage intermediates for stage-level tests.
    """
    with torch.no_grad():
        b = im_left.shape[0]
        cl = trunk(sd, im_left)
        pl = fpn(sd, *cl)
        cr = trunk(sd, im_right)
        pr = fpn(sd, *cr)
        probs, deltas, shapes = rpn_head(sd, pl, pr)
        rois_l, rois_r, extra = proposal.proposal_layer(probs, deltas, im_info, shapes)
        rl, rr = rois_l.view(-1, 5), rois_r.view(-1, 5)
        sem_l = _pyr_batched(pl[:4], rl, im_info)
        sem_r = _pyr_batched(pr[:4], rr, im_info)
        sem = torch.cat((sem_l, sem_r), 1)
        bbox_pred, dim_orien, cls_prob = box_head(sd, sem)
        dense = _pyr_batched(pl[:4], rl, im_info, kpts=True)
        kpts_prob, left_prob, right_prob = kpts_head(sd, dense)
        out = {
            'rois_left': rois_l, 'rois_right': rois_r,
            'cls_prob': cls_prob.view(b, -1, cls_prob.shape[1]),
            'bbox_pred': bbox_pred.view(b, -1, bbox_pred.shape[1]),
            'dim_orien_pred': dim_orien.view(b, -1, dim_orien.shape[1]),
            'kpts_prob': kpts_prob, 'left_border_prob': left_prob, 'right_border_prob': right_prob,
        }
        if keep:
            out.update({'c_left': cl, 'p_left': pl, 'c_right': cr, 'p_right': pr,
                        'rpn_probs': probs, 'rpn_deltas': deltas, 'rpn_shapes': shapes,
                        'proposal_extra': extra, 'sem_feat': sem, 'kpts_feat': dense})
        return out


def _pyr_batched(maps, rois, im_info, kpts=False):
    return pyramid_roi_feat(maps, rois, im_info, kpts)
