"""Tie audit of the proposal stage (test infrastructure).

Two runs of the SAME proposal algorithm (proposal_layer.py:42-145) on inputs that differ by float rounding -- the reference
network's RPN outputs and the HIP network's -- can only produce different proposal sets through discrete decisions that came out
differently.  There are exactly three kinds of decision: (a) which anchors are among the 6000 best, (b) their order, (c) for a
pair of candidates, whether IoU > 0.7.  Everything downstream (greedy suppression, left/right intersection, the first 300) is a
deterministic function of those.  The audit enumerates EVERY decision that differs between the two runs and requires each to be a
near-tie of the REFERENCE run's own margin: |score - 6000th score|, |score_i - score_j|, |IoU - 0.7| within what the measured
input difference can move them.  No differing decision -> the two keep lists must be identical.  So 100 % of the reference's
proposals are either matched or explained by a listed tie; nothing is excused by a fraction.
"""
import numpy as np
import torch

from oracle import config as C
from oracle import proposal as oprop


def proposal_run(fg, deltas, im_info, shapes, order=None):
    """The reference's proposal layer (its bit-exact restatement oracle/proposal.py) on (A,) foreground scores and (A, 6) deltas.
    order: the candidate order a recorded run used (the reference's torch.sort is not stable: among EXACTLY equal scores -- a
    float32 softmax near 1.0 has thousands -- its order is whatever that torch build does); default: stable descending sort, the
    order the HIP path uses.  Returns rois and the intermediates the audit needs."""
    A = fg.shape[0]
    probs = torch.zeros((1, A, 2), dtype=torch.float32)
    probs[0, :, 1] = torch.as_tensor(fg)
    rl, rr, ex = oprop.proposal_layer(probs, torch.as_tensor(deltas).view(1, A, 6).float(), torch.as_tensor(im_info).view(1, 3).float(),
                                      [list(map(int, s)) for s in shapes],
                                      order=None if order is None else torch.as_tensor(np.asarray(order, np.int64)).view(1, -1))
    return {'rois_left': rl, 'rois_right': rr, 'order': ex['order'][0].numpy(), 'dets_left': ex['dets_left'][0].numpy(),
            'dets_right': ex['dets_right'][0].numpy(), 'keep_left': np.asarray(ex['keep_left'][0]), 'keep_right': np.asarray(ex['keep_right'][0]),
            'keep': np.asarray(ex['keep'][0]), 'fg': np.asarray(fg, np.float32)}


def reference_run_from_golden(g):
    """The reference run's proposal stage rebuilt from what the golden kept of it: all foreground scores, and the deltas of the
    7000 best anchors (the others cannot reach the NMS; their deltas are left zero)."""
    A = g['rpn_fg'].shape[0]
    deltas = np.zeros((A, 6), np.float32)
    deltas[g['rpn_top_idx']] = g['rpn_top_deltas']
    return proposal_run(g['rpn_fg'], deltas, g['im_info'], g['rpn_shapes'], order=g['rpn_order'])


def _iou_f32(a, b):
    """nms_cuda_kernel.cu:31-39 (devIoU) in float32, a (n, 4) against b (m, 4) -> (n, m)."""
    f = np.float32
    left = np.maximum(a[:, None, 0], b[None, :, 0]); right = np.minimum(a[:, None, 2], b[None, :, 2])
    top = np.maximum(a[:, None, 1], b[None, :, 1]); bottom = np.minimum(a[:, None, 3], b[None, :, 3])
    w = np.maximum(right - left + f(1), f(0)); h = np.maximum(bottom - top + f(1), f(0))
    inter = w * h
    sa = (a[:, 2] - a[:, 0] + f(1)) * (a[:, 3] - a[:, 1] + f(1))
    sb = (b[:, 2] - b[:, 0] + f(1)) * (b[:, 3] - b[:, 1] + f(1))
    return inter / (sa[:, None] + sb[None, :] - inter)


def audit(ref, hip, thresh=C.RPN_NMS_THRESH, chunk=400):
    """ref / hip: proposal_run() dicts.  Returns a report dict; report['unexplained'] lists every differing decision whose
    reference margin exceeds what the measured input difference explains (must be empty)."""
    s_ref, s_hip = ref['fg'], hip['fg']
    eps_s = float(np.abs(s_ref - s_hip).max())
    tol_s = 2.0 * eps_s + 1e-12
    rep = {'eps_score': eps_s, 'unexplained': []}
    o_ref, o_hip = ref['order'], hip['order']
    n = o_ref.shape[0]
    T = float(s_ref[o_ref[-1]])                                    # the 6000th score of the reference run
    in_ref, in_hip = set(o_ref.tolist()), set(o_hip.tolist())
    # (a) membership of the top-n
    only_ref, only_hip = sorted(in_ref - in_hip), sorted(in_hip - in_ref)
    rep['membership_flips'] = len(only_ref) + len(only_hip)
    for a in only_ref + only_hip:
        m = abs(float(s_ref[a]) - T)
        if m > tol_s:
            rep['unexplained'].append(('top-%d membership' % n, int(a), m))
    rep['scores_within_tol_of_the_cut'] = int((np.abs(s_ref - T) <= tol_s).sum())
    # (b) order of the common candidates
    pos_hip = {a: i for i, a in enumerate(o_hip.tolist())}
    common = np.asarray([a for a in o_ref.tolist() if a in pos_hip], np.int64)       # in reference order
    hr = np.asarray([pos_hip[a] for a in common.tolist()], np.int64)
    inv = 0
    for i0 in range(0, common.shape[0], chunk):
        blk = hr[i0:i0 + chunk, None] > hr[None, :]                                   # i before j in ref, after j in hip
        blk &= (np.arange(i0, min(i0 + chunk, common.shape[0]))[:, None] < np.arange(common.shape[0])[None, :])
        ii, jj = np.nonzero(blk)
        inv += ii.shape[0]
        if ii.shape[0]:
            gaps = np.abs(s_ref[common[i0 + ii]] - s_ref[common[jj]])
            bad = gaps > tol_s
            for k in np.nonzero(bad)[0][:10]:
                rep['unexplained'].append(('order', (int(common[i0 + ii[k]]), int(common[jj[k]])), float(gaps[k])))
    rep['order_inversions'] = inv
    # (c) IoU > thresh decisions between common candidates, both eyes
    row_ref = {a: i for i, a in enumerate(o_ref.tolist())}
    ir = np.asarray([row_ref[a] for a in common.tolist()], np.int64)
    rep['iou_flips'] = 0
    eps_b = 0.0
    for eye in ('dets_left', 'dets_right'):
        br, bh = ref[eye][ir, :4].astype(np.float32), hip[eye][hr, :4].astype(np.float32)
        eps_b = max(eps_b, float(np.abs(br - bh).max()) if br.size else 0.0)
        side = np.minimum(br[:, 2] - br[:, 0], br[:, 3] - br[:, 1]) + np.float32(1)
        for i0 in range(0, br.shape[0], chunk):
            a = _iou_f32(br[i0:i0 + chunk], br)
            b = _iou_f32(bh[i0:i0 + chunk], bh)
            ii, jj = np.nonzero((a > np.float32(thresh)) != (b > np.float32(thresh)))
            rep['iou_flips'] += int(ii.shape[0])
            for k in range(ii.shape[0]):
                m = abs(float(a[ii[k], jj[k]]) - thresh)
                tol = 8.0 * max(eps_b, 1e-7) / float(min(side[i0 + ii[k]], side[jj[k]])) + 1e-6
                if m > tol:
                    rep['unexplained'].append(('IoU %s' % eye, (int(common[i0 + ii[k]]), int(common[jj[k]])), m))
    rep['eps_box_px'] = eps_b
    rep['decisions_that_differ'] = rep['membership_flips'] + rep['order_inversions'] + rep['iou_flips']
    rep['same_keep'] = bool(np.array_equal(o_ref[ref['keep']], o_hip[hip['keep']]))
    if rep['decisions_that_differ'] == 0:
        assert rep['same_keep'], 'no discrete decision differs, yet the kept proposals do: the algorithm is not the same'
    return rep


def hip_run_from_workspace(plan, rois_left, rois_right, b=0):
    """The HIP proposal stage's ACTUAL intermediates of the plan's last forward, read out of the proposal workspace
    (include/srcnn_hip.h: srcnn_proposal_workspace_layout) -- the kernels' own candidate order, decoded boxes and keep lists, not a
    re-computation.  Call on the stream the forward ran on, after synchronising.  rois_left / rois_right: the forward's own outputs
    (an eager forward hands its result buffers over and the plan allocates fresh ones)."""
    import ctypes
    from stereo_rcnn_amd import _lib
    from stereo_rcnn_amd.model.utils.config import cfg
    L = _lib.lib()
    B, A, pre, post = plan.B, plan.A, cfg.TEST.RPN_PRE_NMS_TOP_N, plan.post
    n = pre if 0 < pre < A else A
    off = (ctypes.c_size_t * 5)()
    _lib.check(L.srcnn_proposal_workspace_layout(B, A, pre, off, 5), 'srcnn_proposal_workspace_layout')
    ws = _lib.workspace(L.srcnn_proposal_workspace_bytes(B, A, pre, post), plan.dev, 'proposal')
    raw = ws.cpu().numpy()
    order = raw[off[0]:off[0] + B * n * 4].view(np.int32).reshape(B, n)[b].astype(np.int64)
    dets = raw[off[1]:off[1] + B * 2 * n * 5 * 4].view(np.float32).reshape(B, 2, n, 5)[b]
    keep = raw[off[2]:off[2] + B * 2 * n * 4].view(np.int32).reshape(B, 2, n)[b]
    num = raw[off[3]:off[3] + B * 2 * 4].view(np.int32).reshape(B, 2)[b]
    kl, kr = keep[0, :num[0]].astype(np.int64), keep[1, :num[1]].astype(np.int64)
    return {'order': order, 'dets_left': dets[0].copy(), 'dets_right': dets[1].copy(), 'keep_left': kl, 'keep_right': kr,
            'keep': np.intersect1d(kl, kr)[:post], 'fg': plan.probs[b, :, 1].cpu().numpy().astype(np.float32),
            'rois_left': rois_left[b:b + 1].cpu(), 'rois_right': rois_right[b:b + 1].cpu()}
