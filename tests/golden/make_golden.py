"""Generates the committed golden vectors from the CPU oracle (run in the authoring container):

    python tests/golden/make_golden.py [full|small|ops]

  ops   -> ops_golden.npz   : NMS / ROIAlign known-answer vectors (pins the C oracle)
  small -> small_r101_seed3.npz : whole-network oracle outputs at a reduced input (192x640)
  full  -> full_r101_seed3.npz  : whole-network oracle outputs at BASELINE size (375x1242 -> 600x1987)

Weights and inputs are pure functions of the seed (stereo_rcnn_amd/fixture.py), so the GPU
box regenerates them bit-identically and compares the HIP path with these files without
needing the oracle to run the full-size network there (minutes of CPU).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import net, ops, postprocess  # noqa: E402
from stereo_rcnn_amd import fixture       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sample_positions(numel, k, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, numel, size=min(k, numel))


def network_case(name, height, width, target_short, seed=3):
    torch.set_num_threads(min(os.cpu_count(), 16))
    sd = fixture.make_state_dict(seed)
    l, r, info = fixture.make_inputs(seed, height, width, target_short=target_short)
    t = time.time()
    out = net.forward(sd, l, r, info, keep=True)
    print(name, 'oracle forward %.1fs' % (time.time() - t), tuple(l.shape))
    det = postprocess.decode_detections(out, info)
    cd = postprocess.class_detections(det)
    g = {'im_info': info.numpy(), 'input_shape': np.array(l.shape)}
    for k in ('rois_left', 'rois_right', 'cls_prob', 'bbox_pred', 'dim_orien_pred', 'kpts_prob',
              'left_border_prob', 'right_border_prob'):
        g[k] = out[k].numpy()
    # sampled stage intermediates (position list + values), NCHW flat indexing
    for key, tensors in (('c_left', out['c_left']), ('p_left', out['p_left']), ('c_right', out['c_right']),
                         ('p_right', out['p_right'])):
        for i, tns in enumerate(tensors):
            pos = sample_positions(tns.numel(), 4096, 100 + i)
            g['%s%d_pos' % (key, i)] = pos
            g['%s%d_val' % (key, i)] = tns.reshape(-1)[torch.from_numpy(pos)].numpy()
            g['%s%d_absmean' % (key, i)] = np.float64(tns.double().abs().mean())
    pos = sample_positions(out['rpn_probs'].shape[1], 20000, 7)
    g['rpn_pos'] = pos
    g['rpn_probs_val'] = out['rpn_probs'][0, torch.from_numpy(pos)].numpy()
    g['rpn_deltas_val'] = out['rpn_deltas'][0, torch.from_numpy(pos)].numpy()
    e = out['proposal_extra']
    g['order'] = e['order'][0].numpy().astype(np.int32)
    g['order_scores'] = out['rpn_probs'][0, e['order'][0], 1].numpy()
    g['keep_left'] = e['keep_left'][0]
    g['keep_right'] = e['keep_right'][0]
    g['keep'] = e['keep'][0].astype(np.int32)
    for k in ('boxes_left', 'boxes_right', 'dim_orien', 'kpts', 'scores'):
        g['det_' + k] = det[k].numpy()
    g['cls_keep_idx'] = cd['inds'][cd['order']][torch.from_numpy(cd['keep'].astype(np.int64))].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **g)
    print('wrote', name, sum(v.nbytes for v in g.values()) / 1e6, 'MB raw')


def ops_case():
    rng = np.random.default_rng(2024)
    g = {}
    # clustered, score-sorted boxes; answers from the C oracle, cross-checked with the numpy restatement
    n = 1500
    nc = 120
    cx = rng.uniform(0, 1987, nc); cy = rng.uniform(0, 600, nc); s = rng.uniform(16, 300, nc)
    idx = rng.integers(0, nc, n)
    x = cx[idx] + rng.normal(0, 0.15, n) * s[idx]; y = cy[idx] + rng.normal(0, 0.15, n) * s[idx]
    bw = s[idx] * rng.uniform(0.7, 1.4, n); bh = s[idx] * rng.uniform(0.5, 1.2, n)
    b = np.stack([x - bw / 2, y - bh / 2, x + bw / 2, y + bh / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, 1986); b[:, 1::2] = np.clip(b[:, 1::2], 0, 599)
    dets = np.concatenate([b, np.sort(rng.uniform(0, 1, n))[::-1, None]], 1).astype(np.float32)
    g['nms_dets'] = dets
    for th in (0.3, 0.7):
        k = ops.nms(dets, th)
        assert np.array_equal(k, ops.nms_py(dets, th))
        g['nms_keep_%d' % int(th * 10)] = k
    feat = rng.normal(0, 1, (2, 6, 19, 63)).astype(np.float32)
    rois = np.array([[0, 100, 50, 400, 300], [1, 0, 0, 0, 0], [0, 1500, 10, 1986, 599], [1, 700, 200, 720, 230],
                     [0, -30, -10, 50, 40], [1, 1900, 550, 2100, 700]], np.float32)
    g['ra_feat'] = feat
    g['ra_rois'] = rois
    g['ra_scale'] = np.float32(19 / 600.0)
    o = ops.roi_align_forward(feat, rois, 8, 8, float(g['ra_scale']))
    assert np.array_equal(o, ops.roi_align_forward_py(feat, rois, 8, 8, float(g['ra_scale'])))
    g['ra_out8'] = o
    g['ra_avg7'] = ops.roi_align_avg(feat, rois, 7, 7, float(g['ra_scale']))
    np.savez_compressed(os.path.join(HERE, 'ops_golden.npz'), **g)
    print('wrote ops_golden')


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'small'
    if what == 'ops':
        ops_case()
    elif what == 'small':
        network_case('small_r101_seed3', 120, 400, 192)
    elif what == 'full':
        network_case('full_r101_seed3', 375, 1242, 600)
