"""CPU: pins the C oracle (oracle/csrc/oracle_ops.c) against an independent numpy/pure-Python
restatement and against the committed known-answer vectors (tests/golden/ops_golden.npz).
The reference ships no tests/golden vectors for these ops, so these are the pins."""
import os

import numpy as np
import pytest

from oracle import ops

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ops_golden.npz')


def _rand_dets(rng, n):
    nc = max(1, n // 10)
    cx = rng.uniform(0, 1987, nc); cy = rng.uniform(0, 600, nc); s = rng.uniform(16, 300, nc)
    idx = rng.integers(0, nc, n)
    x = cx[idx] + rng.normal(0, 0.15, n) * s[idx]; y = cy[idx] + rng.normal(0, 0.15, n) * s[idx]
    bw = s[idx] * rng.uniform(0.7, 1.4, n); bh = s[idx] * rng.uniform(0.5, 1.2, n)
    b = np.stack([x - bw / 2, y - bh / 2, x + bw / 2, y + bh / 2], 1)
    sc = np.sort(rng.uniform(0, 1, n))[::-1]
    return np.concatenate([b, sc[:, None]], 1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 5, 64, 65, 130, 700])
@pytest.mark.parametrize("thresh", [0.3, 0.7])
def test_nms_c_equals_numpy_restatement(n, thresh):
    d = _rand_dets(np.random.default_rng(n), n)
    assert np.array_equal(ops.nms(d, thresh), ops.nms_py(d, thresh))


def test_nms_mask_reduces_to_keep_list():
    """OR-reducing the 64-bit mask rows in index order (nms_cuda_kernel.cu:131-144) == keep list."""
    d = _rand_dets(np.random.default_rng(9), 300)
    mask = ops.nms_mask(d, 0.7)
    cb = mask.shape[1]
    remv = np.zeros(cb, np.uint64)
    keep = []
    for i in range(300):
        if not (int(remv[i // 64]) >> (i % 64)) & 1:
            keep.append(i)
            remv[i // 64:] |= mask[i, i // 64:]
    assert np.array_equal(np.asarray(keep, np.int32), ops.nms(d, 0.7))


def test_nms_semantics():
    assert ops.nms(np.zeros((0, 5), np.float32), 0.5).shape == (0,)
    # strict '>' : IoU exactly equal to the threshold does NOT suppress (nms_cuda_kernel.cu:78)
    a = np.array([[0, 0, 9, 9, 1.0], [0, 0, 9, 4, 0.9]], np.float32)    # IoU = 50/100 = 0.5
    assert ops.nms(a, 0.5).tolist() == [0, 1]
    assert ops.nms(a, 0.49).tolist() == [0]
    # "+1" convention: two single-pixel boxes at the same pixel overlap fully
    b = np.array([[3, 3, 3, 3, 1.0], [3, 3, 3, 3, 0.5]], np.float32)
    assert ops.nms(b, 0.99).tolist() == [0]
    # suppression is not transitive through removed boxes
    c = np.array([[0, 0, 10, 10, .9], [4, 0, 14, 10, .8], [8, 0, 18, 10, .7]], np.float32)
    assert ops.nms(c, 0.4).tolist() == [0, 2]


def test_nms_golden_vectors():
    g = np.load(GOLD)
    for th, key in ((0.3, 'nms_keep_3'), (0.7, 'nms_keep_7')):
        assert np.array_equal(ops.nms(g['nms_dets'], th), g[key])


def test_roi_align_c_equals_python_loops():
    rng = np.random.default_rng(1)
    feat = rng.normal(0, 1, (2, 3, 10, 13)).astype(np.float32)
    rois = np.array([[0, 10, 5, 100, 60], [1, 0, 0, 0, 0], [1, 150, 100, 210, 170], [0, -20, -5, 30, 20]], np.float32)
    for a in (2, 5, 8):
        assert np.array_equal(ops.roi_align_forward(feat, rois, a, a, 10 / 160.0),
                              ops.roi_align_forward_py(feat, rois, a, a, 10 / 160.0))


def test_roi_align_semantics():
    feat = np.arange(2 * 1 * 4 * 5, dtype=np.float32).reshape(2, 1, 4, 5)
    # a lattice point exactly on a pixel returns that pixel; a 1-px roi puts ALL lattice points inside
    # [x1, x1 + 1/(A-1)*...]: roi (1,1)-(1,1), scale 1 -> width = 1 -> bin = 1/(A-1)
    out = ops.roi_align_forward(feat, np.array([[0, 1, 1, 1, 1]], np.float32), 2, 2, 1.0)
    assert out.shape == (1, 1, 2, 2)
    assert out[0, 0, 0, 0] == feat[0, 0, 1, 1] and out[0, 0, 1, 1] == feat[0, 0, 2, 2]
    # batch index selects the image
    out1 = ops.roi_align_forward(feat, np.array([[1, 1, 1, 1, 1]], np.float32), 2, 2, 1.0)
    assert out1[0, 0, 0, 0] == feat[1, 0, 1, 1]
    # points outside [0,H)x[0,W) are zero (roi_align_kernel.cu:54-55)
    out2 = ops.roi_align_forward(feat, np.array([[0, 10, 10, 12, 12]], np.float32), 3, 3, 1.0)
    assert not out2.any()
    # wrong roi width -> op refuses (roi_align_cuda.c:19-22)
    assert ops.roi_align_forward(feat, np.zeros((2, 4), np.float32), 2, 2, 1.0) is None
    # avg pool of the lattice
    lat = ops.roi_align_forward(feat, np.array([[0, 0, 0, 3, 2]], np.float32), 3, 3, 1.0)
    avg = ops.roi_align_avg(feat, np.array([[0, 0, 0, 3, 2]], np.float32), 2, 2, 1.0)
    assert np.allclose(avg[0, 0, 0, 0], lat[0, 0, :2, :2].mean())


def test_roi_align_golden_vectors():
    g = np.load(GOLD)
    s = float(g['ra_scale'])
    assert np.array_equal(ops.roi_align_forward(g['ra_feat'], g['ra_rois'], 8, 8, s), g['ra_out8'])
    assert np.array_equal(ops.roi_align_avg(g['ra_feat'], g['ra_rois'], 7, 7, s), g['ra_avg7'])
