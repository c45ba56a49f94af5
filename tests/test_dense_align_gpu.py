"""Dense alignment (A15/A16): HIP `align_parallel` vs the CPU oracle on the same inputs, plus an
oracle-independent property (a planted constant disparity is recovered)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n, seed):
    from oracle import dense_align as oda
    rng = np.random.default_rng(seed)
    calib = oda.KITTI_DEMO_CALIB
    poses = []
    for _ in range(n):
        z = rng.uniform(7, 45)
        x = rng.uniform(-0.6, 0.6) * z * 0.8
        poses.append([x, rng.uniform(1.4, 1.8), z, 1.6 * rng.uniform(0.9, 1.1), 1.5 * rng.uniform(0.9, 1.1),
                      4.0 * rng.uniform(0.9, 1.1), rng.uniform(-np.pi, np.pi)])
    poses = torch.tensor(poses, dtype=torch.float32)
    boxes = torch.tensor([oda.project_box(calib, p) for p in poses], dtype=torch.float32)
    boxes[:, 0::2].clamp_(0, 1241)
    boxes[:, 1::2].clamp_(0, 374)
    kp = torch.zeros(n, 5)
    kp[:, 3] = boxes[:, 0] + rng.uniform(0, 3, n).astype(np.float32)
    kp[:, 4] = boxes[:, 2] - rng.uniform(0, 3, n).astype(np.float32)
    return calib, poses, boxes, kp


def _run_both(dev, l, r, info, calib, poses, boxes, kp):
    from oracle import dense_align as oda
    from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
    st_ref, dis_ref, ex = oda.align_parallel(calib, float(info[0, 2]), l, r, boxes, kp, poses, return_extra=True)
    st, dis = align_parallel(calib, float(info[0, 2]), l.to(dev), r.to(dev), boxes.to(dev), kp.to(dev), poses.to(dev))
    torch.cuda.synchronize()
    return st.cpu(), dis.cpu(), st_ref, dis_ref, ex


@pytest.mark.parametrize("seed,n", [(1, 1), (2, 6), (3, 12)])
def test_align_parallel_matches_oracle(dev, seed, n):
    from stereo_rcnn_amd import fixture
    l, r, info = fixture.make_inputs(seed, 375, 1242)
    calib, poses, boxes, kp = _scene(n, seed)
    st, dis, st_ref, dis_ref, ex = _run_both(dev, l, r, info, calib, poses, boxes, kp)
    assert torch.equal(st, st_ref)
    d = (dis - dis_ref).abs()
    # the argmin over 50+20 hypotheses is discrete: identical index -> float-level agreement;
    # a flipped near-tie moves the result by one 0.05 m fine step (a few 1e-2 px at most here)
    exact = d < 1e-3
    assert float(exact.float().mean()) >= 0.8, (d, dis, dis_ref)
    fb = 721.5377 * 0.5327
    step_px = fb * 0.05 / (ex['fine_depth'] ** 2) * 1.5 + 1e-3 if ex else None
    assert bool((d <= torch.clamp(step_px, min=2e-3)).all()), (d, step_px)


def test_align_parallel_no_valid_pixels(dev):
    """Objects whose box does not see the 3-D box: status 0; all-invalid -> dis_init (dense_align.py:272-277)."""
    from oracle import dense_align as oda
    from stereo_rcnn_amd import fixture
    l, r, info = fixture.make_inputs(4, 375, 1242)
    calib = oda.KITTI_DEMO_CALIB
    poses = torch.tensor([[-3.0, 1.6, 12.0, 1.6, 1.5, 4.0, 0.3], [30.0, 1.6, 12.0, 1.6, 1.5, 4.0, 0.3]])
    good = torch.tensor([oda.project_box(calib, poses[0])], dtype=torch.float32)
    boxes = torch.cat((good, good), 0)            # second pose is far off to the right of its (wrong) box
    kp = torch.zeros(2, 5)
    kp[:, 3], kp[:, 4] = boxes[:, 0], boxes[:, 2]
    st, dis, st_ref, dis_ref, _ = _run_both(dev, l, r, info, calib, poses, boxes, kp)
    assert st_ref.tolist() == [1.0, 0.0] and torch.equal(st, st_ref)
    assert float((dis - dis_ref).abs().max()) < 2e-2
    st, dis, st_ref, dis_ref, _ = _run_both(dev, l, r, info, calib, poses[1:], boxes[1:], kp[1:])
    assert st_ref.tolist() == [0.0] and torch.equal(st, st_ref)
    assert float((dis - dis_ref).abs().max()) < 1e-4          # == dis_init


def test_planted_disparity_is_recovered(dev):
    """Right image = left shifted by a constant disparity; a fronto-parallel-ish object at the matching
    depth must align to that disparity (no oracle involved)."""
    from oracle import dense_align as oda       # only for the calibration constants / box projection helper
    from stereo_rcnn_amd import fixture
    from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
    rng = np.random.default_rng(0)
    H, W, d0 = 375, 1242, 24.0                   # original-image pixels
    tex = 0.6 * fixture._smooth_noise(rng, H, W + 64, 24) + 0.4 * fixture._smooth_noise(rng, H, W + 64, 5)
    left = np.clip(np.rint(tex[:, :W] * 255), 0, 255).astype(np.uint8)
    right = np.clip(np.rint(tex[:, int(d0):W + int(d0)] * 255), 0, 255).astype(np.uint8)   # x_r = x_l - d0
    tl, s = fixture.preprocess(left)
    tr, _ = fixture.preprocess(right)
    calib = oda.KITTI_DEMO_CALIB
    fb = calib.p2[0, 0] * (calib.p2[0, 3] - calib.p3[0, 3]) / calib.p2[0, 0]
    z = fb / d0
    pose = torch.tensor([[0.0, 1.6, z + 2.0, 1.6, 1.5, 4.0, 0.0]])      # start 2 m off; search spans +-12.5 m
    box = torch.tensor([oda.project_box(calib, [0.0, 1.6, z, 1.6, 1.5, 4.0, 0.0])], dtype=torch.float32)
    kp = torch.zeros(1, 5)
    kp[:, 3], kp[:, 4] = box[:, 0], box[:, 2]
    st, dis = align_parallel(calib, s, tl.to(dev), tr.to(dev), box.to(dev), kp.to(dev), pose.to(dev))
    assert st.tolist() == [1.0]
    # the front face of the (frontal) box is l/2 = 2 m closer than its centre: the aligned CENTRE depth z*
    # satisfies fb/(z*-2) ~ d0  ->  reported disparity fb/z* + 0.5
    z_star = fb / d0 + 2.0
    assert abs(float(dis[0]) - (fb / z_star + 0.5)) < 0.6, float(dis[0])


@pytest.mark.parametrize("seed", [2, 3])
def test_align_parallel_vs_reference_code_golden(dev, seed):
    """HIP dense alignment against the REFERENCE'S OWN align_parallel (tests/golden/reference_misc.npz, written by
    tests/golden/make_reference_golden.py): same status, aligned disparity equal up to one fine depth step on a flipped
    near-tie of the discrete argmin."""
    import os
    from oracle.dense_align import KITTI_DEMO_CALIB as calib       # calibration constants only (== the reference's demo/calib.txt)
    from stereo_rcnn_amd import fixture
    from stereo_rcnn_amd.model.dense_align.dense_align import align_parallel
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_misc.npz'))
    t = 'da%d_' % seed
    l, r, info = fixture.make_inputs(seed, 375, 1242)
    st, dis = align_parallel(calib, float(info[0, 2]), l.to(dev), r.to(dev), torch.from_numpy(g[t + 'boxes']).to(dev),
                             torch.from_numpy(g[t + 'kpts']).to(dev), torch.from_numpy(g[t + 'poses']).to(dev))
    torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), g[t + 'status'])
    d = np.abs(dis.cpu().numpy() - g[t + 'best_dis'])
    z = g[t + 'poses'][:, 2]
    step_px = 721.5377 * 0.5327 * 0.05 / np.maximum(z - 13.0, 1.5) ** 2 * 1.5 + 2e-3     # one fine step at the nearest bracket depth
    assert float((d < 1e-3).mean()) >= 0.8 and bool((d <= step_px).all()), (d, step_px)
