"""The tie audit (tests/tie_audit.py) on the CPU: the reference run's proposal stage rebuilt from the golden reproduces the
reference's rois, auditing a run against itself finds nothing, and a run on float-rounding-sized perturbations of the same inputs
differs only through decisions the audit lists -- each a near-tie of the reference's own margins."""
import os

import numpy as np
import torch

import tie_audit

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _gold():
    return np.load(os.path.join(GOLD, 'reference_net_cv_370x1224_r101_seed4.npz'))


def test_golden_input_is_the_opencv_restatement_of_the_370x1224_frame():
    """The golden's network input is what the product's preprocessing produces for a 370x1224 KITTI frame: 600x1985 (cv2's
    cvRound), not the truncating fixture resize's 600x1984."""
    import hashlib
    from oracle import preprocess as opre
    from stereo_rcnn_amd import engine, fixture
    g = _gold()
    seed, h, w, short = [int(v) for v in g['spec']]
    assert (h, w, short) == (370, 1224, 600) and list(g['input_shape']) == [1, 3, 600, 1985]
    assert engine.preprocess_size(h, w, short)[:2] == (600, 1985)
    lu, _ = fixture.synthetic_pair(seed, h, w)
    tl, s = opre.prepare_image(lu)
    assert hashlib.sha256(np.ascontiguousarray(tl).tobytes()).digest() == g['input_sha256'].tobytes()
    assert abs(float(g['im_info'][0, 2]) - s) < 1e-7


def test_reference_proposal_stage_rebuilt_from_the_golden_reproduces_its_rois():
    g = _gold()
    ref = tie_audit.reference_run_from_golden(g)
    assert torch.equal(ref['rois_left'], torch.from_numpy(g['rois_left'])) and torch.equal(ref['rois_right'], torch.from_numpy(g['rois_right']))
    assert ref['order'].shape == (6000,) and set(ref['order'].tolist()) <= set(g['rpn_top_idx'].tolist())
    rep = tie_audit.audit(ref, ref)
    assert rep['decisions_that_differ'] == 0 and rep['same_keep'] and not rep['unexplained'] and rep['eps_score'] == 0.0


def test_audit_explains_every_difference_under_rounding_sized_perturbations():
    g = _gold()
    ref = tie_audit.reference_run_from_golden(g)
    rng = np.random.default_rng(0)
    A = g['rpn_fg'].shape[0]
    deltas = np.zeros((A, 6), np.float32)
    deltas[g['rpn_top_idx']] = g['rpn_top_deltas']
    fg2 = (g['rpn_fg'].astype(np.float64) + rng.uniform(-1e-6, 1e-6, A)).astype(np.float32)
    d2 = (deltas.astype(np.float64) + rng.uniform(-2e-6, 2e-6, deltas.shape)).astype(np.float32)
    hip = tie_audit.proposal_run(fg2, d2, g['im_info'], g['rpn_shapes'])
    rep = tie_audit.audit(ref, hip)
    print({k: v for k, v in rep.items() if k != 'unexplained'})
    assert not rep['unexplained'], rep['unexplained'][:5]
    assert rep['decisions_that_differ'] > 0                      # a seeded random RPN has ties at every scale of perturbation
    if not rep['same_keep']:
        assert rep['decisions_that_differ'] > 0
    # a perturbation far beyond rounding is NOT excused by the measured eps alone: margins are judged against 2 x eps, so the
    # audit scales with the actual input difference and stays a statement about ties
    assert rep['eps_score'] < 2e-6
