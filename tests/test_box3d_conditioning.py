"""CPU: what "3D box L-inf vs reference <= 1e-4" can mean, quantified on a well-conditioned fixture.

Known cars are projected into detections that the solver's model explains exactly; the reference's solver (the library's host
build == the reference's scipy path bit for bit, tests/test_solvers_cpu.py) is then run on those detections and on copies
moved by the detector's measured error (1e-5).  The spread of ITS OWN end point is the resolution below which
"identical 3-D boxes" is not defined for any implementation whose detections are not bit-equal to the reference's."""
import numpy as np

from conditioning import IM_SHAPE, _wrap, perturb, solve4, spread_4dof, well_posed_cases


def test_fixture_is_well_posed_and_the_solver_recovers_the_planted_boxes():
    cases = well_posed_cases(48, 11)
    assert len(cases) == 48
    errs = []
    for case, planted in cases:
        st, x, ns = solve4(case)
        assert st == 1 and ns in (0, 2)            # scipy: converged / "desired error not necessarily achieved"
        errs.append(float(np.abs(_wrap(x - planted)).max()))
    errs = np.array(errs)
    # the solver stops on scipy's default step tolerance, 1e-3 .. 1e-2 short of the optimum it is heading for
    assert np.median(errs) < 1e-2 and errs.max() < 0.5 and np.median(errs) > 1e-5, (np.median(errs), errs.max())


def test_scipy_path_and_host_build_agree_bit_for_bit_on_the_fixture():
    from oracle.dense_align import KITTI_DEMO_CALIB as calib
    from stereo_rcnn_amd.model.utils import box_estimator as be
    for case, _ in well_posed_cases(6, 5):
        alpha, dim, bl, br, kpts = case
        st_s, x_s = be.solve_x_y_z_theta_from_kpt(IM_SHAPE, calib, alpha, dim, bl, br, kpts)
        st_n, x_n, _ = solve4(case)
        assert st_s == st_n == 1 and np.array_equal(np.asarray(x_s, np.float64), x_n)


def test_reference_end_point_spread_under_detector_sized_perturbations(capsys):
    """The number behind DESIGN section 7: even on exactly explained detections, moving the inputs by 1e-5 moves the reference's
    own 4-DoF end point by more than 1e-4 for most objects -- the iteration count changes -- while the sub-population that
    keeps its iteration path (spread <= 1e-4) stays within ~10 x the perturbation."""
    cases = well_posed_cases(48, 11)
    spreads = np.array([spread_4dof(c, 1e-5, 8, seed=i) for i, (c, _) in enumerate(cases)])
    assert np.isfinite(spreads).all()
    tight = spreads <= 1e-4
    with capsys.disabled():
        print('\n4-DoF end-point spread of the REFERENCE solver, 48 well-posed cars, inputs +-1e-5 (8 draws each): median %.1e, '
              'p90 %.1e, max %.1e; %d/48 within 1e-4 (median of those %.1e)'
              % (np.median(spreads), np.quantile(spreads, 0.9), spreads.max(), int(tight.sum()), np.median(spreads[tight])))
    assert 8 <= int(tight.sum()) <= 40          # both populations exist: the claim is neither "always chaotic" nor "never"
    assert np.median(spreads[tight]) < 5e-5     # a stable iteration path amplifies the input error by a small factor only
    assert np.median(spreads) > 3e-5            # the typical object moves by more than its inputs did
    # a larger perturbation does not make the stable population vanish, it moves with the inputs
    s4 = np.array([spread_4dof(c, 1e-4, 4, seed=100 + i) for i, (c, _) in enumerate(cases[:16])])
    assert np.isfinite(s4).all() and np.median(s4) < 2e-2
