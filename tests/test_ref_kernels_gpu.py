"""GPU: the REFERENCE'S OWN CUDA kernels (oracle/_ref, built unchanged for gfx950 by oracle/build.py:build_ref) against
the C restatement of the oracle and against the product's HIP kernels, on the MI355X.  This is the pin for the two
native operators: the same source the reference compiles with nvcc, executed here."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ref():
    from oracle import ref_ops
    if not ref_ops.available('fma') or not ref_ops.available('nofma'):
        pytest.skip('oracle/_ref/libref_ops*.so not built (python -m oracle.build where /root/reference exists)')
    return ref_ops


def _boxes(n, seed, size):
    g = np.random.default_rng(seed)
    cx, cy = g.uniform(0, 1987, n), g.uniform(0, 600, n)
    w, h = size * g.uniform(0.5, 1.5, n), size * g.uniform(0.5, 1.5, n)
    d = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, np.sort(g.uniform(0, 1, n))[::-1]], 1).astype(np.float32)
    d[::7, :4] = np.round(d[::7, :4])            # integer coordinates: exact IoU ties around the threshold
    return d


@pytest.mark.parametrize("n,size", [(1, 50), (64, 80), (65, 80), (1000, 60), (6000, 40), (6000, 150), (4097, 300)])
@pytest.mark.parametrize("thresh", [0.7, 0.3])
def test_reference_nms_kernel_vs_oracle_and_product(dev, ref, n, size, thresh):
    from oracle import ops as oops
    from stereo_rcnn_amd.model.nms.nms_gpu import nms_gpu
    d = _boxes(n, n + int(100 * thresh), size)
    t = torch.from_numpy(d).to(dev)
    want = ref.nms(t, thresh, 'fma').cpu().numpy()
    assert np.array_equal(ref.nms(t, thresh, 'nofma').cpu().numpy(), want)      # devIoU has no multiply-add to contract
    assert np.array_equal(np.asarray(oops.nms(d, thresh), dtype=np.int32), want)           # C restatement
    assert np.array_equal(nms_gpu(t, thresh).view(-1).cpu().numpy(), want)                 # product kernel


@pytest.mark.parametrize("a", [8, 15])
@pytest.mark.parametrize("shape,scale", [((1, 8, 38, 125), 1 / 16.), ((2, 16, 19, 63), 1 / 32.), ((1, 4, 150, 497), 1 / 4.)])
def test_reference_roi_align_kernel_vs_oracle_and_product(dev, ref, a, shape, scale):
    from oracle import ops as oops
    from stereo_rcnn_amd.model.roi_align.functions.roi_align import RoIAlignFunction
    g = np.random.default_rng(a + shape[2])
    feat = g.standard_normal(shape).astype(np.float32)
    n = 60
    x1, y1 = g.uniform(-20, 1900, n), g.uniform(-20, 560, n)
    rois = np.stack([g.integers(0, shape[0], n), x1, y1, x1 + g.uniform(1, 400, n), y1 + g.uniform(1, 200, n)], 1).astype(np.float32)
    rois[0, 1:] = [10, 10, 10, 10]                      # degenerate roi
    rois[1, 1:] = [1980, 590, 2100, 700]                # hanging over the border -> zero taps
    f, r = torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev)
    exact = ref.roi_align_forward(f, r, a, a, scale, 'nofma').cpu().numpy()
    # restatement and product are built without floating-point contraction: bit-equal to the reference kernel built the same way
    assert np.array_equal(oops.roi_align_forward(feat, rois, a, a, scale), exact)
    got = RoIAlignFunction(a, a, scale)(f, r).cpu().numpy()
    assert np.array_equal(got, exact)
    # the same source with floating-point contraction allowed (what nvcc's default -fmad=true does to `ph * bin + start` and
    # to the blend): the lattice coordinate moves by <= 1 float ulp, the sample by ~1e-6 relative -- compiler latitude, far
    # inside every tolerance; the restatements follow the un-contracted source semantics
    fma = ref.roi_align_forward(f, r, a, a, scale, 'fma').cpu().numpy()
    # one ulp of a coordinate near 500 is 3e-5 of a pixel; the test maps are white noise (unit gradient per pixel)
    close = np.isclose(fma, exact, rtol=2e-5, atol=2e-4)
    # (a coordinate that lands exactly on a lattice line or on the map border can flip floor() / the zero test by that ulp)
    assert float(close.mean()) > 0.99
    print('contracted vs un-contracted reference build: %d of %d elements differ at all, %d beyond 2e-4'
          % (int((fma != exact).sum()), fma.size, int((~close).sum())))
