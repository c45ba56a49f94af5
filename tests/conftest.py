import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, 'tests') not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _serving_plans():
    """On a GPU box the whole suite runs on the conv plans the product serves with: `serving.enter(n > 1)` -- every streamed entry
    point -- adopts stereo_rcnn_amd/plans/mi355x.json once per process, and some of its shape keys (the 300-roi head GEMMs) do not
    depend on the frame size.  A test that compares a lone run with a streamed one bit for bit must not straddle that moment
    (another tile / split-K plan adds the K products in another order), so the plans are adopted up front; shapes the file does
    not list are tuned in situ as always."""
    import torch
    if torch.cuda.is_available():
        from stereo_rcnn_amd import serving
        serving.load_shipped_plans()
    yield


def pytest_sessionfinish(session, exitstatus):
    """Largest float deviations the parity tests observed (tests/tolerances.py): printed, and left in gpurun_out/ on a GPU box."""
    try:
        import tolerances
    except ImportError:
        from tests import tolerances
    m = tolerances.measured()
    if m:
        print("\nmeasured maxima (tests/tolerances.py): " + ", ".join("%s %.2e" % (k, m[k]) for k in sorted(m)))
        tolerances.dump(os.path.join(ROOT, 'gpurun_out', 'measured_tolerances.json'))
