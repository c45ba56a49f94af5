"""CPU: the C-ABI library loads and exports every symbol include/srcnn_hip.h declares
(no compute calls without a GPU); the product path has no fallback."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'srcnn_hip.h')


def _declared():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r'SRCNN_API\s+[\w\s\*]+?\b(\w+)\s*\(', txt)))


@pytest.fixture(scope='module')
def lib_path():
    import __graft_entry__ as ge
    return ge.build()


def test_header_declares_the_legacy_symbols():
    names = _declared()
    assert 'nms_cuda' in names and 'roi_align_forward_cuda' in names      # nms_cuda.h:4, roi_align_cuda.h:1
    assert len(names) >= 25


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    out = subprocess.check_output(['nm', '-D', '--defined-only', lib_path]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln}
    assert set(_declared()) <= exported
    # nothing but the C ABI leaks out (internal C++ symbols are hidden)
    leaked = [s for s in exported if s.startswith('_ZN5srcnn')]
    assert not leaked, leaked[:3]


def test_ctypes_table_matches_header(lib_path):
    from stereo_rcnn_amd import _lib
    assert _lib.declared_symbols() == _declared()
    L = _lib.lib()
    assert L.srcnn_version() >= 100
    assert L.srcnn_nms_workspace_bytes(6000) >= 6000 * 94 * 8
    assert L.srcnn_proposal_workspace_bytes(1, 298476, 6000, 300) > 2 * 6000 * 94 * 8


def test_workspace_query_and_argument_errors_without_gpu(lib_path):
    """Argument validation happens before any launch, so it is checkable on a CPU-only host."""
    from stereo_rcnn_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc()
    d.x, d.w, d.y = 1, 1, 1
    d.B, d.H, d.W, d.Cin, d.x_cstride = 1, 8, 8, 48, 48     # Cin not a multiple of 32
    d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad = 8, 8, 64, 1, 1, 1, 0
    assert L.srcnn_conv2d(ctypes.byref(d), None, 0, None) == -1
    assert b'multiple of 32' in L.srcnn_last_error()
    assert L.roi_align_forward_cuda(8, 8, 1.0, None, 1, 1, 4, 4, None, 3, 4, None, None) == 0   # roi_cols != 5
    # chained / grouped launches validate before they launch, too
    assert L.srcnn_conv2d_chain(None, 1, None) == -1 and L.srcnn_conv2d_group(None, 1, None) == -1
    three = (_lib.ConvDesc * 3)()
    assert L.srcnn_conv2d_chain(three, 4, None) == -1 and b'1 to 3' in L.srcnn_last_error()
    assert L.srcnn_conv2d_chain_supported(2, 8, 2, 2, 2) == 1 and L.srcnn_conv2d_chain_supported(3, 8, 2, 2, 2) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from stereo_rcnn_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no fallback'):
        _lib.lib()


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, 'stereo_rcnn_amd')):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M):
                    bad.append(f)
    assert not bad, bad
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    # bench.py touches the oracle only inside cpu_baseline()
    body = bench.split('def cpu_baseline')[1].split('\ndef ')[0]
    assert bench.count('from oracle') == body.count('from oracle') > 0
