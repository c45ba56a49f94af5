"""Shims that let the reference's own Python (read-only, /root/reference/lib) run in THIS container
(py3.10, torch 2.10, no GPU, no OpenCV/easydict/torchvision) so that golden vectors can be produced by
the REFERENCE code rather than by a restatement.  Used only by make_reference_golden.py; nothing here
is imported by the product, the tests or the GPU box (where /root/reference does not exist).

What is shimmed, and why it does not change what the reference computes:
  * sys.path gets lib/, lib/model/rpn, lib/model/utils: the reference uses py2 implicit relative imports
    (`from generate_anchors import ...`, `import kitti_utils as utils`).
  * easydict / cv2 / torchvision: absent offline.  EasyDict is re-created (attribute dict); cv2 and torchvision
    are empty modules (nothing on the paths exercised here calls them).
  * generate_anchors.py: one py2 `print` statement in its __main__ block and `xrange` -> fixed IN MEMORY at load.
  * the two cffi extensions (`model.nms._ext.nms`, `model.roi_align._ext.roi_align`) are CUDA + THC and cannot be
    built: `nms_cuda` and the legacy autograd `RoIAlignFunction` are served by the oracle's C restatement
    (oracle/csrc/oracle_ops.c).  So NMS and ROIAlign themselves are NOT pinned by these goldens -- everything
    around them (network wiring, anchors, proposal layer, level routing, heads, softmaxes, solvers, dense
    alignment, KITTI helpers) is the reference's code.
  * `.cuda()` is made the identity and `torch.cuda.FloatTensor` etc. alias the CPU types (the code runs on CPU tensors); `torch.cuda.is_available()` is forced True
    only while `model.nms.nms_wrapper` is imported so that it binds `nms_gpu`.
  * torch-0.3 semantics that changed: `F.upsample(mode='bilinear')` and `F.grid_sample` were align_corners=True (restored
    here; dense_align.py:195-197 normalises its grid with (size-1)/2, i.e. for exactly that convention);
    `Variable(volatile=True)` is a no-op wrapper (generation runs under no_grad).
    Scalar indexing returned Python numbers in 0.3 and 0-dim tensors now, so a few scalar expressions run in
    float32 instead of double here; where that matters the comparison tolerances in the tests say so.
  * `torch.cat` accepts an input with fewer dimensions as torch 0.3 did (missing trailing dimensions count as size 1).
  * `scipy.array` (removed numpy alias) is pointed at `numpy.array` inside the reference's box_estimator module, and that
    module's `minimize` is wrapped by a recorder that forwards to scipy (make_reference_golden.py:solver_golden).
"""
import re
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference/lib'


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def install(oracle_ops):
    sys.dont_write_bytecode = True            # never write __pycache__ next to the reference
    for p in (REF, REF + '/model/rpn', REF + '/model/utils'):
        if p not in sys.path:
            sys.path.insert(0, p)
    m = types.ModuleType('easydict')
    m.EasyDict = EasyDict
    sys.modules['easydict'] = m
    sys.modules['cv2'] = types.ModuleType('cv2')
    tv = types.ModuleType('torchvision')
    tv.models, tv.utils = types.ModuleType('torchvision.models'), types.ModuleType('torchvision.utils')
    sys.modules.update({'torchvision': tv, 'torchvision.models': tv.models, 'torchvision.utils': tv.utils})

    path = REF + '/model/rpn/generate_anchors.py'
    src = re.sub(r'^(\s*)print (.+)$', r'\1print(\2)', open(path).read(), flags=re.M)
    ga = types.ModuleType('generate_anchors')
    ga.__file__ = path
    ga.xrange = range
    exec(compile(src, path, 'exec'), ga.__dict__)
    sys.modules['generate_anchors'] = ga

    ext_nms = types.ModuleType('model.nms._ext.nms')

    def nms_cuda(keep, dets, num_out, thresh):
        k = oracle_ops.nms(dets.detach().numpy().astype(np.float32), float(thresh))
        keep[:len(k), 0] = torch.from_numpy(np.asarray(k, dtype=np.int32))
        num_out[0] = len(k)
        return 1
    ext_nms.nms_cuda = nms_cuda
    pkg = types.ModuleType('model.nms._ext')
    pkg.nms, pkg.__path__ = ext_nms, []
    sys.modules.update({'model.nms._ext': pkg, 'model.nms._ext.nms': ext_nms})

    fn = types.ModuleType('model.roi_align.functions.roi_align')

    class RoIAlignFunction(object):     # the legacy (instance-forward) autograd.Function cannot be called in torch 2.x
        def __init__(self, aligned_height, aligned_width, spatial_scale):
            self.ah, self.aw, self.scale = int(aligned_height), int(aligned_width), float(spatial_scale)

        def __call__(self, features, rois):
            out = oracle_ops.roi_align_forward(features.detach().numpy().astype(np.float32),
                                               rois.detach().numpy().astype(np.float32), self.ah, self.aw, self.scale)
            return torch.from_numpy(out)
    fn.RoIAlignFunction = RoIAlignFunction
    sys.modules['model.roi_align.functions.roi_align'] = fn

    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ('FloatTensor', 'DoubleTensor', 'LongTensor', 'IntTensor', 'ByteTensor'):   # `.type(torch.cuda.FloatTensor)`
        setattr(torch.cuda, name, getattr(torch, name))
    torch.nn.Module.cuda = lambda self, *a, **k: self
    F.upsample = lambda x, size=None, scale_factor=None, mode='nearest', align_corners=None: F.interpolate(
        x, size=size, scale_factor=scale_factor, mode=mode, align_corners=(True if mode == 'bilinear' else None))

    _grid_sample = F.grid_sample          # torch 0.3 (and up to 1.2): grid_sample sampled with align_corners=True semantics
    F.grid_sample = lambda inp, grid, mode='bilinear', padding_mode='zeros', align_corners=None: _grid_sample(
        inp, grid, mode=mode, padding_mode=padding_mode, align_corners=True)

    # torch 0.3's TH catArray treated a tensor with fewer dimensions as having size 1 in the missing trailing ones
    # (box_3d.py:97 concatenates an (H, W) tensor to an (H, W, 2) one along dim 2): same rule here
    _cat = torch.cat

    def cat_legacy(tensors, dim=0, **kw):
        tensors = list(tensors)
        nd = max(t.dim() for t in tensors)
        if any(t.dim() != nd for t in tensors):
            tensors = [t.reshape(tuple(t.shape) + (1,) * (nd - t.dim())) for t in tensors]
        return _cat(tensors, dim, **kw)
    torch.cat = cat_legacy

    avail = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        from model.nms import nms_wrapper  # noqa: F401  (binds nms_gpu, which needs the _ext stub above)
    finally:
        torch.cuda.is_available = avail
