"""The REFERENCE'S OWN NMS / ROIAlign kernels on the GPU (TEST INFRASTRUCTURE ONLY).

oracle/_ref/libref_ops*.so are lib/model/nms/src/nms_cuda_kernel.cu and lib/model/roi_align/src/roi_align_kernel.cu
compiled unchanged for gfx950 (oracle/build.py:build_ref).  This module only loads them through ctypes and feeds them
torch device tensors; it is used by tests/test_ref_kernels_gpu.py to check the C restatement (oracle/csrc) and the
product kernels against the reference's real code.  Entry points used:
  nms_cuda_compute(keep_out*, num_out*, boxes*, boxes_num, boxes_dim, thresh)        nms_cuda_kernel.cu:86-161
  ROIAlignForwardLaucher(bottom, scale, num_rois, H, W, C, ah, aw, rois*, top*, stream)   roi_align_kernel.cu:68-86
"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def available(kind='fma'):
    return os.path.exists(os.path.join(HERE, '_ref', 'libref_ops.so' if kind == 'fma' else 'libref_ops_nofma.so'))


def lib(kind='fma'):
    if kind not in _LIBS:
        L = ctypes.CDLL(os.path.join(HERE, '_ref', 'libref_ops.so' if kind == 'fma' else 'libref_ops_nofma.so'))
        L.nms_cuda_compute.restype = None
        L.nms_cuda_compute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_float]
        L.ROIAlignForwardLaucher.restype = ctypes.c_int
        L.ROIAlignForwardLaucher.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p]
        _LIBS[kind] = L
    return _LIBS[kind]


def nms(dets, thresh, kind='fma'):
    """dets (N, 5) float32 device tensor, score-sorted -> kept indices (k,) int32 (device), as nms_gpu.py:7-12."""
    dets = dets.contiguous().float()
    n = int(dets.shape[0])
    keep = torch.zeros((n,), dtype=torch.int32, device=dets.device)
    num = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    torch.cuda.synchronize()
    lib(kind).nms_cuda_compute(keep.data_ptr(), num.data_ptr(), dets.data_ptr(), n, int(dets.shape[1]), float(thresh))
    torch.cuda.synchronize()
    return keep[:int(num[0])]


def roi_align_forward(features, rois, ah, aw, scale, kind='fma'):
    """features (B, C, H, W), rois (n, 5) device tensors -> (n, C, ah, aw), as functions/roi_align.py:15-31."""
    features, rois = features.contiguous().float(), rois.contiguous().float()
    b, c, h, w = features.shape
    out = torch.zeros((int(rois.shape[0]), c, ah, aw), dtype=torch.float32, device=features.device)
    torch.cuda.synchronize()
    lib(kind).ROIAlignForwardLaucher(features.data_ptr(), float(scale), int(rois.shape[0]), h, w, c, ah, aw,
                                     rois.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
    return out
