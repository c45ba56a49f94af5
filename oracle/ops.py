"""CPU oracle for the reference's two native operators (TEST INFRASTRUCTURE ONLY).

Two independent restatements are kept on purpose:
  * the C one (oracle/csrc/oracle_ops.c, via ctypes) -- fast, used as THE oracle;
  * the numpy/pure-Python one below (`*_py`) -- slow, used only to pin the C one
    on small cases (tests/test_oracle_ops.py).
Reference lines followed: see the header of oracle_ops.c.
Parity status: PINNED -- the reference's own CUDA kernels, built unchanged for gfx950 (oracle/_ref), give bit-identical
keep lists and ROIAlign outputs on the MI355X (tests/test_ref_kernels_gpu.py).
"""
import ctypes
import math

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.oracle_nms.restype = ctypes.c_int
        _lib.oracle_nms_mask.restype = ctypes.c_int
        _lib.oracle_roi_align_forward.restype = ctypes.c_int
        _lib.oracle_avgpool2x2_s1.restype = ctypes.c_int
    return _lib


def _fptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- NMS
def nms(dets, thresh):
    """Greedy NMS on score-sorted dets (N, >=4) float32 -> kept indices int32 (k,).

    nms_wrapper.py:13-21 returns [] for empty input; here an empty int32 array.
    """
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), np.int32)
    keep = np.zeros((n,), np.int32)
    num = ctypes.c_int(0)
    rc = lib().oracle_nms(_fptr(dets), ctypes.c_int(n), ctypes.c_int(dets.shape[1]),
                          ctypes.c_float(thresh), _fptr(keep), ctypes.byref(num))
    assert rc == 0
    return keep[:num.value].copy()


def nms_mask(dets, thresh):
    """The (N, ceil(N/64)) uint64 suppression mask of nms_cuda_kernel.cu:41-85."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), np.uint64)
    lib().oracle_nms_mask(_fptr(dets), ctypes.c_int(n), ctypes.c_int(dets.shape[1]),
                          ctypes.c_float(thresh), _fptr(mask))
    return mask


def iou_matrix_py(boxes):
    """float32 IoU(+1) matrix, numpy elementwise (each op rounds to float32)."""
    b = np.asarray(boxes, np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    one = np.float32(1.0)
    zero = np.float32(0.0)
    left = np.maximum(x1[:, None], x1[None, :])
    right = np.minimum(x2[:, None], x2[None, :])
    top = np.maximum(y1[:, None], y1[None, :])
    bottom = np.minimum(y2[:, None], y2[None, :])
    w = np.maximum(right - left + one, zero)
    h = np.maximum(bottom - top + one, zero)
    inter = w * h
    area = (x2 - x1 + one) * (y2 - y1 + one)
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (area[:, None] + area[None, :] - inter)


def nms_py(dets, thresh):
    """Independent restatement: mask rows OR-reduced in index order
    (nms_cuda_kernel.cu:131-144), masks from a vectorised float32 IoU matrix."""
    dets = np.asarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), np.int32)
    over = iou_matrix_py(dets[:, :4]) > np.float32(thresh)
    over = np.triu(over, k=1)  # box i may only suppress j > i (:74-76)
    removed = np.zeros((n,), bool)
    keep = []
    for i in range(n):
        if not removed[i]:
            keep.append(i)
            removed |= over[i]
    return np.asarray(keep, np.int32)


# ---------------------------------------------------------------------- ROIAlign
def roi_align_forward(feat, rois, ah, aw, scale):
    """Legacy lattice ROIAlign. feat (B,C,H,W) f32, rois (n,5) [b,x1,y1,x2,y2] -> (n,C,ah,aw)."""
    feat = np.ascontiguousarray(feat, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    if rois.ndim != 2 or rois.shape[1] != 5:
        return None  # roi_align_cuda.c:19-22 returns 0 and leaves the output untouched
    b, c, h, w = feat.shape
    n = rois.shape[0]
    out = np.zeros((n, c, ah, aw), np.float32)
    lib().oracle_roi_align_forward(ctypes.c_int(ah), ctypes.c_int(aw), ctypes.c_float(scale),
                                   _fptr(feat), ctypes.c_int(b), ctypes.c_int(c), ctypes.c_int(h),
                                   ctypes.c_int(w), _fptr(rois), ctypes.c_int(n), _fptr(out))
    return out


def avgpool2x2_s1(x):
    x = np.ascontiguousarray(x, np.float32)
    n, c, ah, aw = x.shape
    out = np.zeros((n, c, ah - 1, aw - 1), np.float32)
    lib().oracle_avgpool2x2_s1(_fptr(x), ctypes.c_int(n * c), ctypes.c_int(ah), ctypes.c_int(aw), _fptr(out))
    return out


def roi_align_avg(feat, rois, a_h, a_w, scale):
    """RoIAlignAvg.forward (modules/roi_align.py:26-29): (A+1)^2 lattice then avg_pool2d(2,1)."""
    return avgpool2x2_s1(roi_align_forward(feat, rois, a_h + 1, a_w + 1, scale))


def roi_align_forward_py(feat, rois, ah, aw, scale):
    """Pure-Python loop restatement of roi_align_kernel.cu:27-68 (tiny cases only)."""
    f32 = np.float32
    feat = np.asarray(feat, f32)
    rois = np.asarray(rois, f32)
    _, c, height, width = feat.shape
    flat = feat.reshape(-1)
    n = rois.shape[0]
    out = np.zeros((n, c, ah, aw), f32)
    scale = f32(scale)
    for i in range(n):
        bi = rois[i, 0]
        sw, sh = f32(rois[i, 1] * scale), f32(rois[i, 2] * scale)
        ew, eh = f32(rois[i, 3] * scale), f32(rois[i, 4] * scale)
        rw = max(f32(float(f32(ew - sw)) + 1.0), f32(0))
        rh = max(f32(float(f32(eh - sh)) + 1.0), f32(0))
        bh = f32(float(rh) / (float(ah) - 1.0))
        bw = f32(float(rw) / (float(aw) - 1.0))
        img_start = int(f32(f32(f32(bi * f32(c)) * f32(height)) * f32(width)))
        for ch in range(c):
            for ph in range(ah):
                for pw in range(aw):
                    hh = f32(f32(f32(ph) * bh) + sh)
                    ww = f32(f32(f32(pw) * bw) + sw)
                    if hh < 0 or hh >= height or ww < 0 or ww >= width:
                        continue
                    hs = int(min(math.floor(hh), height - 2))
                    ws = int(min(math.floor(ww), width - 2))
                    hr = float(f32(hh - f32(hs)))
                    wr = float(f32(ww - f32(ws)))
                    ul = img_start + (ch * height + hs) * width + ws
                    # usual arithmetic conversions, left to right (roi_align_kernel.cu:64-67): double products for the two
                    # upper taps, float x float first for the lower ones
                    dl_h = f32(flat[ul + width] * f32(hr))
                    dr_hw = f32(f32(flat[ul + width + 1] * f32(hr)) * f32(wr))
                    v = (float(flat[ul]) * (1. - hr) * (1. - wr)
                         + float(flat[ul + 1]) * (1. - hr) * wr
                         + float(dl_h) * (1. - wr)
                         + float(dr_hw))
                    out[i, ch, ph, pw] = f32(v)
    return out
