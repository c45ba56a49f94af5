"""ctypes binding of libsrcnn_hip.so (the C-ABI drop-in boundary, include/srcnn_hip.h).

The reference binds its native ops with cffi (`torch.utils.ffi`,
lib/model/nms/_ext/nms/__init__.py:2-15); cffi is not in this image, ctypes gives the
same contract.  There is NO fallback: if the HIP library is missing or fails to load,
importing any operator raises (the product path never routes through the CPU oracle).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRCNN_LIB_PATH") or os.path.join(_HERE, "libsrcnn_hip.so")   # override: A/B builds of the library (dev)

FMT_F32, FMT_SPLIT16 = 0, 1     # SRCNN_FMT_* (include/srcnn_hip.h)
REC_COLS = 32                   # SRCNN_REC_COLS: detection record row (include/srcnn_hip.h lists the columns)

c_int, c_float, c_double, c_void_p, c_size_t = (ctypes.c_int, ctypes.c_float, ctypes.c_double,
                                                ctypes.c_void_p, ctypes.c_size_t)


class ConvDesc(ctypes.Structure):
    """struct srcnn_conv_desc (include/srcnn_hip.h)."""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("y", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("x_cstride", c_int),
                ("OH", c_int), ("OW", c_int), ("Cout", c_int),
                ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int),
                ("y_cstride", c_int), ("y_coffset", c_int), ("res_cstride", c_int),
                ("relu", c_int), ("mode", c_int),
                ("precision", c_int), ("w_lo", c_void_p), ("w_inv_scale", c_float),
                ("tile_mr", c_int), ("tile_nr", c_int), ("splits", c_int),
                ("x_format", c_int), ("y_format", c_int), ("res_format", c_int),
                ("tile_waves", c_int), ("tile_stages", c_int), ("layer_tag", c_int),
                ("m_limit", c_void_p), ("m_limit_mul", c_int),
                ("x2", c_void_p), ("Cin2", c_int), ("H2", c_int), ("W2", c_int), ("x2_cstride", c_int), ("stride2", c_int),
                ("head_w", c_void_p), ("head_bias", c_void_p), ("head_y", c_void_p), ("head_cout", c_int), ("head_scale", c_float),
                ("head_wf", c_void_p), ("head_rows", c_int), ("head_parts", c_int), ("head_plane", ctypes.c_longlong),
                ("up_top", c_void_p), ("up_format", c_int), ("up_H", c_int), ("up_W", c_int)]


_SIGNATURES = {
    # name: (restype, argtypes)
    "srcnn_version": (c_int, []),
    "srcnn_last_error": (ctypes.c_char_p, []),
    "srcnn_nms_workspace_bytes": (c_size_t, [c_int]),
    "srcnn_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "srcnn_nms_batched_workspace_bytes": (c_size_t, [c_int, c_int]),
    "srcnn_nms_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                  c_void_p, c_size_t, c_void_p]),
    "nms_cuda": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "roi_align_forward_cuda": (c_int, [c_int, c_int, c_float, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "srcnn_pyramid_roi_align": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                        c_int, c_float, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p]),
    "srcnn_pool2x2_s1": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_int, c_void_p]),
    "srcnn_act_convert": (c_int, [c_void_p, c_int, c_void_p, c_int, ctypes.c_longlong, c_int, c_void_p]),
    "srcnn_gather_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "srcnn_decode_kept_kpts": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                       c_void_p]),
    "srcnn_conv2d_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "srcnn_conv2d": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_size_t, c_void_p]),
    "srcnn_conv2d_chain_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "srcnn_conv2d_chain": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p]),
    "srcnn_conv2d_group": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p]),
    "srcnn_range_flag_read": (c_int, [c_int]),
    "srcnn_range_flag_device_word": (c_void_p, []),
    "srcnn_range_flag_bind": (c_int, [c_void_p]),
    "srcnn_preprocess": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "srcnn_stem_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "srcnn_stem_pack_pair": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "srcnn_maxpool3x3s2_ceil": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "srcnn_upsample_add": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                   c_int, c_void_p]),
    "srcnn_subsample2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "srcnn_nhwc_to_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "srcnn_nchw_to_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "srcnn_rpn_score": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "srcnn_rpn_score_levels": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "srcnn_rpn_score_parts": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(c_int),
                                      c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "srcnn_proposal_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "srcnn_proposal_workspace_layout": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_size_t), c_int]),
    "srcnn_proposal_layer": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_int), c_int, c_void_p,
                                     c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "srcnn_softmax_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "srcnn_box_head_tail": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srcnn_kpts_tail": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srcnn_decode_detections": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int] + [c_void_p] * 4 + [c_void_p]),
    "srcnn_class_nms_workspace_bytes": (c_size_t, [c_int]),
    "srcnn_class_nms": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "srcnn_pack_detections": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "srcnn_dense_align_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "srcnn_dense_align_workspace_layout": (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_size_t), c_int]),
    "srcnn_dense_align": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_double, c_double, c_double,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "srcnn_box3d_workspace_bytes": (c_size_t, [c_int, c_int]),
    "srcnn_infer_boundary": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "srcnn_solve_4dof": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_float,
                                 c_void_p, c_void_p]),
    "srcnn_solve_4dof_scalar": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_float,
                                        c_void_p, c_void_p]),
    "srcnn_solve_3dof_scalar": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "srcnn_align_inputs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srcnn_solve_3dof": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "srcnn_solve_4dof_records_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double,
                                              c_float, c_void_p, c_int]),
    "srcnn_solve_3dof_records_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double,
                                              c_void_p, c_void_p, c_void_p, c_int]),
    "srcnn_solve_4dof_host": (c_int, [c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "srcnn_solve_3dof_host": (c_int, [c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p, c_void_p,
                                      c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srcnn_solver_evaluate_host": (c_int, [c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srcnn_program_create": (c_void_p, []),
    "srcnn_program_destroy": (None, [c_void_p]),
    "srcnn_program_begin": (c_int, [c_void_p, c_void_p]),
    "srcnn_program_end": (c_int, [c_void_p]),
    "srcnn_program_recording": (c_int, []),
    "srcnn_program_record_event": (c_int, [c_void_p, c_void_p]),
    "srcnn_program_wait_event": (c_int, [c_void_p, c_void_p, c_int]),
    "srcnn_program_size": (c_int, [c_void_p]),
    "srcnn_program_run": (c_int, [c_void_p, c_void_p]),
    "srcnn_stream_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "srcnn_stream_create_cu_mask": (c_int, [c_int, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(c_void_p)]),
    "srcnn_stream_destroy": (c_int, [c_void_p]),
    "srcnn_probe_placement": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "srcnn_prof_enable": (c_int, [c_int]),
    "srcnn_prof_read_launches": (c_int, [ctypes.POINTER(ctypes.c_float), c_int]),
    "srcnn_prof_read": (c_int, [ctypes.POINTER(c_double), ctypes.POINTER(c_double),
                                ctypes.POINTER(ctypes.c_longlong)]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load the HIP library (once).  Raises if it is absent: no CPU/eager fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsrcnn_hip.so not found at %s - build it with `python -m stereo_rcnn_amd.csrc.build` "
                "(or __graft_entry__.build()). The product path has no fallback." % LIB_PATH)
        from . import streams
        streams.hip_may_start()            # the hardware-queue count HIP will start with is decided no later than here
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what="srcnn call"):
    if rc != 0:
        msg = lib().srcnn_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "libsrcnn_hip operates on device memory only"
    assert t.is_contiguous(), "tensor must be contiguous"
    return t.data_ptr()


def stream():
    """Handle of torch's current HIP stream: every launch goes where torch's allocator expects it."""
    return torch.cuda.current_stream().cuda_stream


class Workspace(object):
    """Grow-only device scratch owned by the caller side (one per device)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


_workspaces = {}
_recording_refs = None      # while a launch program records: every workspace handed out (the program keeps them alive)


def workspace(nbytes, device, key="default"):
    # one scratch buffer per (device, purpose, stream): forwards in flight on different streams never share scratch
    k = (str(device), key, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
    if k not in _workspaces:
        _workspaces[k] = Workspace()
    buf = _workspaces[k].get(nbytes, device)
    if _recording_refs is not None:
        _recording_refs.append(buf)
    return buf
