"""`RoIAlignFunction` - reference lib/model/roi_align/functions/roi_align.py:7-31 (forward only)."""
import torch

from .... import _lib


class RoIAlignFunction(object):
    """Callable with the reference's constructor/forward signature.  Inference only:
    the backward op (roi_align_kernel.cu:94-143) is training code and out of scope."""

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        if not features.is_cuda:
            raise NotImplementedError          # functions/roi_align.py:28-29
        features = features.contiguous().float()
        rois = rois.contiguous().float()
        b, c, h, w = features.shape
        n = int(rois.shape[0])
        out = torch.zeros((n, c, self.aligned_height, self.aligned_width), dtype=torch.float32,
                          device=features.device)   # zero-filled as functions/roi_align.py:22
        # return value (1 ok / 0 bad roi shape) is ignored by the reference caller too
        _lib.lib().roi_align_forward_cuda(self.aligned_height, self.aligned_width, self.spatial_scale,
                                          features.data_ptr(), b, c, h, w, rois.data_ptr(), n,
                                          int(rois.shape[1]) if rois.dim() == 2 else 0, out.data_ptr(),
                                          _lib.stream())
        return out

    __call__ = forward
