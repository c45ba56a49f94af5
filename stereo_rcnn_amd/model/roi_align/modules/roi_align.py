"""ROIAlign modules with the reference's constructor / call surface (lib/model/roi_align/modules/roi_align.py):
`RoIAlign(ah, aw, scale)`, `RoIAlignAvg`, `RoIAlignMax`, each called as `module(features, rois, scale)`.

One implementation, three pooling policies: the native lattice op samples an (ah + extra) x (aw + extra) grid of single
bilinear taps per roi; 'avg' / 'max' reduce every 2x2 neighbourhood of that grid (stride 1) back to ah x aw.  These are
the op-level drop-ins (NCHW in, NCHW out); `_StereoRCNN.forward` itself uses the fused NHWC pyramid kernel
(`srcnn_pyramid_roi_align`), which performs the lattice + average in one pass."""
import torch
import torch.nn as nn

from .... import _lib
from ..functions.roi_align import RoIAlignFunction


def _pool2x2_s1(lattice, take_max):
    """avg_pool2d / max_pool2d(kernel_size=2, stride=1) of the (n, C, h, w) lattice as a library launch (srcnn_pool2x2_s1)."""
    n, c, h, w = lattice.shape
    out = torch.empty((n, c, h - 1, w - 1), dtype=torch.float32, device=lattice.device)
    _lib.check(_lib.lib().srcnn_pool2x2_s1(lattice.data_ptr(), n * c, h, w, out.data_ptr(), int(take_max), _lib.stream()),
               "srcnn_pool2x2_s1")
    return out


class _LatticeAlign(nn.Module):
    extra = 0            # lattice points added per side before pooling
    reduce = None        # callable on the (n, C, ah + extra, aw + extra) lattice, or None

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super().__init__()
        # same attribute names as the reference modules; `spatial_scale` is stored but, as there, the scale actually used
        # is the one passed at call time (the pyramid level decides it)
        self.aligned_height, self.aligned_width = int(aligned_height), int(aligned_width)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois, scale):
        op = RoIAlignFunction(self.aligned_height + self.extra, self.aligned_width + self.extra, scale)
        lattice = op(features, rois)
        return lattice if self.reduce is None else type(self).reduce(lattice)


class RoIAlign(_LatticeAlign):
    """The bare lattice."""


class RoIAlignAvg(_LatticeAlign):
    """(A+1) x (A+1) lattice, mean of each 2x2 neighbourhood: what the Stereo R-CNN heads use."""
    extra = 1
    reduce = staticmethod(lambda t: _pool2x2_s1(t, False))


class RoIAlignMax(_LatticeAlign):
    """(A+1) x (A+1) lattice, max of each 2x2 neighbourhood."""
    extra = 1
    reduce = staticmethod(lambda t: _pool2x2_s1(t, True))
