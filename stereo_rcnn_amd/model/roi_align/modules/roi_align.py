"""`RoIAlign` / `RoIAlignAvg` modules - reference lib/model/roi_align/modules/roi_align.py:6-29."""
import torch.nn as nn
from torch.nn.functional import avg_pool2d

from ..functions.roi_align import RoIAlignFunction


class RoIAlign(nn.Module):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super(RoIAlign, self).__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois, scale):
        return RoIAlignFunction(self.aligned_height, self.aligned_width, scale)(features, rois)


class RoIAlignAvg(nn.Module):
    """(A+1)x(A+1) point lattice + 2x2/stride-1 average (roi_align.py:26-29).

    This module is the op-level drop-in (NCHW in / NCHW out).  The forward pass of
    `_StereoRCNN` uses the fused NHWC pyramid kernel (srcnn_pyramid_roi_align) instead."""

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super(RoIAlignAvg, self).__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois, scale):
        x = RoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, scale)(features, rois)
        return avg_pool2d(x, kernel_size=2, stride=1)
