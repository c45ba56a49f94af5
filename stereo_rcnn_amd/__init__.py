"""MI355X-native Stereo R-CNN inference path (gfx950 HIP kernels behind the
reference's Python operator surface).  See DESIGN.md."""

import os as _os

import torch as _torch

# Hosts of MI355X boxes have hundreds of cores; torch's default intra-op pool (one thread per
# core) makes the small CPU-side tensor ops of weight loading orders of magnitude slower.
if _torch.get_num_threads() > 16 and 'OMP_NUM_THREADS' not in _os.environ:
    _torch.set_num_threads(16)
