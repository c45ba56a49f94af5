"""MI355X-native Stereo R-CNN inference path (gfx950 HIP kernels behind the
reference's Python operator surface).  See DESIGN.md."""
