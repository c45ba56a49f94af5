"""`nms(dets, thresh, force_cpu=False)` - reference lib/model/nms/nms_wrapper.py:13-21."""
from .nms_gpu import nms_gpu


def nms(dets, thresh, force_cpu=False):
    """Returns [] for empty input (nms_wrapper.py:15-16), else IntTensor (k, 1) of kept rows.

    `force_cpu=True` selected the reference's nms_cpu, which is dead and wrong
    (np.maximum for xx2/yy2, nms_cpu.py:23-24; `.numpy()` on a CUDA tensor, :7) and is
    never taken because cfg.USE_GPU_NMS is True (config.py:196).  It is rejected here
    instead of silently computing something else.
    """
    if dets.shape[0] == 0:
        return []
    if force_cpu:
        raise NotImplementedError("force_cpu NMS is not part of the MI355X path (reference nms_cpu.py is dead code)")
    return nms_gpu(dets, thresh)
