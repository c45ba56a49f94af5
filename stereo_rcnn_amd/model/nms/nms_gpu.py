"""`nms_gpu` (reference lib/model/nms/nms_gpu.py:7-12) over the HIP library."""
import torch

from ... import _lib


def nms_gpu(dets, thresh):
    """dets (N, >=4) float32 on the GPU, score-sorted -> keep (k, 1) int32 (device).

    Same contract as the reference: `keep`/`num_out` are allocated here on the device
    (nms_gpu.py:8-9) and the result is sliced with num_out (one D2H sync, :11).  The
    native call itself performs no host round trip (the reference's copies a 4.5 MB mask).
    """
    if not dets.is_cuda:
        raise NotImplementedError("nms_gpu needs a device tensor (the reference's CUDA op has no CPU branch)")
    dets = dets.contiguous().float()
    n, dim = int(dets.shape[0]), int(dets.shape[1])
    keep = torch.zeros((n, 1), dtype=torch.int32, device=dets.device)
    num_out = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    L = _lib.lib()
    ws = _lib.workspace(L.srcnn_nms_workspace_bytes(n), dets.device, "nms")
    _lib.check(L.srcnn_nms(keep.data_ptr(), dets.data_ptr(), num_out.data_ptr(), n, dim, float(thresh),
                           ws.data_ptr(), ws.numel(), _lib.stream()), "srcnn_nms")
    return keep[:int(num_out[0])]
