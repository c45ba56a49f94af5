"""Multi-GPU driver pieces: stereo pairs shard embarrassingly, detections are gathered.

The reference has no parallelism at all (test.sh pins one GPU and
test_net.py loops over images with batch_size=1).  MI355X design: one process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI), pair i -> rank i mod world, weights
replicated, NO collective on the data path; one all_gather of fixed-size padded detection
records per step (~29 KB/image: latency-bound on xGMI, issued asynchronously).
"""
import torch
import torch.distributed as dist

# one row per detection: [score, left box 4, right box 4, dim_orien 5, kpts 5, roi index | 4-DoF status, x y z theta (4-DoF) |
# alignment status, disparity, final x y z theta, alpha] -- include/srcnn_hip.h SRCNN_REC_COLS; columns 20.. are filled by
# the device 3-D stage (pipeline.launch_3d) and are zero in a detector-only record
REC_COLS = 32


def shard_indices(num_items, rank, world_size):
    """Pair indices owned by `rank`: i with i % world_size == rank (order preserved)."""
    return list(range(rank, num_items, world_size))


def pack_records(cls_det, max_det=300):
    """Per-class detection dict (postprocess.class_detections) -> fixed (max_det + 1, REC_COLS)
    float32 record; row 0 holds the valid count."""
    dl = cls_det['dets_left']
    dev = dl.device
    k = min(int(dl.shape[0]), max_det)
    rec = torch.zeros((max_det + 1, REC_COLS), dtype=torch.float32, device=dev)
    rec[0, 0] = k
    if k:
        rec[1:k + 1, 0] = dl[:k, 4]
        rec[1:k + 1, 1:5] = dl[:k, :4]
        rec[1:k + 1, 5:9] = cls_det['dets_right'][:k, :4]
        rec[1:k + 1, 9:14] = cls_det['dim_orien'][:k]
        rec[1:k + 1, 14:19] = cls_det['kpts'][:k]
        rec[1:k + 1, 19] = cls_det['keep_idx'][:k].float()
    return rec


def pack_records_device(det, keep_idx, num, j=1, out=None):
    """Same record, built on the device by one native launch from the -1 padded keep list, no host sync
    (det: postprocess.decode_detections; keep_idx/num: postprocess.class_nms_device).  `out`: optional
    (n + 1, REC_COLS) float32 destination, e.g. one row block of a buffer that is gathered every few steps.
    CPU tensors (the gloo tests) take an equivalent torch path."""
    n = int(keep_idx.shape[0])
    dev = keep_idx.device
    if keep_idx.is_cuda:
        from . import _lib
        rec = torch.empty((n + 1, REC_COLS), dtype=torch.float32, device=dev) if out is None else out
        assert rec.is_contiguous() and tuple(rec.shape) == (n + 1, REC_COLS) and rec.dtype == torch.float32
        n_cls = int(det['scores'].shape[1])
        _lib.check(_lib.lib().srcnn_pack_detections(det['scores'].data_ptr(), det['boxes_left'].data_ptr(),
                                                    det['boxes_right'].data_ptr(), det['dim_orien'].data_ptr(),
                                                    det['kpts'].data_ptr(), keep_idx.data_ptr(), num.data_ptr(), n,
                                                    n_cls, j, REC_COLS, rec.data_ptr(), _lib.stream()),
                   "srcnn_pack_detections")
        return rec
    idx = keep_idx.clamp(min=0).long()
    valid = (torch.arange(n, device=dev) < num.to(torch.int64)).float().unsqueeze(1)
    body = torch.zeros((n, REC_COLS), dtype=torch.float32, device=dev)
    body[:, 0] = det['scores'][idx, j]
    body[:, 1:5] = det['boxes_left'][idx, 4 * j:4 * j + 4]
    body[:, 5:9] = det['boxes_right'][idx, 4 * j:4 * j + 4]
    body[:, 9:14] = det['dim_orien'][idx, 5 * j:5 * j + 5]
    body[:, 14:19] = det['kpts'][idx]
    body[:, 19] = idx.float()
    head = torch.zeros((1, REC_COLS), dtype=torch.float32, device=dev)
    head[0, 0] = num[0].float()
    rec = torch.cat((head, body * valid), 0)
    if out is not None:
        out.copy_(rec)
        return out
    return rec


def unpack_records(rec):
    k = int(rec[0, 0])
    body = rec[1:k + 1]
    return {'scores': body[:, 0], 'boxes_left': body[:, 1:5], 'boxes_right': body[:, 5:9],
            'dim_orien': body[:, 9:14], 'kpts': body[:, 14:19], 'roi_index': body[:, 19].long(),
            'solve_status': body[:, 20], 'pose_4dof': body[:, 21:25], 'align_status': body[:, 25], 'disparity': body[:, 26],
            'pose': body[:, 27:31], 'alpha': body[:, 31]}


def gather_detections(rec, async_op=False):
    """all_gather of one fixed-size record (or a stack of records, any leading shape) per rank ->
    (world,) + rec.shape on every rank.  Works with the gloo backend on CPU tensors (tests) and nccl/RCCL on
    device tensors."""
    if not dist.is_initialized():
        return rec.unsqueeze(0), None
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    work = dist.all_gather_into_tensor(out.view(-1), rec.contiguous().view(-1), async_op=async_op) \
        if rec.is_cuda else dist.all_gather(list(out.unbind(0)), rec.contiguous(), async_op=async_op)
    return out, work


def objects_to_record(objs, max_det=300):
    """Host-side list of final 3-D objects (pipeline.detect_3d) -> the same fixed (max_det + 1, REC_COLS) float32 record the
    device 3-D stage fills (include/srcnn_hip.h lists the columns): what a rank contributes to the gather of a split."""
    import numpy as np
    rec = np.zeros((max_det + 1, REC_COLS), np.float32)
    k = min(len(objs), max_det)
    rec[0, 0] = k
    for i, o in enumerate(objs[:k]):
        r = rec[1 + i]
        r[0] = o['score']
        r[1:5], r[5:9] = o['box_left'], o['box_right']
        r[9:12] = o['dim']
        r[12], r[13] = np.sin(o['alpha']), np.cos(o['alpha'])
        r[14:19] = o['kpts']
        r[19] = o.get('roi_index', -1)
        r[20] = 1.0
        r[21:24], r[24] = o['xyz_init'], o.get('theta_init', 0.0)
        r[25] = 1.0 if o['aligned'] else 0.0
        r[26] = o.get('disparity', 0.0)
        r[27:30], r[30] = o['xyz'], o['theta']
        r[31] = o['alpha']
    return torch.from_numpy(rec)


def gather_split_records(records, num_frames_total, rank, world):
    """records: this rank's per-frame records in shard order (frame i of the split lives on rank i % world).  Returns, on every
    rank, the (num_frames_total, max_det + 1, REC_COLS) tensor of the whole split in frame order: ONE all_gather of the padded
    per-rank stacks (ranks own ceil or floor(N / world) frames; short stacks are zero-padded)."""
    per = -(-num_frames_total // world)
    shape = tuple(records[0].shape) if records else (301, REC_COLS)
    dev = records[0].device if records else torch.device('cpu')
    stack = torch.zeros((per,) + shape, dtype=torch.float32, device=dev)
    for i, r in enumerate(records):
        stack[i] = r
    out, work = gather_detections(stack)
    if work is not None:
        work.wait()
    full = torch.zeros((num_frames_total,) + shape, dtype=torch.float32, device=dev)
    for r in range(out.shape[0]):
        idx = shard_indices(num_frames_total, r, world if dist.is_initialized() else 1)
        for j, frame in enumerate(idx):
            full[frame] = out[r, j]
    return full


# ------------------------------------------------------------------------------------------------ host side of a rank
def local_world_size():
    import os
    return max(1, int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')) or 1))


def host_solver_threads(max_threads=16):
    """Threads one rank gives its host Newton-CG solves (pipeline.HOST_SOLVER_THREADS): the ranks of a node share the host
    cores, so the budget is cpu_count / LOCAL_WORLD_SIZE, at most `max_threads` (one thread per ~8 detections is enough)."""
    import os
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    return max(1, min(max_threads, cpus // local_world_size()))


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(device_index, sysfs='/sys'):
    """CPUs of the NUMA node the GPU hangs off (PCI bus id -> /sys/bus/pci/devices/<bdf>/numa_node ->
    /sys/devices/system/node/node<N>/cpulist), or None when the platform does not say."""
    import os
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, props.pci_device_id)
        with open(os.path.join(sysfs, 'bus/pci/devices', bdf, 'numa_node')) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, 'devices/system/node/node%d/cpulist' % node)) as f:
            return node, _parse_cpulist(f.read())
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def pin_to_gpu_numa(device_index, cpus=None):
    """Restrict this process (and the solver threads it spawns: they inherit the mask) to the CPUs of its GPU's NUMA node,
    intersected with the mask it already has.  Best effort: returns a short description, or None if nothing was changed."""
    import os
    node = None
    if cpus is None:
        found = gpu_numa_cpus(device_index)
        if found is None:
            return None
        node, cpus = found
    try:
        allowed = os.sched_getaffinity(0) & set(cpus)
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
    except (AttributeError, OSError):
        return None
    return {'numa_node': node, 'cpus': len(allowed)}
