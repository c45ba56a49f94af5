"""Stereo 3-D detection of one pair, end to end - the flow of the reference's demo.py:137-326
(= test_net.py:131-331): network forward, decode, per-class NMS, border inference, 4-DoF box solve,
dense alignment, 3-DoF rectification.

Every stage works in place on the image's fixed-size detection record (include/srcnn_hip.h: SRCNN_REC_COLS) through
launches into the HIP library on one stream -- class NMS -> pack -> srcnn_infer_boundary -> 4-DoF solve ->
srcnn_dense_align (masked, fixed batch) -> 3-DoF solve -- nothing per object in Python.  The solvers are scipy's
Newton-CG restated in double precision (csrc/box_solver.h), one source built for the device and for the host:
  solver='host' (default)  the two solves run on the HOST build between the device stages (record down, solve in C
                           threads, record up): results bit-identical to the reference's scipy path, because the host
                           build calls the same libm (pow, cos, sin, atan2) numpy does.  Streamed, the solves of one pair
                           hide behind the forward of the next: same throughput as 'device'.
  solver='device'          the solves are kernels (one detection per workgroup): no host round trip between the detector
                           and the final boxes, ONE device-to-host copy.  ocml's cos / sin / atan2 and exact squares
                           differ from glibc in last bits; the chaotic Newton-CG end points then differ.
  solver='scipy'           the reference's own arrangement (host numpy `infer_boundary`, scipy solves, optionally fanned
                           out to a process pool) as the comparison path: model/utils/box_estimator.py, kitti_utils.py."""
import collections
import math as m

import numpy as np
import torch

from . import _lib, postprocess
from . import streams as _streams
from .model.dense_align.dense_align import MAX_PIXELS, align_parallel, check_status
from .model.utils import box_estimator, kitti_utils
from .model.utils.config import cfg

REC_COLS = _lib.REC_COLS


class _PlainCalib(object):
    """Picklable stand-in for FrameCalibrationData: the solvers only read p2 and p3."""

    def __init__(self, p2, p3):
        self.p2, self.p3 = np.asarray(p2, dtype=np.float64), np.asarray(p3, dtype=np.float64)


def _solve_task(task):
    """One solver call (plain numpy in, plain numpy out): scipy (kind 4 / 3; possibly in a worker process) or the native
    Newton-CG compiled for the host (kind 14 / 13; in-process, ~35 us each)."""
    kind, im_shape, p2, p3, args = task
    calib = _PlainCalib(p2, p3)
    if kind in (4, 14):
        fn = box_estimator.solve_x_y_z_theta_from_kpt if kind == 4 else box_estimator.solve_x_y_z_theta_from_kpt_native
        status, state = fn(im_shape, calib, *args)
        return status, (np.asarray(state, dtype=np.float64) if status or np.ndim(state) else None)
    fn = box_estimator.solve_x_y_theta_from_kpt if kind == 3 else box_estimator.solve_x_y_theta_from_kpt_native
    state, z = fn(im_shape, calib, *args)
    return np.asarray(state, dtype=np.float64), float(z)


def _alpha32(alpha):
    """The 3-DoF rectification reads alpha back from the float32 `poses_all` tensor (demo.py:313 -> poses[7]), the 4-DoF
    solve uses the float64 atan2 result directly (demo.py:288-293): mirror both."""
    return float(np.float32(alpha))


def _f64(row):
    """The 3-DoF call takes rows of torch tensors in the reference (demo.py:312-318: boxes_all / kpts_all slices), whose elements
    are Python floats under torch 0.3: all arithmetic on them is double, unlike the numpy float32 rows of the 4-DoF call."""
    return np.asarray(row, dtype=np.float64)


def _scale32(im_info):
    """im_info[0, 2] as the reference reads it: a float32 tensor element turned into a Python float."""
    return float(im_info.view(-1, 3)[0, 2])


def _noop(_):
    return None


def _worker_init():
    import os
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    torch.set_num_threads(1)


class SolverPool(object):
    """Process pool for the scipy comparison path (`solver='scipy'`).  Each scipy solve is ~2 ms of single-threaded Python;
    workers are spawned (not forked: the parent owns a HIP context), come up single-threaded and never touch the GPU.
    `workers=None`: cpu_count / world size (ranks of one node share the host cores), at most 32 -- more is slower
    (profiles/full_pipeline_solver_pool_r01.txt)."""

    def __init__(self, workers=None):
        import multiprocessing as mp
        import os
        if workers is None:
            world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')) or 1)
            workers = max(1, min(32, (os.cpu_count() or 1) // max(world, 1)))
        self.workers = int(workers)
        keys = ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS')
        saved = {k: os.environ.get(k) for k in keys}
        os.environ.update({k: '1' for k in keys})
        try:
            self._pool = mp.get_context('spawn').Pool(self.workers, initializer=_worker_init)
            self._pool.map(_noop, range(self.workers))          # workers are up (and have imported) before the first pair
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def map(self, tasks):
        if not tasks:
            return []
        return self._pool.map(_solve_task, tasks, chunksize=-(-len(tasks) // self._pool._processes))   # one message per worker

    def submit(self, tasks):
        """Asynchronous form of map(): returns an object whose .get() yields the results."""
        n = max(len(tasks), 1)
        return self._pool.map_async(_solve_task, tasks, chunksize=-(-n // self._pool._processes))

    def close(self):
        self._pool.close()
        self._pool.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ------------------------------------------------------------------------------------------------ device flow
class _Stage3D(object):
    """Device + pinned host buffers of the 3-D stage for one in-flight pair (n = rois per image)."""

    def __init__(self, n, im_w, dev):
        L = _lib.lib()
        self.n = n
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.rec = f32(n + 1, REC_COLS)
        self.state = torch.empty((2, n, 4), dtype=torch.float64, device=dev)       # [0] 4-DoF, [1] final
        self.boxes, self.borders, self.poses, self.valid = f32(n, 4), f32(n, 2), f32(n, 7), f32(n)
        self.align_status, self.best_dis = f32(n), f32(n)
        self.ws = torch.empty(int(L.srcnn_box3d_workspace_bytes(n, 4096)), dtype=torch.uint8, device=dev)
        self.rec_host = torch.empty((n + 1, REC_COLS), dtype=torch.float32, pin_memory=True)
        self.state_host = torch.empty((2, n, 4), dtype=torch.float64, pin_memory=True)
        self.align_host = torch.empty((2, n), dtype=torch.float32, pin_memory=True)     # solver='host': status, disparity
        self.event = torch.cuda.Event()
        self.phase = 0               # solver='host': 1 = waiting for the 4-DoF solve, 2 = for the 3-DoF solve, 0 = complete
        self.ctx = None
        self.worker, self.error, self.done = False, None, None     # streamed flow: a worker thread runs the host phases


_stages = {}


def _stage(n, dev, slot):
    key = (str(dev), n, slot)
    if key not in _stages:
        _stages[key] = _Stage3D(n, 4096, dev)
    return _stages[key]


# Keypoint branch on the detections that survive class NMS only (Plan.kpts_for_kept) instead of on all 300 rois inside the
# forward: the flows below read no other row (nor do the reference's scripts, demo.py:196-257), every roi's keypoints are
# computed independently of the other rois, and the device-side keep count bounds the tower's launches without a host
# read-back.  The kept detections get the full head's values up to the engine's plan-to-plan rounding (~1e-5 relative on
# the probabilities: the row-limited launches are tuned to their own tile / split-K plans).  Off: the forward computes the
# branch for every roi, as `_StereoRCNN.forward` does by default.
import os as _os
LAZY_KPTS = _os.environ.get('SRCNN_LAZY_KPTS', '1') != '0'


def _lazy(model):
    return LAZY_KPTS and model.precision == 'f16x3' and not model.use_graph


def launch_3d(out, im_left_data, im_right_data, im_info, scale, calib, im_shape, eval_thresh=0.05, class_index=1,
              dense_align=True, slot=0, solver='device', lazy=None, async_host=False):
    """Everything after the forward, asynchronously on the current stream.  out: the forward's tuple; scale: im_info[0, 2] as a
    Python float (passing it spares a device read).  Returns a handle for collect_3d().
    solver='host': only class NMS, record and borders are launched here; the two Newton-CG solves then run on the HOST (the
    same row functions built for the host, bit-identical to the reference's scipy path) in step_3d() / collect_3d(), with
    the dense alignment on the device in between.
    lazy=(plan, precision): `out` comes from forward(kpts=False); the keypoint branch runs here, on the kept detections."""
    L = _lib.lib()
    det = postprocess.decode_detections(*out[:8], im_info)
    keep_idx, num = postprocess.class_nms_device(det, class_index, eval_thresh, cfg.TEST.NMS)
    n = int(keep_idx.shape[0])
    if lazy is not None:
        lazy[0].kpts_for_kept(out[0][0].contiguous(), keep_idx, num, im_info.view(-1, 3)[0:1].contiguous().float(), det['kpts'], lazy[1])
    assert int(im_shape[1]) <= 4095, "image wider than the 3-D stage's column buffer"
    st = _stage(n, keep_idx.device, slot)
    from . import distributed as sdist
    sdist.pack_records_device(det, keep_idx, num, class_index, out=st.rec)
    s = _lib.stream()
    cal = (float(calib.p2[0, 0]), float(calib.p2[0, 2]), float(calib.p2[1, 2]), float(calib.p2[0, 3] - calib.p3[0, 3]))
    im_h, im_w = int(im_shape[0]), int(im_shape[1])
    _lib.check(L.srcnn_infer_boundary(st.rec.data_ptr(), n, REC_COLS, im_w, st.ws.data_ptr(), st.ws.numel(), s),
               "srcnn_infer_boundary")
    st.phase, st.ctx = 0, None
    if solver == 'host':
        st.rec_host.copy_(st.rec, non_blocking=True)
        st.event.record()
        st.phase = 1
        st.ctx = (torch.cuda.current_stream(), im_left_data, im_right_data, float(scale), cal, im_h, im_w, float(eval_thresh),
                  bool(dense_align))
        st.worker, st.error = False, None
        if async_host and ASYNC_HOST_PHASES:
            st.done = _threading.Event()
            st.worker = True
            _host_executor().submit(_host_phases, st, keep_idx.device.index if keep_idx.device.index is not None else torch.cuda.current_device())
        return st
    assert solver == 'device', solver
    _lib.check(L.srcnn_solve_4dof(st.rec.data_ptr(), n, REC_COLS, im_h, im_w, cal[0], cal[1], cal[2], cal[3],
                                  float(eval_thresh), st.state[0].data_ptr(), s), "srcnn_solve_4dof")
    if dense_align:
        _lib.check(L.srcnn_align_inputs(st.rec.data_ptr(), n, REC_COLS, st.boxes.data_ptr(), st.borders.data_ptr(),
                                        st.poses.data_ptr(), st.valid.data_ptr(), s), "srcnn_align_inputs")
        _, _, H, W = im_left_data.shape
        ws = _lib.workspace(L.srcnn_dense_align_workspace_bytes(H, W, n, MAX_PIXELS), im_left_data.device, "dense_align")
        _lib.check(L.srcnn_dense_align(_lib.ptr(im_left_data), _lib.ptr(im_right_data), H, W, float(scale), cal[0], cal[1],
                                       cal[2], cal[3], st.boxes.data_ptr(), st.borders.data_ptr(), st.poses.data_ptr(),
                                       st.valid.data_ptr(), n, MAX_PIXELS, st.align_status.data_ptr(),
                                       st.best_dis.data_ptr(), ws.data_ptr(), ws.numel(), s), "srcnn_dense_align")
        _lib.check(L.srcnn_solve_3dof(st.rec.data_ptr(), n, REC_COLS, im_h, im_w, cal[0], cal[1], cal[2], cal[3],
                                      st.align_status.data_ptr(), st.best_dis.data_ptr(), st.state[1].data_ptr(), s),
                   "srcnn_solve_3dof")
    st.rec_host.copy_(st.rec, non_blocking=True)
    st.state_host.copy_(st.state, non_blocking=True)
    st.event.record()
    return st


HOST_SOLVER_THREADS = 0          # srcnn_solve_*_records_host: <= 0 = one thread per 8 detections, at most 16
# Streamed flow, solver='host' (default; SRCNN_ASYNC_HOST=0 = on the loop thread between launches): the host phases of a pair (wait for
# its device stage, 4-DoF solves, record back + dense alignment launch, wait, 3-DoF solves) on a WORKER thread per pair in flight.
# Same calls on the same streams in the same order per pair either way.  History of the switch (round 6): built for VERDICT r5 item
# 8 and first measured with the KITTI loop's 16 PNG decoders as threads of this interpreter -- the loop thread's busy time per pair
# fell 3.55 -> 2.14 ms and the throughput did not move (configs[3] 138.1 -> 136.7 pairs/s, 123 at six in flight: workers, decoders
# and loop on one GIL; profiles/flow3d_async_host_r06.txt), so it stayed off.  With the decoders in processes of their own
# (test_net._DecodeWorkers) the same switch gives configs[3] 155.6-163.4 -> 169.7-170.0 pairs/s on one box and leaves the
# tensor-fed flow where it was (GPU-bound at 5.7 ms per pair: 175.2 vs 175.5) -- profiles/config3_host_side_r06.txt.
ASYNC_HOST_PHASES = _os.environ.get('SRCNN_ASYNC_HOST', '1') != '0'
_host_pool = None
import threading as _threading
_timers_lock = _threading.Lock()


def _host_executor():
    global _host_pool
    if _host_pool is None:
        import concurrent.futures as cf
        _host_pool = cf.ThreadPoolExecutor(max_workers=8, thread_name_prefix='srcnn-host3d')
    return _host_pool


def _host_phases(st, dev_index):
    """Worker thread: every remaining host phase of one pair, blocking on its own device work only."""
    try:
        with torch.cuda.device(dev_index):
            while st.phase:
                step_3d(st, block=True, _worker=True)
    except BaseException as e:          # re-raised by collect_3d on the caller's thread
        st.error = e
    finally:
        st.done.set()
# host-side accounting of the record flow (bench.py --config 3): a dict {'solve_s', 'gpu_wait_s'} that step_3d / collect_3d add
# to -- wall seconds of the host Newton-CG calls, and seconds the host sat in event.synchronize() waiting for the device
TIMERS = None


def _timed(key, t0):
    if TIMERS is not None:
        import time
        dt = time.perf_counter() - t0
        with _timers_lock:
            if TIMERS is not None:
                TIMERS[key] = TIMERS.get(key, 0.0) + dt


def _now():
    import time
    return time.perf_counter() if TIMERS is not None else 0.0


def step_3d(st, block=True, _worker=False):
    """solver='host' handles: run the next host phase.  Phase 1: 4-DoF solves on the pinned record, record back to the device,
    dense alignment launched.  Phase 2: 3-DoF solves.  block=True waits for the device work the phase needs; block=False (the
    streamed flow's opportunistic pass over the pairs in flight) returns at once when that work has not finished -- the host then
    goes on launching the next pair's forward instead of sitting in front of an event, and the phase is run on a later pass or,
    at the latest, when the pair's slot is needed (collect_3d blocks)."""
    if st.phase == 0:
        return
    if getattr(st, 'worker', False) and not _worker:
        return                                       # a worker thread owns this pair's host phases (collect_3d waits for it)
    if not block and not st.event.query():
        return
    L = _lib.lib()
    stream, iml, imr, scale, cal, im_h, im_w, thresh, dense = st.ctx
    n = st.n
    t0 = _now()
    st.event.synchronize()
    _timed('gpu_wait_s', t0)
    if st.phase == 1:
        if st.rec_host[0, 1] > 0:                              # range guard: collect_3d raises
            st.phase = 0
            return
        t0 = _now()
        _lib.check(L.srcnn_solve_4dof_records_host(st.rec_host.data_ptr(), n, REC_COLS, im_h, im_w, cal[0], cal[1], cal[2],
                                                   cal[3], thresh, st.state_host[0].data_ptr(), HOST_SOLVER_THREADS),
                   "srcnn_solve_4dof_records_host")
        _timed('solve_s', t0)
        st.state_host[1].zero_()
        if not dense or not bool((st.rec_host[1:1 + int(st.rec_host[0, 0]), 20] > 0).any()):
            st.phase = 0
            return
        with torch.no_grad(), torch.cuda.stream(stream):
            s = _lib.stream()
            st.rec.copy_(st.rec_host, non_blocking=True)
            _lib.check(L.srcnn_align_inputs(st.rec.data_ptr(), n, REC_COLS, st.boxes.data_ptr(), st.borders.data_ptr(),
                                            st.poses.data_ptr(), st.valid.data_ptr(), s), "srcnn_align_inputs")
            _, _, H, W = iml.shape
            ws = _lib.workspace(L.srcnn_dense_align_workspace_bytes(H, W, n, MAX_PIXELS), iml.device, "dense_align")
            _lib.check(L.srcnn_dense_align(_lib.ptr(iml), _lib.ptr(imr), H, W, scale, cal[0], cal[1], cal[2], cal[3],
                                           st.boxes.data_ptr(), st.borders.data_ptr(), st.poses.data_ptr(), st.valid.data_ptr(),
                                           n, MAX_PIXELS, st.align_status.data_ptr(), st.best_dis.data_ptr(), ws.data_ptr(),
                                           ws.numel(), s), "srcnn_dense_align")
            st.align_host[0].copy_(st.align_status, non_blocking=True)
            st.align_host[1].copy_(st.best_dis, non_blocking=True)
            st.event.record()
        st.phase = 2
        return
    t0 = _now()
    _lib.check(L.srcnn_solve_3dof_records_host(st.rec_host.data_ptr(), n, REC_COLS, im_h, im_w, cal[0], cal[1], cal[2], cal[3],
                                               st.align_host[0].data_ptr(), st.align_host[1].data_ptr(),
                                               st.state_host[1].data_ptr(), HOST_SOLVER_THREADS),
               "srcnn_solve_3dof_records_host")
    _timed('solve_s', t0)
    st.phase = 0


def collect_3d(st):
    """Wait for a launch_3d() handle and turn its record into the object list detect_3d returns."""
    if getattr(st, 'worker', False):
        t0 = _now()
        st.done.wait()
        _timed('main_wait_s', t0)
        st.worker = False
        if st.error is not None:
            err, st.error = st.error, None
            raise err
    while st.phase:
        step_3d(st)
    st.ctx = None
    t0 = _now()
    st.event.synchronize()
    _timed('gpu_wait_s', t0)
    rec, state = st.rec_host.numpy(), st.state_host.numpy()
    if rec[0, 1] > 0:           # SPLIT16 range guard of THIS pair's forward (its plan's own word, copied and cleared by the pack)
        from . import engine
        raise engine.Split16RangeError('SPLIT16 range exceeded in %s'
                                       % engine.TAG_NAMES.get(int(rec[0, 1]), 'layer tag %d' % (int(rec[0, 1]) - 1)))
    head = getattr(st, 'batch_head', None)
    if head is not None:
        # image b > 0 of a batched forward: the ONE forward's flag word was copied (and cleared) by the pack of image 0 only, so
        # this record alone would look clean whatever the shared forward did -- in whatever order the handles are collected
        head.event.synchronize()
        hflag = int(head.rec_host[0, 1])
        if hflag > 0:
            from . import engine
            raise engine.Split16RangeError('SPLIT16 range exceeded in %s (flag of the batch\'s shared forward, carried by its first image)'
                                           % engine.TAG_NAMES.get(hflag, 'layer tag %d' % (hflag - 1)))
    k = int(rec[0, 0])
    objs = []
    for i in range(k):
        row = rec[1 + i]
        if not row[20] > 0:                                    # 4-DoF status (demo.py:293)
            continue
        if row[25] < 0:
            check_status(row[25:26])                           # lattice overflow: raises
        aligned = bool(row[25] > 0)
        xyz4 = state[0, i, 0:3].copy()
        o = {'box_left': row[1:5].copy(), 'box_right': row[5:9].copy(), 'score': float(row[0]),
             'dim': row[9:12].astype(np.float64), 'alpha': m.atan2(float(row[12]), float(row[13])),
             'xyz': state[1, i, 0:3].copy() if aligned else xyz4, 'theta': float(state[1, i, 3] if aligned else state[0, i, 3]),
             'kpts': row[14:19].copy(), 'aligned': aligned, 'xyz_init': xyz4, 'theta_init': float(state[0, i, 3]),
             'roi_index': int(row[19])}
        if aligned:
            o['disparity'] = float(row[26])
        objs.append(o)
    return objs


# ------------------------------------------------------------------------------------------------ scipy comparison flow
def _detect_3d_scipy(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh, class_index, dense_align, pool,
                     native=False):
    k4, k3 = (14, 13) if native else (4, 3)
    with torch.no_grad():
        lazy = _lazy(model)                   # the same form of the keypoint branch as the record flows (see LAZY_KPTS)
        out = model(im_left_data, im_right_data, im_info, kpts=not lazy, alias_outputs=True)
        det = postprocess.decode_detections(*out[:8], im_info)
        if lazy:
            keep_idx, num = postprocess.class_nms_device(det, class_index, eval_thresh, cfg.TEST.NMS)
            plan, precision = _plan_of(model, im_left_data, 0)
            plan.kpts_for_kept(out[0][0].contiguous(), keep_idx, num, im_info.view(-1, 3)[0:1].contiguous(), det['kpts'], precision)
        cls = postprocess.class_detections(det, class_index, eval_thresh, cfg.TEST.NMS)
        # this flow packs no record, so nothing else would read (and clear) the forward's range word: a tripped flag would be
        # charged to the next record-flow forward on this plan, and THIS result would come from out-of-range activations
        if getattr(model, 'precision', 'f32') == 'f16x3':
            model.check_range(reset=True)
    dets_left = cls['dets_left'].cpu().numpy()
    if dets_left.shape[0] == 0:
        return []
    dets_right = cls['dets_right'].cpu().numpy()
    dim_orien = cls['dim_orien'].cpu().numpy()
    kpts = cls['kpts'].cpu().numpy().copy()
    # demo.py:259-265: replace the regressed borders when they are narrower than half the inferred ones
    inferred = kitti_utils.infer_boundary(im_shape, dets_left)
    for i in range(dets_left.shape[0]):
        if kpts[i, 4] - kpts[i, 3] < 0.5 * (inferred[i, 1] - inferred[i, 0]):
            kpts[i, 3:5] = inferred[i]
    run = pool.map if pool is not None else (lambda tasks: [_solve_task(t) for t in tasks])
    cand = [i for i in range(dets_left.shape[0]) if dets_left[i, -1] > eval_thresh]            # demo.py:282-283
    alphas = [m.atan2(dim_orien[i, 3], dim_orien[i, 4]) for i in cand]
    res4 = run([(k4, tuple(im_shape), calib.p2, calib.p3,
                 (a, dim_orien[i, 0:3], dets_left[i, 0:4], dets_right[i, 0:4], kpts[i])) for i, a in zip(cand, alphas)])
    solved = []
    for i, alpha, (status, state) in zip(cand, alphas, res4):                                    # demo.py:291-302
        if status > 0:
            solved.append({'box_left': dets_left[i, 0:4].copy(), 'box_right': dets_right[i, 0:4].copy(),
                           'score': float(dets_left[i, 4]), 'dim': dim_orien[i, 0:3].astype(np.float64), 'alpha': alpha,
                           'xyz': np.array(state[0:3], dtype=np.float64), 'theta': float(state[3]),
                           'kpts': kpts[i].copy(), 'aligned': False,
                           'xyz_init': np.array(state[0:3], dtype=np.float64), 'theta_init': float(state[3])})
    if not solved or not dense_align:
        return solved
    dev = im_left_data.device
    f32 = lambda rows: torch.tensor(np.asarray(rows), dtype=torch.float32, device=dev)
    boxes = f32([o['box_left'] for o in solved])
    kp = f32([o['kpts'] for o in solved])
    poses = f32([[o['xyz'][0], o['xyz'][1], o['xyz'][2], o['dim'][0], o['dim'][1], o['dim'][2], o['theta']]
                 for o in solved])
    succ, dis_final = align_parallel(calib, _scale32(im_info), im_left_data, im_right_data, boxes, kp, poses)   # demo.py:306-308
    succ, dis_final = check_status(succ.cpu().numpy()), dis_final.cpu().numpy()
    todo = [i for i in range(len(solved)) if succ[i] > 0]                                      # demo.py:311-319
    res3 = run([(k3, tuple(im_shape), calib.p2, calib.p3,
                 (_alpha32(solved[i]['alpha']), solved[i]['dim'], _f64(solved[i]['box_left']), float(dis_final[i]),
                  _f64(solved[i]['kpts'])))
                for i in todo])
    for i, (state, z) in zip(todo, res3):
        o = solved[i]
        o['xyz'] = np.array([state[0], state[1], z], dtype=np.float64)
        o['theta'] = float(state[2])
        o['aligned'] = True
        o['disparity'] = float(dis_final[i])
    return solved


# ------------------------------------------------------------------------------------------------ public entry points
def detect_3d(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh=0.05, class_index=1,
              dense_align=True, pool=None, solver='host', slot=0):
    """One preprocessed pair -> list of dicts (one per solved object, descending score):
    box_left (4), box_right (4), score, dim (w,h,l), alpha, xyz (3), theta, aligned (bool), xyz_init / theta_init (the 4-DoF
    solve), disparity (aligned objects), kpts (5, borders after the inference step).
    solver: 'host' (default: device record flow, the two Newton-CG solves on the host in C -- final boxes BIT-IDENTICAL to
    the reference's scipy path, same throughput as 'device' when streamed); 'device' (Newton-CG kernels, no host bounce at
    all, one D2H copy; its libm differs from glibc in last bits, so chaotic end points differ); 'scipy' (the
    reference's own arrangement: host numpy + scipy per object, optional `pool`); 'host_py' (that arrangement with the native
    solver called per object)."""
    from . import engine
    if solver in ('scipy', 'host_py'):
        try:
            return _detect_3d_scipy(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh, class_index,
                                    dense_align, pool if solver == 'scipy' else None, native=(solver == 'host_py'))
        except engine.Split16RangeError:
            if model.precision == 'f32':
                raise
            prev, model.precision = model.precision, 'f32'       # same fallback as the record flows below
            try:
                objs = _detect_3d_scipy(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh, class_index,
                                        dense_align, pool if solver == 'scipy' else None, native=(solver == 'host_py'))
            finally:
                model.precision = prev
            _note_guard_trip(model, im_left_data, 0)             # only after a re-run that succeeded (its own error is not masked)
            return objs
    with torch.no_grad():
        lazy = _lazy(model)
        scale = _scale32(im_info)            # a blocking device read: before the forward is enqueued, not behind it
        out = model(im_left_data, im_right_data, im_info, slot=slot, kpts=not lazy, alias_outputs=True)
        st = launch_3d(out, im_left_data, im_right_data, im_info, scale, calib, im_shape, eval_thresh,
                       class_index, dense_align, slot, solver, lazy=_plan_of(model, im_left_data, slot) if lazy else None)
    try:
        return collect_3d(st)
    except engine.Split16RangeError:
        if model.precision == 'f32':
            raise
        # an activation left the f16 range: this pair again on the exact fp32 engine (F32 activations), never garbage
        prev, model.precision = model.precision, 'f32'
        try:
            objs = detect_3d(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh, class_index,
                             dense_align, pool, solver, slot)
        finally:
            model.precision = prev
        _note_guard_trip(model, im_left_data, slot)
        return objs


def _plan_of(model, im_left_data, slot):
    B, _, H, W = im_left_data.shape
    return model._get_plan(int(B), int(H), int(W), slot), model.precision


def detect_3d_images(model, img_left_u8, img_right_u8, calib, eval_thresh=0.05, class_index=1, dense_align=True, slot=0,
                     wait=True, solver='host', async_host=False):
    """The same from the decoded uint8 RGB images on the device: preprocessing fused in front of the forward
    (model.forward_images), then the device 3-D flow.  wait=False returns the handle for collect_3d()."""
    with torch.no_grad():
        lazy = _lazy(model)
        out, iml, imr, info = model.forward_images(img_left_u8, img_right_u8, slot=slot, kpts=not lazy, alias_outputs=True)
        from . import engine
        scale = float(np.float32(engine.preprocess_size(int(img_left_u8.shape[0]), int(img_left_u8.shape[1]),
                                                        cfg.TEST.SCALES[0])[2]))
        st = launch_3d(out, iml, imr, info, scale, calib, tuple(img_left_u8.shape), eval_thresh, class_index, dense_align, slot,
                       solver, lazy=_plan_of(model, iml, slot) if lazy else None, async_host=async_host and not wait)
    return collect_3d(st) if wait else st


def image_outputs(out, b):
    """The forward's outputs of image b of a batch as a batch of one (what launch_3d / decode take): the roi-major head
    outputs (kpts / border probabilities) are rows [b * n, (b + 1) * n)."""
    n = int(out[0].shape[1])
    if out[5] is None:          # forward(kpts=False)
        return (out[0][b:b + 1], out[1][b:b + 1], out[2][b:b + 1], out[3][b:b + 1], out[4][b:b + 1], None, None, None)
    return (out[0][b:b + 1], out[1][b:b + 1], out[2][b:b + 1], out[3][b:b + 1], out[4][b:b + 1],
            out[5][b * n:(b + 1) * n], out[6][b * n:(b + 1) * n], out[7][b * n:(b + 1) * n])


def launch_3d_batch(model, im_left_data, im_right_data, im_info, calibs, im_shapes, eval_thresh=0.05, class_index=1,
                    dense_align=True, slot=0, solver='host', scales=None):
    """BASELINE configs[2] form: ONE forward over a batch of B pairs (rois carry the batch index, proposal_layer.py:139), then
    the 3-D stage of every image of the batch (the reference's post-processing is per image: demo.py:153,212) enqueued on the
    current stream, each with its own stage buffers.  Returns the handles for collect_3d_batch().

    scales: the B resize factors (im_info[:, 2]) as Python floats when the caller has them on the host.  Otherwise they are read
    from the device BEFORE the forward is enqueued: reading im_info[b, 2] per image afterwards, as this function did until round
    6, is a blocking copy behind the whole forward and every earlier image's 3-D stage -- eight of them made the batch form
    host-serialised (53 of its 61 ms per batch, tools/batch_host_probe.py)."""
    B = int(im_left_data.shape[0])
    if scales is None:
        scales = [float(v) for v in im_info.view(-1, 3)[:, 2].cpu()]        # float32 elements -> Python floats, as _scale32
    with torch.no_grad():
        lazy = _lazy(model)
        out = model(im_left_data, im_right_data, im_info, slot=slot, kpts=not lazy, alias_outputs=True)
        pl = _plan_of(model, im_left_data, slot) if lazy else None
        handles = []
        for b in range(B):
            info_b = im_info.view(-1, 3)[b:b + 1]
            handles.append(launch_3d(image_outputs(out, b), im_left_data[b:b + 1], im_right_data[b:b + 1], info_b,
                                     scales[b], calibs[b], im_shapes[b], eval_thresh, class_index, dense_align,
                                     (slot, b), solver, lazy=pl))
        for h in handles[1:]:
            h.batch_head = handles[0]        # see collect_3d: the shared forward's range flag lives in image 0's record
    return handles


def collect_3d_batch(handles):
    """Object lists of the batch, one per image.  The batch shares one forward and therefore one range flag: the record of the
    first image carries it (the pack clears the word), and a tripped flag condemns the whole batch -- every other handle of the
    batch looks at image 0's record too (`batch_head`), so per-handle collect_3d() in any order raises as well.

    solver='host': the host phases run image-major per phase, not phase-major per image -- every image's 4-DoF solves and
    dense-alignment launch first, then every image's 3-DoF solves -- so the host solves image b + 1 while the device aligns image
    b, instead of sitting in front of each alignment's event in turn (eight waits of a busy GPU's queueing latency per batch)."""
    for phase in (1, 2):
        for st in handles:
            if st.phase == phase:
                step_3d(st)
    return [collect_3d(st) for st in handles]


def detect_3d_batch(model, im_left_data, im_right_data, im_info, calibs, im_shapes, eval_thresh=0.05, class_index=1,
                    dense_align=True, slot=0, solver='host'):
    """B pairs -> B object lists (detect_3d's dicts), one batched forward; out-of-range SPLIT16 activations re-run the batch on
    the exact fp32 engine."""
    from . import engine
    try:
        return collect_3d_batch(launch_3d_batch(model, im_left_data, im_right_data, im_info, calibs, im_shapes, eval_thresh,
                                                class_index, dense_align, slot, solver))
    except engine.Split16RangeError:
        if model.precision == 'f32':
            raise
        prev, model.precision = model.precision, 'f32'
        try:
            objs = detect_3d_batch(model, im_left_data, im_right_data, im_info, calibs, im_shapes, eval_thresh, class_index,
                                   dense_align, slot, solver)
        finally:
            model.precision = prev
        _note_guard_trip(model, im_left_data, slot)
        return objs


RECALIBRATE_AFTER_TRIPS = 2      # range-guard trips (pairs re-run on the fp32 engine) after which the scales are widened
# Widening the scales changes the low-order bits of EVERY later frame (one glitched frame would cost them for good), so it can be
# switched off: SRCNN_AUTO_RECALIBRATE=0 (or pipeline.AUTO_RECALIBRATE = False) keeps counting trips -- guard_trips(model) -- and
# leaves the decision to the caller (model.calibrate_activation_scales(frames) is the controlled way; reset_guard_trips(model)).
AUTO_RECALIBRATE = _os.environ.get('SRCNN_AUTO_RECALIBRATE', '1') != '0'
MAX_SHIFT_WIDENING = 10          # automatic re-calibration never lowers a tensor group's shift by more than this many bits in total (x1000)


def guard_trips(model):
    """Pairs re-run on the fp32 engine since the last calibration / reset (0 if the model has no engine-side weights yet)."""
    return int(getattr(model._weights, 'guard_trips', 0)) if model._weights is not None else 0


def reset_guard_trips(model):
    if model._weights is not None:
        model._weights.guard_trips = 0


def _note_guard_trip(model, im_left_data, slot, may_recalibrate=True):
    """A pair left the f16 range of the SPLIT16 engine and was re-run (successfully) on the exact fp32 engine.  The activation
    scales come from the frames the calibration saw (the first forward, unless calibrate_activation_scales was called): a stream of
    frames unlike them would silently run at fp32-engine speed for good.  After RECALIBRATE_AFTER_TRIPS trips -- if AUTO_RECALIBRATE
    and the caller is not streaming -- the offending frame, still the input of this slot's plan, is merged into the calibration (one
    fp32 forward; plans re-record their launch programs), unless that would widen some group by more than MAX_SHIFT_WIDENING bits
    beyond its first calibration: then the frame is an outlier, the scales stay, and only the counter says so."""
    import logging
    log = logging.getLogger('stereo_rcnn_amd')
    w = model._weights
    if w is None:
        return
    w.guard_trips = getattr(w, 'guard_trips', 0) + 1
    log.warning('SPLIT16 range guard tripped (%d since the last calibration): pair re-run on the fp32 engine', w.guard_trips)
    if w.guard_trips >= RECALIBRATE_AFTER_TRIPS and AUTO_RECALIBRATE and may_recalibrate:
        first = getattr(w, 'first_shifts', None)
        if first is None:
            first = w.first_shifts = dict(w.shifts)
        before = (dict(w.shifts), list(w.fuse_shortcut), dict(w.calibration_max), w.calib_epoch)
        B, _, H, W = im_left_data.shape
        plan = model._get_plan(int(B), int(H), int(W), slot)
        with torch.no_grad():
            plan.calibrate(merge=True)
        if any(first.get(g, k) - k > MAX_SHIFT_WIDENING for g, k in w.shifts.items()):
            changed = w.shifts != before[0] or w.fuse_shortcut != before[1]
            w.shifts, w.fuse_shortcut, w.calibration_max = before[0], before[1], before[2]
            if changed:
                w.calib_epoch += 1             # (calibrate() bumped it for the widened scales: the restored ones are another epoch)
            # a rejected widening is remembered -- the trip counter starts over, so an outlier frame costs one fp32
            # re-run per trip, not a calibration forward + a device sync + re-recorded launch programs on every one
            w.guard_trips = 0
            w.rejected_widenings = getattr(w, 'rejected_widenings', 0) + 1
            log.warning('SPLIT16 activation scales NOT widened: the offending frame needs more than %d bits beyond the first calibration '
                        '(an outlier frame); it keeps running on the fp32 engine', MAX_SHIFT_WIDENING)
            return
        w.guard_trips = 0
        log.warning('SPLIT16 activation scales widened to cover the offending frame (frames much smaller than it lose low-order bits: '
                    'calibrate_activation_scales() on representative frames is the controlled way)')


def _slot_streams(n):
    """The in-flight slots' HIP streams: the process-wide set of streams.main_streams (created once per device: per-stream scratch
    buffers (_lib.workspace) and recorded launch programs are keyed by stream, so fresh streams on every call would grow them
    without bound -- and fresh streams per CALLER would run out of hardware queues, see streams.main_streams)."""
    # the serving regime of bench.py's headline (serving.py): branches stay on the main streams with n > 1, the hardware-queue
    # supply is checked, and the shipped throughput-tuned conv plans are adopted (once; MI355X only; matching shapes only)
    from . import serving
    serving.enter(n)
    return _streams.main_streams(n, kind=_streams.MAIN_KIND if _streams.MAIN_KIND in _streams.KINDS else 'dedicated')


def detect_3d_stream(model, frames, pool=None, eval_thresh=0.05, class_index=1, dense_align=True, slots=4, solver='host'):
    """Generator form for a sequence of pairs: yields one object list per frame, in order, with up to `slots` pairs in flight
    on their own HIP streams.  frames: iterable of (im_left_data, im_right_data, im_info, calib, im_shape[, scale]) with device
    tensors, or (img_left_u8, img_right_u8, calib) with uint8 device images -- or PAGE-LOCKED host images, which the fused
    preprocessing kernel reads where they are (no copy in any stream; they must stay unmodified until the pair is yielded).  Per pair the results are
    those of detect_3d (same launches).  solver='host': the Newton-CG solves of the pairs in flight run on the host between
    the launches (a phase of every pair whose device work has finished, per new frame: never waiting), use slots >= 4 (measured:
    profiles/config3_plans_slots_r05.txt).  solver='scipy' (needs `pool`) keeps the staged
    host/scipy arrangement."""
    prev_n = _streams.pairs_in_flight()
    try:
        for objs in _detect_3d_stream(model, frames, pool, eval_thresh, class_index, dense_align, slots, solver):
            yield objs
    finally:
        _streams.set_pairs_in_flight(prev_n)      # the plans go back to the caller's regime (one at a time: branches on side streams)


def _detect_3d_stream(model, frames, pool, eval_thresh, class_index, dense_align, slots, solver):
    if solver == 'scipy':
        for objs in _detect_3d_stream_scipy(model, frames, pool, eval_thresh, class_index, dense_align, min(slots, 2)):
            yield objs
        return
    from . import engine
    streams = _slot_streams(max(1, slots))
    inflight = collections.deque()

    retry_slot = len(streams)        # a buffer set no pair in flight can own: the fp32 re-run must not touch slots 0..n-1

    def finish(entry):
        st, frame = entry
        try:
            return collect_3d(st)
        except engine.Split16RangeError:
            if model.precision == 'f32':
                raise
            # the SPLIT16 range guard of THIS pair tripped: the pair again on the exact fp32 engine, on the caller's stream and
            # in its own slot -- the pairs still in flight keep their plans, 3-D stage buffers and images untouched
            prev, model.precision = model.precision, 'f32'
            try:
                if len(frame) == 3:
                    objs = detect_3d_images(model, frame[0], frame[1], frame[2], eval_thresh, class_index, dense_align,
                                            retry_slot, solver=solver)
                else:
                    objs = detect_3d(model, frame[0], frame[1], frame[2], frame[3], frame[4], eval_thresh, class_index, dense_align,
                                     solver=solver, slot=retry_slot)
            finally:
                model.precision = prev
            if len(frame) != 3:
                # streamed: other pairs are in flight -- count the trip, never re-calibrate here (that synchronises the device and
                # changes the scales under the pairs in flight); the caller re-calibrates between streams if it wants to
                _note_guard_trip(model, frame[0], retry_slot, may_recalibrate=False)
            return objs

    for k, frame in enumerate(frames):
        slot = k % len(streams)
        if len(inflight) == len(streams):                      # the slot's buffers are still in use by the oldest pair
            yield finish(inflight.popleft())
        s = streams[slot]
        s.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(s):
            if len(frame) == 3:
                st = detect_3d_images(model, frame[0], frame[1], frame[2], eval_thresh, class_index, dense_align, slot, wait=False,
                                      solver=solver, async_host=True)
            else:
                l, r, info, calib, im_shape = frame[:5]
                scale = float(np.float32(frame[5])) if len(frame) > 5 else _scale32(info)
                lazy = _lazy(model)
                out = model(l, r, info, slot=slot, kpts=not lazy, alias_outputs=True)
                st = launch_3d(out, l, r, info, scale, calib, im_shape, eval_thresh, class_index, dense_align, slot, solver,
                               lazy=_plan_of(model, l, slot) if lazy else None, async_host=True)
        for older, _ in inflight:                              # solver='host': a host phase of every pair in flight whose device
            step_3d(older, block=False)                        # work has finished -- never waiting for one that has not
        inflight.append((st, frame))
    while inflight:
        yield finish(inflight.popleft())


class _Pair(object):
    """One stereo pair on its way through the staged scipy pipeline."""
    __slots__ = ('frame', 'stream', 'stage', 'rec_host', 'event', 'cand', 'alphas', 'pending', 'solved', 'dets', 'succ_host',
                 'dis_host', 'todo', 'objs')


def _detect_3d_stream_scipy(model, frames, pool, eval_thresh, class_index, dense_align, slots):
    """The scipy arrangement, staged: forward + decode + class NMS + record (GPU, async) -> borders + 4-DoF tasks (host ->
    pool, async) -> dense alignment (GPU, async) -> 3-DoF tasks (pool, async) -> results; every loop iteration launches the
    next pair's forward and moves each pair in flight one stage on."""
    from . import distributed as sdist
    streams = _slot_streams(max(1, slots))
    inflight = collections.deque()

    def launch(k, frame):
        p = _Pair()
        p.frame, p.stream, p.stage, p.objs = frame, streams[k % len(streams)], 1, None
        l, r, info = frame[0], frame[1], frame[2]
        p.stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(p.stream):
            out = model(l, r, info, slot=k % len(streams))
            det = postprocess.decode_detections(*out[:8], info)
            keep_idx, num = postprocess.class_nms_device(det, class_index, eval_thresh, cfg.TEST.NMS)
            rec = sdist.pack_records_device(det, keep_idx, num, class_index)
            p.rec_host = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
            p.rec_host.copy_(rec, non_blocking=True)
            p.event = torch.cuda.Event()
            p.event.record(p.stream)
        return p

    def advance(p):
        calib, im_shape = p.frame[3], p.frame[4]
        if p.stage == 1:                                       # detections are on the host -> borders, 4-DoF tasks
            p.event.synchronize()
            rec = p.rec_host.numpy()
            k = int(rec[0, 0])
            body = rec[1:k + 1]
            dl = np.concatenate((body[:, 1:5], body[:, 0:1]), 1)
            dr = np.concatenate((body[:, 5:9], body[:, 0:1]), 1)
            do, kpts = body[:, 9:14].copy(), body[:, 14:19].copy()
            p.dets = (dl, dr, do, kpts)
            if k == 0:
                p.objs, p.stage = [], 9
                return
            inferred = kitti_utils.infer_boundary(im_shape, dl)
            for i in range(k):
                if kpts[i, 4] - kpts[i, 3] < 0.5 * (inferred[i, 1] - inferred[i, 0]):
                    kpts[i, 3:5] = inferred[i]
            p.cand = [i for i in range(k) if dl[i, -1] > eval_thresh]
            p.alphas = [m.atan2(do[i, 3], do[i, 4]) for i in p.cand]
            p.pending = pool.submit([(4, tuple(im_shape), calib.p2, calib.p3, (a, do[i, 0:3], dl[i, 0:4], dr[i, 0:4], kpts[i]))
                                     for i, a in zip(p.cand, p.alphas)])
            p.stage = 2
        elif p.stage == 2:                                     # 4-DoF results -> dense alignment on the GPU
            dl, dr, do, kpts = p.dets
            p.solved = []
            for i, alpha, (status, state) in zip(p.cand, p.alphas, p.pending.get()):
                if status > 0:
                    p.solved.append({'box_left': dl[i, 0:4].copy(), 'box_right': dr[i, 0:4].copy(), 'score': float(dl[i, 4]),
                                     'dim': do[i, 0:3].astype(np.float64), 'alpha': alpha,
                                     'xyz': np.array(state[0:3], dtype=np.float64), 'theta': float(state[3]),
                                     'kpts': kpts[i].copy(), 'aligned': False, 'xyz_init': np.array(state[0:3], dtype=np.float64),
                                     'theta_init': float(state[3])})
            if not p.solved or not dense_align:
                p.objs, p.stage = p.solved, 9
                return
            l, r, info = p.frame[0], p.frame[1], p.frame[2]
            dev = l.device
            f32 = lambda rows: torch.tensor(np.asarray(rows), dtype=torch.float32).to(dev, non_blocking=True)
            with torch.no_grad(), torch.cuda.stream(p.stream):
                succ, dis = align_parallel(calib, float(np.float32(p.frame[5])) if len(p.frame) > 5 else _scale32(info), l, r,
                                           f32([o['box_left'] for o in p.solved]), f32([o['kpts'] for o in p.solved]),
                                           f32([[o['xyz'][0], o['xyz'][1], o['xyz'][2], o['dim'][0], o['dim'][1], o['dim'][2],
                                                 o['theta']] for o in p.solved]))
                p.succ_host = torch.empty(succ.shape, dtype=succ.dtype, pin_memory=True)
                p.dis_host = torch.empty(dis.shape, dtype=dis.dtype, pin_memory=True)
                p.succ_host.copy_(succ, non_blocking=True)
                p.dis_host.copy_(dis, non_blocking=True)
                p.event = torch.cuda.Event()
                p.event.record(p.stream)
            p.stage = 3
        elif p.stage == 3:                                     # aligned disparities -> 3-DoF tasks
            p.event.synchronize()
            succ, dis = check_status(p.succ_host.numpy()), p.dis_host.numpy()
            p.todo = [i for i in range(len(p.solved)) if succ[i] > 0]
            p.pending = pool.submit([(3, tuple(im_shape), calib.p2, calib.p3,
                                      (_alpha32(p.solved[i]['alpha']), p.solved[i]['dim'], _f64(p.solved[i]['box_left']),
                                       float(dis[i]), _f64(p.solved[i]['kpts']))) for i in p.todo])
            p.stage = 4
        elif p.stage == 4:                                     # rectified poses
            dis = p.dis_host.numpy()
            for i, (state, z) in zip(p.todo, p.pending.get()):
                o = p.solved[i]
                o['xyz'] = np.array([state[0], state[1], z], dtype=np.float64)
                o['theta'] = float(state[2])
                o['aligned'] = True
                o['disparity'] = float(dis[i])
            p.objs, p.stage = p.solved, 9

    def drain_ready():
        while inflight and inflight[0].stage == 9:
            yield inflight.popleft().objs

    for k, frame in enumerate(frames):
        new = launch(k, frame)
        for q in list(inflight):
            if q.stage != 9:
                advance(q)
        inflight.append(new)
        for objs in drain_ready():
            yield objs
    while inflight:
        for q in list(inflight):
            if q.stage != 9:
                advance(q)
        for objs in drain_ready():
            yield objs


def write_kitti_results(result_dir, file_number, calib, objects):
    """test_net.py:329-330: one KITTI line per object."""
    for o in objects:
        kitti_utils.write_detection_results(result_dir, file_number, calib, o['box_left'], o['xyz'], o['dim'],
                                            o['theta'], o['score'])
