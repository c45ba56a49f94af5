"""Stereo 3-D detection of one pair, end to end - the flow of the reference's demo.py:137-326
(= test_net.py:131-331): network forward, decode, per-class NMS, border inference, 4-DoF box solve,
dense alignment, 3-DoF rectification.  Device work goes through the HIP library; the two scipy
solvers and `infer_boundary` are host code, as in the reference (see model/utils/box_estimator.py
for why they cannot be anything else and still return the reference's boxes)."""
import math as m

import numpy as np
import torch

from . import postprocess
from .model.dense_align.dense_align import align_parallel, check_status
from .model.utils import box_estimator, kitti_utils
from .model.utils.config import cfg


class _PlainCalib(object):
    """Picklable stand-in for FrameCalibrationData: the solvers only read p2 and p3."""

    def __init__(self, p2, p3):
        self.p2, self.p3 = np.asarray(p2, dtype=np.float64), np.asarray(p3, dtype=np.float64)


def _solve_task(task):
    """One solver call in a worker process (plain numpy in, plain numpy out)."""
    kind, im_shape, p2, p3, args = task
    calib = _PlainCalib(p2, p3)
    if kind == 4:
        status, state = box_estimator.solve_x_y_z_theta_from_kpt(im_shape, calib, *args)
        return status, (np.asarray(state, dtype=np.float64) if status or np.ndim(state) else None)
    state, z = box_estimator.solve_x_y_theta_from_kpt(im_shape, calib, *args)
    return np.asarray(state, dtype=np.float64), float(z)


def _alpha32(alpha):
    """The 3-DoF rectification reads alpha back from the float32 `poses_all` tensor (demo.py:313 -> poses[7]), the 4-DoF
    solve uses the float64 atan2 result directly (demo.py:288-293): mirror both."""
    return float(np.float32(alpha))


def _noop(_):
    return None


def _worker_init():
    import os
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    torch.set_num_threads(1)


class SolverPool(object):
    """Process pool for the host-side scipy solvers (A14 / A17).  Each solve is ~2 ms of single-threaded Python
    (scipy's Newton-CG driver dominates, not the cost function), an image has tens of objects and the hosts of MI355X
    boxes have hundreds of cores: fan the objects of a pair out.  Results are identical to the serial path (same
    function, same inputs, deterministic optimiser).  Workers are spawned (not forked: the parent owns a HIP context)
    and never touch the GPU."""

    def __init__(self, workers=8):
        import multiprocessing as mp
        import os
        # the workers must come up single-threaded: BLAS / OpenMP pools sized for a 256-core host inside every worker
        # oversubscribe the machine (measured: 16 workers 6x SLOWER than 8 without this).  The libraries read these
        # variables when they load, i.e. in the child, which inherits the environment at spawn time.
        keys = ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS')
        saved = {k: os.environ.get(k) for k in keys}
        os.environ.update({k: '1' for k in keys})
        try:
            self._pool = mp.get_context('spawn').Pool(int(workers), initializer=_worker_init)
            self._pool.map(_noop, range(int(workers)))          # workers are up (and have imported) before the first pair
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


def detect_3d(model, im_left_data, im_right_data, im_info, calib, im_shape, eval_thresh=0.05, class_index=1,
              dense_align=True, pool=None):
    """Returns a list of dicts (one per solved object, descending score):
    box_left (4), box_right (4), score, dim (w,h,l), alpha, xyz (3), theta, aligned (bool).
    `pool`: optional SolverPool; the two solver stages then run in parallel over the objects of the pair."""
    with torch.no_grad():
        out = model(im_left_data, im_right_data, im_info)
        det = postprocess.decode_detections(*out[:8], im_info)
        cls = postprocess.class_detections(det, class_index, eval_thresh, cfg.TEST.NMS)
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
    res4 = run([(4, tuple(im_shape), calib.p2, calib.p3,
                 (a, dim_orien[i, 0:3], dets_left[i, 0:4], dets_right[i, 0:4], kpts[i])) for i, a in zip(cand, alphas)])
    solved = []
    for i, alpha, (status, state) in zip(cand, alphas, res4):                                    # demo.py:291-302
        if status > 0:
            solved.append({'box_left': dets_left[i, 0:4].copy(), 'box_right': dets_right[i, 0:4].copy(),
                           'score': float(dets_left[i, 4]), 'dim': dim_orien[i, 0:3].astype(np.float64), 'alpha': alpha,
                           'xyz': np.array(state[0:3], dtype=np.float64), 'theta': float(state[3]),
                           'kpts': kpts[i].copy(), 'aligned': False,
                           'xyz_init': np.array(state[0:3], dtype=np.float64)})   # 4-DoF solve, before alignment
    if not solved or not dense_align:
        return solved
    dev = im_left_data.device
    f32 = lambda rows: torch.tensor(np.asarray(rows), dtype=torch.float32, device=dev)
    boxes = f32([o['box_left'] for o in solved])
    kp = f32([o['kpts'] for o in solved])
    poses = f32([[o['xyz'][0], o['xyz'][1], o['xyz'][2], o['dim'][0], o['dim'][1], o['dim'][2], o['theta']]
                 for o in solved])
    succ, dis_final = align_parallel(calib, float(im_info.view(-1, 3)[0, 2]), im_left_data, im_right_data, boxes, kp,
                                     poses)                   # demo.py:306-308
    succ, dis_final = check_status(succ.cpu().numpy()), dis_final.cpu().numpy()
    todo = [i for i in range(len(solved)) if succ[i] > 0]                                      # demo.py:311-319
    res3 = run([(3, tuple(im_shape), calib.p2, calib.p3,
                 (_alpha32(solved[i]['alpha']), solved[i]['dim'], solved[i]['box_left'], float(dis_final[i]), solved[i]['kpts']))
                for i in todo])
    for i, (state, z) in zip(todo, res3):
        o = solved[i]
        o['xyz'] = np.array([state[0], state[1], z], dtype=np.float64)
        o['theta'] = float(state[2])
        o['aligned'] = True
        o['disparity'] = float(dis_final[i])
    return solved


class _Pair(object):
    """One stereo pair on its way through the streaming pipeline."""
    __slots__ = ('frame', 'stream', 'stage', 'rec_host', 'event', 'cand', 'alphas', 'pending', 'solved', 'dets', 'succ_host',
                 'dis_host', 'todo', 'objs')


def detect_3d_stream(model, frames, pool, eval_thresh=0.05, class_index=1, dense_align=True, slots=2):
    """Generator form of detect_3d for a sequence of pairs: yields one object list per frame, in order, and keeps the GPU,
    the host thread and the solver pool busy at the same time.  frames: iterable of (im_left_data, im_right_data, im_info,
    calib, im_shape) with device tensors.  Per pair the stages are
        forward + decode + class NMS + record packing (GPU, async)  ->  borders + 4-DoF tasks (host -> pool, async)
        ->  dense alignment (GPU, async)  ->  3-DoF tasks (pool, async)  ->  results;
    every loop iteration launches the next pair's forward and moves each pair in flight ONE stage on, so a wait is always
    on work that was started an iteration earlier.  Same per-pair results as detect_3d (same kernels, same solver calls)."""
    import collections
    from . import distributed as sdist
    streams = [torch.cuda.Stream() for _ in range(max(1, slots))]
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
                                     'kpts': kpts[i].copy(), 'aligned': False, 'xyz_init': np.array(state[0:3], dtype=np.float64)})
            if not p.solved or not dense_align:
                p.objs, p.stage = p.solved, 9
                return
            l, r, info = p.frame[0], p.frame[1], p.frame[2]
            dev = l.device
            f32 = lambda rows: torch.tensor(np.asarray(rows), dtype=torch.float32).to(dev, non_blocking=True)
            with torch.no_grad(), torch.cuda.stream(p.stream):
                succ, dis = align_parallel(calib, p.frame[5] if len(p.frame) > 5 else float(info.view(-1, 3)[0, 2]), l, r,
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
                                      (_alpha32(p.solved[i]['alpha']), p.solved[i]['dim'], p.solved[i]['box_left'], float(dis[i]),
                                       p.solved[i]['kpts'])) for i in p.todo])
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
