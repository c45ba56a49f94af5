"""Stereo 3-D detection of one pair, end to end - the flow of the reference's demo.py:137-326
(= test_net.py:131-331): network forward, decode, per-class NMS, border inference, 4-DoF box solve,
dense alignment, 3-DoF rectification.  Device work goes through the HIP library; the two scipy
solvers and `infer_boundary` are host code, as in the reference (see model/utils/box_estimator.py
for why they cannot be anything else and still return the reference's boxes)."""
import math as m

import numpy as np
import torch

from . import postprocess
from .model.dense_align.dense_align import align_parallel
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
    succ, dis_final = succ.cpu().numpy(), dis_final.cpu().numpy()
    todo = [i for i in range(len(solved)) if succ[i] > 0]                                      # demo.py:311-319
    res3 = run([(3, tuple(im_shape), calib.p2, calib.p3,
                 (solved[i]['alpha'], solved[i]['dim'], solved[i]['box_left'], float(dis_final[i]), solved[i]['kpts']))
                for i in todo])
    for i, (state, z) in zip(todo, res3):
        o = solved[i]
        o['xyz'] = np.array([state[0], state[1], z], dtype=np.float64)
        o['theta'] = float(state[2])
        o['aligned'] = True
        o['disparity'] = float(dis_final[i])
    return solved


def write_kitti_results(result_dir, file_number, calib, objects):
    """test_net.py:329-330: one KITTI line per object."""
    for o in objects:
        kitti_utils.write_detection_results(result_dir, file_number, calib, o['box_left'], o['xyz'], o['dim'],
                                            o['theta'], o['score'])
