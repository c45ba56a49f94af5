"""KITTI split evaluation driver - the reference's test_net.py:62-345 (config 4 of BASELINE.json), MI355X layout:
one process per GPU, image ids sharded `i mod world` (the reference loops over them on one GPU with batch size 1),
every frame: PNG decode on the host -> uint8 images to the device -> preprocessing, forward, decode, NMS, borders, 3-D solve,
dense alignment, rectification, all on the device (`pipeline.detect_3d_stream`) -> one KITTI result file per frame (`kitti_utils.write_detection_results`, test_net.py:329-330).
Result files are per frame, so ranks never write to the same file; there is no collective besides the final barrier.

    python -m stereo_rcnn_amd.test_net --kitti-root <.../object/training> --split val.txt --checkpoint model.pth --result-dir out
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m stereo_rcnn_amd.test_net ...

`--kitti-root` holds image_2/, image_3/ and calib/ (the layout lib/datasets/kitti.py:60-75 reads).  The external KITTI
evaluator (C++ `evaluate_object_3d_offline`) consumes <result-dir>/data/*.txt as with the reference.
"""
import argparse
import os
import time

import numpy as np
import torch

from . import engine, pipeline
from .distributed import shard_indices
from .model.stereo_rcnn.resnet import resnet
from .model.utils import kitti_utils
from .model.utils.config import cfg


# PNG decode of the KITTI loop in worker PROCESSES (default; SRCNN_DECODE_PROCESSES=0 = decoder threads inside this process, round 5's form)
DECODE_PROCESSES = os.environ.get('SRCNN_DECODE_PROCESSES', '1') != '0'


# result files + records on a writer thread (SRCNN_WRITER_THREAD=0: written by the loop thread between two launches)
WRITER_THREAD = os.environ.get('SRCNN_WRITER_THREAD', '1') != '0'


class _DecodeWorkers(object):
    """`n` png_worker.py processes and the shared file they decode into (one slot of two images per worker).  decode() hands a
    request to an idle worker and blocks the calling THREAD on its answer (a pipe read: no GIL held), then returns views of the
    worker's slot and the worker itself -- the caller copies the pixels out (into the page-locked ring) and release()s it."""

    def __init__(self, n, nbytes=3 * 512 * 1408):
        import mmap
        import queue
        import subprocess
        import sys
        import tempfile
        self.n, self.nbytes = n, nbytes
        base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
        fd, self.path = tempfile.mkstemp(prefix='srcnn_png_', dir=base)
        os.ftruncate(fd, n * 2 * nbytes)
        self.mm = mmap.mmap(fd, 0)
        os.close(fd)
        self.view = np.frombuffer(self.mm, dtype=np.uint8)
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'png_worker.py')
        self.procs, self.idle = [], queue.Queue()
        try:
            for i in range(n):
                self.procs.append(subprocess.Popen([sys.executable, script, self.path, str(i), str(nbytes)], stdin=subprocess.PIPE,
                                                   stdout=subprocess.PIPE, universal_newlines=True, bufsize=1))
            for i, pr in enumerate(self.procs):
                if pr.stdout.readline().strip() != 'ready':
                    raise RuntimeError('PNG decode worker %d did not start (exit code %s)' % (i, pr.poll()))
                self.idle.put(i)
        except Exception:
            self.close()
            raise
        os.unlink(self.path)              # every worker has it mapped: the file goes away with the last mapping
        self.path = None

    def decode(self, paths):
        import json
        i = self.idle.get()
        pr = self.procs[i]
        try:
            pr.stdin.write(json.dumps({'paths': list(paths)}) + '\n')
            pr.stdin.flush()
            line = pr.stdout.readline()
            if not line:
                raise RuntimeError('PNG decode worker %d died (exit code %s)' % (i, pr.poll()))
            reply = json.loads(line)
            if 'error' in reply:
                raise RuntimeError('PNG decode worker: %s' % reply['error'])
        except Exception:
            self.idle.put(i)
            raise
        base, out = i * 2 * self.nbytes, []
        for k, shape in enumerate(reply['shapes']):
            nb = int(np.prod(shape))
            out.append(self.view[base + k * self.nbytes: base + k * self.nbytes + nb].reshape(shape))
        return i, out

    def release(self, i):
        self.idle.put(i)

    def close(self):
        for pr in self.procs:
            try:
                pr.stdin.close()
            except Exception:
                pass
        for pr in self.procs:
            try:
                pr.wait(timeout=5)
            except Exception:
                pr.kill()
        self.procs = []
        if self.path:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.path = None


_decode_workers = {}


def decode_workers(n):
    """The process-wide set of `n` decode workers (started on first use, reused by later splits, stopped at exit)."""
    w = _decode_workers.get(n)
    if w is None:
        import atexit
        w = _decode_workers[n] = _DecodeWorkers(n)
        atexit.register(w.close)
    return w


def read_split(path):
    with open(path) as fh:
        return [ln.strip() for ln in fh if ln.strip()]


def read_png_rgb(path):
    """uint8 (H, W, 3) RGB.  (The reference reads BGR with cv2 and flips; the device preprocessing takes RGB.)"""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'), dtype=np.uint8)


# A/B switch: SRCNN_ZERO_COPY_IMAGES=0 = round 5's path (decoded images copied to the device with hipMemcpyAsync from a pinned pool)
ZERO_COPY_IMAGES = os.environ.get('SRCNN_ZERO_COPY_IMAGES', '1') != '0'


class _PinnedPool(object):
    """Page-locked staging buffers for the decoded uint8 images.  A host-to-device copy from PAGEABLE memory is staged by the
    runtime and blocks the issuing thread until the device has taken it (measured: 5.8 ms of the loop thread per pair in
    bench.py --config 3 -- the whole step time); from pinned memory it is an asynchronous DMA.  Buffers are recycled once the copy
    that read them has completed (an event per use)."""

    def __init__(self):
        self.free, self.busy = {}, []

    def stage(self, arr, device):
        import numpy as np
        key = tuple(arr.shape)
        for i in range(len(self.busy) - 1, -1, -1):
            ev, k, buf = self.busy[i]
            if ev.query():
                self.free.setdefault(k, []).append(buf)
                self.busy.pop(i)
        lst = self.free.get(key)
        buf = lst.pop() if lst else torch.empty(key, dtype=torch.uint8, pin_memory=True)
        buf.numpy()[...] = arr if isinstance(arr, np.ndarray) else np.asarray(arr)
        dev_t = buf.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.busy.append((ev, key, buf))
        return dev_t


class _PinnedRing(object):
    """Page-locked image buffers the DEVICE reads where they are (round 6): the decode thread of frame j copies its two decoded
    images into ring entry j % n, and the fused preprocessing kernel reads them over the bus -- no hipMemcpy in any stream.  An
    asynchronous H2D copy per image looked free (0.35 ms of the loop thread per pair) and cost the streamed flow 1.0 ms per pair
    (profiles/flow3d_input_path_r06.txt: 5.9 ms with device-resident images or zero-copy, 6.9 with two H2D copies per frame, 8.5
    with the copies on a stream of their own) -- the copies, not their bytes.  Entry j % n is rewritten when frame j + n is
    decoded, `prefetch` frames before it is consumed; frame j's pair has been collected by then if n >= prefetch + slots + 1."""

    def __init__(self, n, nbytes=3 * 512 * 1408):
        self.n, self.nbytes = n, nbytes
        self.bufs = [None] * n

    def put(self, j, left, right):
        e = self.bufs[j % self.n]
        if e is None:
            e = self.bufs[j % self.n] = (torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True),
                                         torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True))
        out = []
        for buf, arr in zip(e, (left, right)):
            arr = np.ascontiguousarray(arr, dtype=np.uint8)
            assert arr.nbytes <= self.nbytes, "image larger than the staging ring's entries"
            view = buf[:arr.nbytes].view(*arr.shape)
            np.copyto(view.numpy(), arr)
            out.append(view)
        return out


def run_split(model, kitti_root, ids, result_dir, device, pool=None, read_image=read_png_rgb, log=None, prefetch=4,
              solver='host', slots=4, detect_stream=None, records=None, timers=None):
    """Processes `ids` (already this rank's shard).  Returns (frames, objects written, seconds).
    `detect_stream(frames)`: replaces the detector (a generator of object lists, one per frame) -- the CPU tests drive the
    sharding / writer / gather logic with it; `records`: a list that receives one detection record per frame
    (distributed.objects_to_record) for the gather of the split.
    Frames go through pipeline.detect_3d_stream: PNG decoding runs `prefetch` frames ahead on host threads, the decoded
    uint8 images are copied to the device and everything else -- preprocessing, forward, decode, NMS, borders, 4-DoF solve,
    dense alignment, 3-DoF rectification -- runs on the record flow with `slots` pairs in flight: solver='host' (default; the
    Newton-CG solves in C on the host between the device stages, bit-identical to the scipy path) or 'device' (solves as
    kernels).  solver='scipy' (with a SolverPool) runs the reference's host arrangement instead: the comparison path.
    `timers`: a dict that receives the host-side seconds of the loop (bench.py --config 3): 'decode_s' (PNG decode + calibration
    parse, summed over the prefetch threads), 'h2d_s' (issuing the uint8 copies), 'write_s' (KITTI result files + record),
    'loop_s' (wall time of the whole loop); pipeline.TIMERS holds the solver and GPU-wait shares of the same loop."""
    import collections
    import concurrent.futures as cf
    t0, n_obj = time.time(), 0
    os.makedirs(os.path.join(result_dir, 'data'), exist_ok=True)
    if timers is not None:
        for k in ('decode_s', 'h2d_s', 'write_s', 'loop_s'):
            timers.setdefault(k, 0.0)

    zero_copy = detect_stream is None and solver in ('device', 'host') and ZERO_COPY_IMAGES
    ring = _PinnedRing(max(1, prefetch) + max(1, slots) + 4) if zero_copy else None
    # decode in worker processes: the default reader on the zero-copy path (a custom `read_image` runs on the prefetch threads)
    workers = None
    if ring is not None and read_image is read_png_rgb and DECODE_PROCESSES:
        try:
            workers = decode_workers(max(1, prefetch))
        except (OSError, RuntimeError) as e:       # no room for the shared file, no process table entries, ...: decode on the threads
            import warnings
            warnings.warn('PNG decode workers unavailable (%s): decoding on %d threads of this process' % (e, max(1, prefetch)))

    def load(j, frame):
        td = time.perf_counter()
        paths = (os.path.join(kitti_root, 'image_2', frame + '.png'), os.path.join(kitti_root, 'image_3', frame + '.png'))
        if workers is not None:
            w, (left, right) = workers.decode(paths)            # this thread sleeps on the worker's pipe meanwhile
            try:
                left, right = ring.put(j, left, right)          # out of the worker's slot into the page-locked ring
            finally:
                workers.release(w)
        else:
            left, right = read_image(paths[0]), read_image(paths[1])
        calib = kitti_utils.read_obj_calibration(os.path.join(kitti_root, 'calib', frame + '.txt'))
        if ring is not None and workers is None:
            left, right = ring.put(j, left, right)              # page-locked views the preprocessing kernel reads directly
        if timers is not None:
            timers['decode_s'] += time.perf_counter() - td        # (float += under the GIL: good enough for a per-pair average)
        return (left, right, calib)

    calibs = collections.deque()
    pinned = _PinnedPool()

    def frames():
        with cf.ThreadPoolExecutor(max_workers=max(1, prefetch)) as ex:
            pending = collections.deque()
            it = iter(ids)
            for j, frame in enumerate(it):
                pending.append(ex.submit(load, j, frame))
                if len(pending) <= prefetch:
                    continue
                yield to_device(pending.popleft().result())
            while pending:
                yield to_device(pending.popleft().result())

    def to_device(loaded):
        left, right, calib = loaded
        calibs.append(calib)
        if detect_stream is not None:
            return (left, right, calib)                         # injected detector: frames stay on the host
        if ring is not None:
            return (left, right, calib)                         # page-locked host images: read by the fused preprocessing where they are
        th = time.perf_counter()
        lu, ru = pinned.stage(left, device), pinned.stage(right, device)
        if timers is not None:
            timers['h2d_s'] += time.perf_counter() - th
        if solver in ('device', 'host'):
            return (lu, ru, calib)                              # preprocessing is fused in front of the forward
        l, scale = engine.preprocess(lu, cfg.TEST.SCALES[0])
        r, _ = engine.preprocess(ru, cfg.TEST.SCALES[0])
        info = torch.tensor([[l.shape[2], l.shape[3], scale]], dtype=torch.float32).to(device)
        return (l, r, info, calib, left.shape, float(scale))

    def results():
        if detect_stream is not None:
            for objs in detect_stream(frames()):
                yield objs
        elif solver in ('device', 'host') or pool is not None:
            for objs in pipeline.detect_3d_stream(model, frames(), pool, solver=solver, slots=slots):
                yield objs
        else:
            for f in frames():
                yield pipeline.detect_3d(model, *f[:5], solver='scipy')

    def write(frame, calib, objs):
        tw = time.perf_counter()
        open(os.path.join(result_dir, 'data', frame + '.txt'), 'w').close()      # a frame without detections still gets a file
        pipeline.write_kitti_results(result_dir, frame, calib, [o for o in objs if o['aligned']])   # test_net.py:322-330
        if records is not None:
            from .distributed import objects_to_record
            records.append(objects_to_record(objs))
        if timers is not None:
            timers['write_s'] += time.perf_counter() - tw

    # result files and records on ONE writer thread, in frame order: the loop thread goes straight back to launching the next pair
    # (with the PNG decoders in processes of their own the writer is the only other user of this interpreter)
    writer = cf.ThreadPoolExecutor(max_workers=1) if WRITER_THREAD else None
    written = collections.deque()
    try:
        for k, (frame, objs) in enumerate(zip(ids, results())):
            calib = calibs.popleft()
            n_obj += sum(o['aligned'] for o in objs)
            if writer is None:
                write(frame, calib, objs)
            else:
                written.append(writer.submit(write, frame, calib, objs))
                while len(written) > 64:
                    written.popleft().result()                                   # bounded backlog; raises what the writer raised
            if log and (k + 1) % 50 == 0:
                log('%d/%d frames, %.1f frames/s' % (k + 1, len(ids), (k + 1) / (time.time() - t0)))
        while written:
            written.popleft().result()
    finally:
        if writer is not None:
            writer.shutdown(wait=True)
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)
    if timers is not None:
        timers['loop_s'] += time.time() - t0
    return len(ids), n_obj, time.time() - t0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--kitti-root', required=True)
    ap.add_argument('--split', required=True, help='text file with one frame id per line (e.g. data/kitti/splits/val.txt)')
    ap.add_argument('--checkpoint', required=True, help="torch checkpoint with a 'model' state_dict (reference schema) or a bare state_dict")
    ap.add_argument('--result-dir', required=True)
    ap.add_argument('--solver', choices=['host', 'device', 'scipy'], default='host',
                    help="3-D stage: 'host' = native Newton-CG on the host between the device stages, bit-identical to scipy "
                         "(default); 'device' = Newton-CG kernels; 'scipy' = the reference's host arrangement")
    ap.add_argument('--solver-workers', type=int, default=0, help='scipy path: worker processes (0 = cpu_count / ranks, <= 32)')
    ap.add_argument('--precision', choices=['f16x3', 'f32'], default='f16x3')
    ap.add_argument('--gather', action='store_true',
                    help='after the split, all_gather the per-frame detection records (2-D + 3-D fields) over RCCL and let rank 0 '
                         'write <result-dir>/records.pt (the per-frame KITTI files are written by the owning rank either way)')
    args = ap.parse_args(argv)
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    use_dist = 'RANK' in os.environ and world > 1
    from . import serving
    serving.before_hip()                 # before HIP starts: one hardware queue per pair in flight (serving.py, streams.py);
                                         # run_split -> pipeline.detect_3d_stream enters the serving regime (plans, branch placement)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.distributed.init_process_group('nccl')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    model = resnet(('__background__', 'Car'), 101, pretrained=False)
    model.create_architecture()
    sd = torch.load(args.checkpoint, map_location='cpu')
    model.load_state_dict(sd['model'] if 'model' in sd else sd)
    model.cuda()
    model.eval()
    model.precision = args.precision
    ids = read_split(args.split)
    mine = [ids[i] for i in shard_indices(len(ids), rank, world)]
    log = lambda s: print('[rank %d] %s' % (rank, s), flush=True)
    records = [] if args.gather else None
    if args.solver == 'scipy':
        with pipeline.SolverPool(args.solver_workers or None) as pool:
            frames, objs, dt = run_split(model, args.kitti_root, mine, args.result_dir, device, pool, log=log, solver='scipy',
                                         records=records)
    else:
        frames, objs, dt = run_split(model, args.kitti_root, mine, args.result_dir, device, log=log, records=records,
                                     solver=args.solver)
    print('[rank %d] %d frames, %d objects, %.1f s (%.1f frames/s)' % (rank, frames, objs, dt, frames / max(dt, 1e-9)), flush=True)
    if args.gather:
        from .distributed import gather_split_records
        full = gather_split_records([r.to(device) for r in records], len(ids), rank, world)
        if rank == 0:
            torch.save({'ids': ids, 'records': full.cpu()}, os.path.join(args.result_dir, 'records.pt'))
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
