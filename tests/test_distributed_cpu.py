"""CPU: the N>1 path (pair sharding + detection gather) with the gloo backend, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stereo_rcnn_amd import distributed as sdist


def test_shard_indices_partition():
    for world in (1, 2, 3, 8):
        parts = [sdist.shard_indices(3769, r, world) for r in range(world)]     # KITTI val list size
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(3769))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_pack_unpack_roundtrip():
    k = 7
    det = {'dets_left': torch.rand(k, 5), 'dets_right': torch.rand(k, 5), 'dim_orien': torch.rand(k, 5),
           'kpts': torch.rand(k, 5), 'keep_idx': torch.arange(k, dtype=torch.int32) * 3}
    rec = sdist.pack_records(det)
    assert rec.shape == (301, sdist.REC_COLS)
    u = sdist.unpack_records(rec)
    assert torch.equal(u['boxes_left'], det['dets_left'][:, :4]) and torch.equal(u['scores'], det['dets_left'][:, 4])
    assert u['roi_index'].tolist() == [0, 3, 6, 9, 12, 15, 18]
    # device-style packing (no host sync) gives the same record
    full = {'scores': torch.zeros(300, 2), 'boxes_left': torch.zeros(300, 8), 'boxes_right': torch.zeros(300, 8),
            'dim_orien': torch.zeros(300, 10), 'kpts': torch.zeros(300, 5)}
    idx = det['keep_idx'].long()
    full['scores'][idx, 1] = det['dets_left'][:, 4]
    full['boxes_left'][idx, 4:8] = det['dets_left'][:, :4]
    full['boxes_right'][idx, 4:8] = det['dets_right'][:, :4]
    full['dim_orien'][idx, 5:10] = det['dim_orien']
    full['kpts'][idx] = det['kpts']
    keep = torch.full((300,), -1, dtype=torch.int32)
    keep[:k] = det['keep_idx']
    rec2 = sdist.pack_records_device(full, keep, torch.tensor([k], dtype=torch.int32), 1)
    assert torch.allclose(rec2, rec)


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = sdist.shard_indices(10, rank, world)
    rec = torch.zeros(301, sdist.REC_COLS)
    rec[0, 0] = len(mine)
    rec[1:1 + len(mine), 0] = torch.tensor(mine, dtype=torch.float32)
    out, work = sdist.gather_detections(rec)
    if work is not None:
        work.wait()
    got = []
    for r in range(world):
        k = int(out[r, 0, 0])
        got += out[r, 1:1 + k, 0].long().tolist()
    # bench.py's form: the records of G steps packed in place (out=) and gathered by ONE all_gather
    G, n = 3, 300
    buf = torch.zeros(G, n + 1, sdist.REC_COLS)
    for row in range(G):
        det = {'scores': torch.zeros(n, 2), 'boxes_left': torch.zeros(n, 8), 'boxes_right': torch.zeros(n, 8),
               'dim_orien': torch.zeros(n, 10), 'kpts': torch.zeros(n, 5)}
        det['scores'][:, 1] = 100 * rank + 10 * row + torch.arange(n) / 1000.0
        keep = torch.full((n,), -1, dtype=torch.int32)
        keep[:row + 1] = torch.arange(row + 1, dtype=torch.int32)
        ret = sdist.pack_records_device(det, keep, torch.tensor([row + 1], dtype=torch.int32), 1, out=buf[row])
        assert ret.data_ptr() == buf[row].data_ptr()
    stacked, work = sdist.gather_detections(buf)
    if work is not None:
        work.wait()
    assert tuple(stacked.shape) == (world, G, n + 1, sdist.REC_COLS)
    for r in range(world):
        for row in range(G):
            assert int(stacked[r, row, 0, 0]) == row + 1
            assert abs(float(stacked[r, row, 1, 0]) - (100 * r + 10 * row)) < 1e-6
            assert float(stacked[r, row, row + 2:, :].abs().sum()) == 0.0
    q.put((rank, sorted(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, got in res:
        assert got == list(range(10))      # every rank sees every pair's record exactly once


def test_bench_self_launches_n_ranks_dry_run():
    """`python bench.py --gpus 2` with no RANK in the environment re-executes itself under torch.distributed.run with two ranks
    (the driver's SCALE command must measure N ranks).  --dry-run: no GPU, gloo, fake records -- the launcher,
    the env contract, the batched all_gather and the one-JSON-line output are the real ones."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1',
                          '--gather-every', '2', '--dry-run'], cwd=root, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['dry_run'] is True and d['scaling'] == 'weak'
    # the host side of the full-3-D-flow leg ran on BOTH ranks: each rank takes its share of the host
    # cores for its Newton-CG threads (cpu_count / LOCAL_WORLD_SIZE, <= 16), pins itself to a disjoint half of the CPU mask,
    # and solves its record in the library's host build
    legs = d['config']['full_3d_flow_host_side']
    assert [g['rank'] for g in legs] == [0, 1]
    ncpu = len(os.sched_getaffinity(0))
    for g in legs:
        assert g['host_solver_threads'] == max(1, min(16, g['cpus_after_pinning'] // 2))      # budget: mask / LOCAL_WORLD_SIZE
        assert 1 <= g['cpus_after_pinning'] <= max(1, (ncpu + 1) // 2) and g['solved_of_24'] == 24
    # a single process asked for one GPU stays a single process
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--dry-run'], cwd=root,
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])['n_gpus'] == 1


def _fake_objects(frame_left):
    """Deterministic fake detections from the frame's pixels (stands in for the GPU detector in the CPU tests)."""
    import numpy as np
    seed = int(frame_left[0, 0, 0]) + 1
    rng = np.random.default_rng(seed)
    objs = []
    for i in range(seed % 4 + 1):
        x1, y1 = rng.uniform(0, 300), rng.uniform(20, 80)
        objs.append({'score': float(rng.uniform(0.1, 1)), 'box_left': np.array([x1, y1, x1 + 40, y1 + 30], np.float32),
                     'box_right': np.array([x1 - 8, y1, x1 + 32, y1 + 30], np.float32), 'dim': np.array([1.6, 1.5, 4.0]),
                     'alpha': 0.3 * i, 'kpts': np.array([x1 + 5, 1, 0.9, x1, x1 + 40], np.float32),
                     'xyz_init': rng.uniform(-5, 30, 3), 'theta_init': 0.1, 'xyz': rng.uniform(-5, 30, 3), 'theta': 0.2 + i,
                     'aligned': i % 2 == 0, 'disparity': float(rng.uniform(5, 60)), 'roi_index': i})
    return objs


def _split_worker(rank, world, port, root, result_dir, q):
    import numpy as np
    from stereo_rcnn_amd import test_net
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ids = test_net.read_split(os.path.join(root, 'val.txt'))
    mine = [ids[i] for i in sdist.shard_indices(len(ids), rank, world)]
    records = []
    detect = lambda frames: (_fake_objects(f[0]) for f in frames)
    n_frames, n_obj, _ = test_net.run_split(None, root, mine, result_dir, None, detect_stream=detect, records=records)
    full = sdist.gather_split_records(records, len(ids), rank, world)
    q.put((rank, mine, n_obj, full.numpy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_kitti_split_driver_world_size_2_gloo(tmp_path):
    """test_net.py's loop on two ranks (gloo, CPU): frames sharded i mod 2, every frame's KITTI file written by its owner, the
    per-frame records -- 2-D detection AND the 3-D fields (x, y, z, theta, disparity, status: SURVEY 8(e)) -- gathered by one
    all_gather so that every rank holds the whole split in frame order."""
    import numpy as np
    from PIL import Image
    root = tmp_path / 'training'
    for d in ('image_2', 'image_3', 'calib'):
        (root / d).mkdir(parents=True)
    ids = ['%06d' % i for i in range(5)]
    p2 = np.array([721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884]).reshape(3, 4)
    p3 = p2.copy(); p3[0, 3] = -339.5242
    row = lambda name, mat: name + ': ' + ' '.join('%.12e' % v for v in np.ravel(mat))
    for k, frame in enumerate(ids):
        img = np.full((40, 60, 3), k, np.uint8)
        Image.fromarray(img).save(str(root / 'image_2' / (frame + '.png')))
        Image.fromarray(img).save(str(root / 'image_3' / (frame + '.png')))
        (root / 'calib' / (frame + '.txt')).write_text('\n'.join([row('P0', p2), row('P1', p2), row('P2', p2), row('P3', p3),
                                                                 row('R0_rect', np.eye(3)), row('Tr_velo_to_cam', np.eye(3, 4))]) + '\n')
    (root / 'val.txt').write_text('\n'.join(ids) + '\n')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, str(root), str(tmp_path / 'res'), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == ['000000', '000002', '000004'] and got[1][1] == ['000001', '000003']
    assert sorted(os.listdir(str(tmp_path / 'res' / 'data'))) == [f + '.txt' for f in ids]
    full = got[0][3]
    assert full.shape == (5, 301, sdist.REC_COLS)
    total_aligned = 0
    for k, frame in enumerate(ids):
        objs = _fake_objects(np.full((40, 60, 3), k, np.uint8))
        want = sdist.objects_to_record(objs).numpy()
        assert np.array_equal(full[k], want)
        u = sdist.unpack_records(torch.from_numpy(full[k]))
        assert np.allclose(u['pose'][:, :3].numpy(), np.array([o['xyz'] for o in objs], np.float32))
        lines = (tmp_path / 'res' / 'data' / (frame + '.txt')).read_text().splitlines()
        assert len(lines) == sum(o['aligned'] for o in objs)
        total_aligned += len(lines)
    assert total_aligned == got[0][2] + got[1][2]


def test_synthetic_kitti_tree_is_what_run_split_reads(tmp_path):
    """fixture.write_kitti_tree: the layout test_net.run_split takes, `n_ids` frames replaying `distinct` PNG pairs through real
    files (symlinks), calibration readable by the product's own parser."""
    import numpy as np
    from stereo_rcnn_amd import fixture, test_net
    from stereo_rcnn_amd.model.utils import kitti_utils
    root = str(tmp_path / 'training')
    ids = fixture.write_kitti_tree(root, n_ids=7, distinct=2, height=24, width=64)
    assert ids == ['%06d' % i for i in range(7)] == test_net.read_split(os.path.join(root, 'val.txt'))
    a, b = test_net.read_png_rgb(os.path.join(root, 'image_2', '000001.png')), test_net.read_png_rgb(os.path.join(root, 'image_2', '000003.png'))
    c = test_net.read_png_rgb(os.path.join(root, 'image_2', '000002.png'))
    assert a.shape == (24, 64, 3) and a.dtype == np.uint8 and np.array_equal(a, b) and not np.array_equal(a, c)
    left, right = fixture.synthetic_pair(3 + 1, 24, 64)
    assert np.array_equal(a, left) and np.array_equal(test_net.read_png_rgb(os.path.join(root, 'image_3', '000001.png')), right)
    cal = kitti_utils.read_obj_calibration(os.path.join(root, 'calib', '000006.txt'))
    assert abs(cal.p2[0, 0] - 721.5377) < 1e-9 and abs(cal.p2[0, 3] - cal.p3[0, 3] - (44.85728 + 339.5242)) < 1e-6
    assert fixture.KITTI_VAL_IDS == 3769


def test_bench_config3_val_list_replay_dry_run_world_2():
    """BASELINE configs[3] through the driver's own contract, two ranks over gloo, no GPU: the 3769-id list sharded i mod 2,
    every rank replays its frames from PNG files through test_net.run_split (injected detector), writes one KITTI file per
    frame, and the per-frame records of the whole job are gathered by one all_gather."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', '3', '--steps', '7', '--warmup', '1',
                          '--dry-run'], cwd=root, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and d['steps'] == 7 and d['dry_run'] is True and d['unit'] == 'stereo pairs/s' and d['value'] > 0
    assert c['baseline_config_index'] == 3 and c['val_ids'] == 3769 and 'configs[3]' in c['workload']
    assert c['records_gathered'] == [14, 301, 32] and c['result_files_rank0'] == 7 and c['objects_written_rank0'] >= 7
    hm, sat = c['host_ms_per_pair'], c['host_saturation']
    assert hm['png_decode_and_calib_parse'] > 0 and hm['result_files_and_record'] > 0 and hm['main_thread_busy'] > 0
    assert sat['LOCAL_WORLD_SIZE'] == 2 and sat['pairs_per_s_at_which_the_host_saturates'] > 0
    assert {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'roofline'} <= set(d)


def test_png_decode_worker_processes(tmp_path):
    """test_net._DecodeWorkers: png_worker.py processes decode into their slots of a shared file; several threads drive them at
    once (as run_split's prefetch threads do); pixels equal read_png_rgb's; a failing request raises in the caller and leaves the
    worker usable; the worker script imports neither torch nor the package."""
    import concurrent.futures as cf
    import os
    import numpy as np
    from stereo_rcnn_amd import fixture, test_net
    src = open(os.path.join(os.path.dirname(test_net.__file__), 'png_worker.py')).read()
    assert 'import torch' not in src and 'stereo_rcnn_amd' not in src.split('"""')[2]
    ids = fixture.write_kitti_tree(str(tmp_path), 7, distinct=3, height=48, width=160)
    w = test_net._DecodeWorkers(2)
    try:
        def one(f):
            i, (l, r) = w.decode((str(tmp_path / 'image_2' / (f + '.png')), str(tmp_path / 'image_3' / (f + '.png'))))
            try:
                return l.copy(), r.copy()
            finally:
                w.release(i)
        with cf.ThreadPoolExecutor(4) as ex:
            res = list(ex.map(one, ids))
        for f, (l, r) in zip(ids, res):
            assert np.array_equal(l, test_net.read_png_rgb(str(tmp_path / 'image_2' / (f + '.png'))))
            assert np.array_equal(r, test_net.read_png_rgb(str(tmp_path / 'image_3' / (f + '.png'))))
        with pytest.raises(RuntimeError, match='FileNotFoundError'):
            w.decode(('/nonexistent/left.png', '/nonexistent/right.png'))
        assert w.idle.qsize() == 2                         # both workers back in the idle set
        l, r = one(ids[0])
        assert np.array_equal(l, res[0][0])
        assert w.path is None                              # the shared file is unlinked once every worker has mapped it
    finally:
        w.close()
    assert all(pr.poll() is not None for pr in w.procs) or not w.procs
