#!/usr/bin/env python
"""Headline benchmark: stereo pairs/sec @1242x375 (600x1987 network input), ResNet-101 FPN,
batch=1, 300 proposals, no dense-align  (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config 2      # BASELINE configs[2]: batch = 8 pairs per forward, full 3-D pipeline incl. dense alignment
    python bench.py --config 3      # BASELINE configs[3]: the 3769-id KITTI val list replayed from PNG files, sharded i mod N, gathered
    python bench.py --config 4      # BASELINE configs[4]: ResNet-50 trunk, 2x resolution (network input 1200x3974), batch = 4
(same JSON contract; `config.workload` names the BASELINE entry; the default, --config 1, is the headline.)

One step = one pass of the hot path over one synthetic stereo pair per GPU (batch = 1 per forward; by default four
forwards are in flight per GPU, each on a HIP stream with a hardware queue of its own, `--streams 1` = strictly sequential, also reported):
  _StereoRCNN.forward (trunk+FPN on both eyes, stereo RPN, proposals, ROIAlign, heads)
  + detection decode + per-class NMS  (the reference's det_time region, demo.py:137-220, plus :231-257)
  + (under torch.distributed.run) the detection record of every step packed on the device and gathered over RCCL/xGMI,
    one all_gather per `--gather-every` steps on a side stream (all of them inside the timed region).
Inputs are preprocessed and resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md dense MFMA peaks (TFLOP/s) for the dtype each conv engine issues
PEAKS = {'f32': 157.3, 'f16x3': 2500.0}
ENGINE_DESC = {
    'f32': 'conv_mfma_kernel (implicit-GEMM conv, v_mfma_f32_32x32x2_f32, exact fp32)',
    'f16x3': 'conv_f16s_kernel (implicit-GEMM conv, both operands DMA-ed to LDS, 3x v_mfma_f32_32x32x16_f16 '
             'error-compensated split, fp32 accumulate)',
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='timed steps (default 100; --config 3: the rank\'s whole shard of the 3769-id val list)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--graph', action='store_true',
                    help='replay the forward as one hipGraph instead of launching eagerly (measured slower on MI355X: the '
                         'replay serialises the side-stream branches; eager launches are not CPU-bound here)')
    ap.add_argument('--no-graph', action='store_true', help='(default) launch eagerly')
    ap.add_argument('--no-program', action='store_true',
                    help='walk the Python launch code every forward instead of replaying the native launch list '
                         '(srcnn_program_run; default: replay -- same launches and streams, ~230 ctypes calls fewer per forward)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true',
                    help='skip the `parity` block (demo pair, 3-D boxes, dense-alignment indices against the reference goldens / oracle; ~25 s)')
    ap.add_argument('--no-sustained', action='store_true', help='skip the `sustained` leg (2 s soak + 300 steps of the headline loop)')
    ap.add_argument('--precision', choices=['f32', 'f16x3'], default='f16x3',
                    help="conv engine: f16x3 = fp32-class error-compensated split on the f16 MFMA (default, "
                         "passes the same parity tests), f32 = exact fp32 MFMA")
    ap.add_argument('--streams', type=int, default=0,
                    help='stereo pairs in flight per GPU: each forward is batch=1 on its own HIP stream and buffer set '
                         '(default 0 = the workload\'s own: 3 for --config 1 -- the other pairs fill the launch-synchronous phases and '
                         'the idle CUs of the small layers --, 2 batches for --config 2, 1 for --config 4); '
                         '--streams 1 = strictly one pair at a time (also reported as one_pair_at_a_time)')
    ap.add_argument('--plans', default='',
                    help='JSON file of autotuned conv plans: loaded if it exists (no tuning launches in this process), '
                         'written after the run otherwise (used to keep the rocprofv3 kernel statistics free of trial plans)')
    ap.add_argument('--gather-every', type=int, default=8,
                    help='(multi-GPU) steps whose detection records share one RCCL all_gather')
    ap.add_argument('--no-3d-leg', action='store_true',
                    help='skip the short full-3-D-flow measurement (configs[2] flow: + borders, 4-DoF solve, dense alignment, '
                         '3-DoF rectification) that the default line carries as `config.full_3d_flow`')
    ap.add_argument('--no-f32-leg', action='store_true',
                    help='skip the short exact-fp32-engine measurement that the default line carries as `engines.f32`')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / collective check without a GPU: every rank packs fake detection records on the CPU and '
                         'gathers them over gloo; prints the JSON line with value 0 (used by the CPU tests)')
    ap.add_argument('--config', type=int, default=1, choices=[1, 2, 3, 4],
                    help='BASELINE.json configs[] index of the workload: 1 = headline (batch 1, forward + decode + class NMS); '
                         '2 = batch 8 per forward + the whole 3-D flow (borders, 4-DoF solve, dense alignment, 3-DoF solve) per '
                         'image; 3 = the KITTI val list (3769 ids) replayed over synthetic PNG pairs through test_net.run_split: PNG decode, '
                         'H2D, fused preprocessing, forward, full 3-D flow, KITTI result files, records gathered (throughput only: no '
                         'dataset offline); 4 = ResNet-50 trunk at network input 1200x3974, batch 4, forward + decode + class NMS')
    ap.add_argument('--layers-out', default='',
                    help='write the full per-layer roofline table (every conv launch: shape, bytes, time, own bound) to this file; '
                         'the JSON line always carries the 15 layer groups that lose most time as roofline.layers')
    ap.add_argument('--no-pmc', action='store_true',
                    help='do not measure roofline.traffic in this run (default at N = 1, --config 1: two short child runs of '
                         'this script under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE`, about a minute); '
                         'the committed profiles/pmc_*_traffic.json is then quoted if it matches the kernel sources')
    ap.add_argument('--no-mix-layers', action='store_true',
                    help='skip the per-layer table of the headline regime itself (roofline.headline.layers: marginal in-mix cost of '
                         'every conv group + workgroup residency from the kernel\'s stamps, stereo_rcnn_amd/mix_table.py; ~30 s)')
    ap.add_argument('--mix-out', default='', help='write the in-mix per-layer tables (text) to this file')
    ap.add_argument('--pmc-child', type=int, default=0, help=argparse.SUPPRESS)     # internal: N forwards, one stream, exit
    ap.add_argument('--no-shipped-plans', action='store_true',
                    help='ignore stereo_rcnn_amd/plans/mi355x.json (conv plans tuned with the multi-stream step as objective, '
                         'tools/tune_headline.py) and tune every shape in situ, each launch alone on the chip')
    ap.add_argument('--tune', choices=['auto', 'isolated', 'concurrent'], default='auto',
                    help="objective of the conv engine's plan autotuner (engine.TUNE_MODE): 'isolated' = latency of the launch alone "
                         "on the chip, 'concurrent' = time per launch with as many copies in flight as the benchmark has batches in "
                         "flight; auto = concurrent for the multi-stream headline, isolated for one batch at a time (the "
                         "one_pair_at_a_time leg always runs on the isolated plan set)")
    ap.add_argument('--height', type=int, default=375)
    ap.add_argument('--width', type=int, default=1242)
    return ap.parse_args()


def cpu_baseline(cfg_id, height, width):
    """The CPU oracle (a port of the reference path; the reference itself cannot be imported or built on the GPU box) timed on
    the host cores on a bounded sample of the workload: three stereo pairs of the batch one after the other (one for configs[4]), full forward + decode + class NMS
    (det_time, demo.py:137-220); for --config 2 the sample leaves the 3-D stage out and says so."""
    from oracle import net as onet
    from oracle import postprocess as opost
    from stereo_rcnn_amd import fixture
    # oneDNN scales poorly past a few dozen threads on this network (measured on the 256-core GPU
    # host: 8-16 threads are fastest, 256 threads are ~100x slower), so the baseline uses <= 16.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    if cfg_id == 4:
        sd = fixture.make_state_dict(5, layers=fixture.R50)
        lu, ru = fixture.synthetic_pair(5, 750, 2484)
        l, sc = fixture.preprocess(lu, 1200, max_size=1 << 30)
        r, _ = fixture.preprocess(ru, 1200, max_size=1 << 30)
        info = torch.tensor([[l.shape[2], l.shape[3], sc]], dtype=torch.float32)
    else:
        sd = fixture.make_state_dict(3)
        l, r, info = fixture.make_inputs(3, height, width)
    n = 1 if cfg_id == 4 else 3             # ~10 s of wall time on the host cores either way
    t0 = time.time()
    for _ in range(n):
        out = onet.forward(sd, l, r, info)
        det = opost.decode_detections(out, info)
        opost.class_detections(det)
    dt = time.time() - t0
    return {'value': n / dt, 'unit': 'stereo pairs/s', 'cores': cores, 'kind': 'port',
            'sample': '%d stereo pair(s) %dx%d of the batch (network input %dx%d), batch 1, one after the other: full forward + decode + '
                      'class NMS%s, %.1f s' % (n, width, height, l.shape[3], l.shape[2],
                                               ' (3-D stage not in the sample)' if cfg_id == 2 else '', dt)}


def csrc_hash():
    """sha256 over the kernel sources: stamps PMC-derived files so that a stale one is detected (profiles/pmc_*_traffic.json)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'stereo_rcnn_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(base, '*.hip')) + glob.glob(os.path.join(base, '*.h'))):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


# kernel names of the conv engine's launches (single, grouped, chained; the stem's fp32-input form)
CONV_KERNELS = ('conv_f16s_kernel', 'conv_f16x3_kernel', 'conv_group_kernel', 'conv_chain_kernel')


def measure_traffic_live(plans_path, steps=3, timeout_s=240, per_kernel=None):
    """HBM-side bytes of the conv engine per step, measured in THIS run on THIS box: two child processes of this script
    (`--pmc-child`: plans preloaded, `steps` forwards one at a time, nothing else) under rocprofv3's counter collection, one
    pass per counter as the MI355X guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Returns
    (fetch_kb_per_step, write_kb_per_step, conv_launches_per_step) or None when rocprofv3 is unavailable or a pass fails.
    per_kernel (a dict, filled in place): short kernel name -> {'calls', 'us', 'fetch_kb', 'write_kb'} per forward for EVERY kernel
    of the forward (durations from the FETCH_SIZE pass's kernel trace: every launch alone on the chip) -- roofline.non_conv."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None
    out = {}
    launches = None

    def short(name):
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*$', '', name)                     # drop the argument list
        return name.replace('srcnn::', '')

    def steady(rows, key):
        """the rows of the `steps` measured forwards: the child brackets them with null_kernel marker launches (the warm-up
        forwards before the first marker -- calibration on the fp32 engine, in-situ tuning -- are not the forward)"""
        rows = sorted(rows, key=key)
        marks = [i for i, r in enumerate(rows) if 'null_kernel' in r['Kernel_Name']]
        if len(marks) != steps + 1:
            raise ValueError('expected %d markers, found %d' % (steps + 1, len(marks)))
        return [r for i, r in enumerate(rows) if marks[0] < i < marks[-1] and 'null_kernel' not in r['Kernel_Name']]

    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='srcnn_pmc_', dir='/tmp')
        try:
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'p', '--', sys.executable,
                   os.path.abspath(__file__), '--pmc-child', str(steps), '--plans', plans_path]
            env = dict(os.environ, TMPDIR='/tmp')
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None
            total, seen = 0.0, set()
            crow = [row for f in files for row in csv.DictReader(open(f)) if row['Counter_Name'] == counter]
            for row in steady(crow, lambda q: int(q['Dispatch_Id'])):
                name = row['Kernel_Name']
                if any(k in name for k in CONV_KERNELS):
                    total += float(row['Counter_Value'])
                    seen.add(row['Dispatch_Id'])
                if per_kernel is not None:
                    e = per_kernel.setdefault(short(name), {'calls': 0.0, 'us': 0.0, 'fetch_kb': 0.0, 'write_kb': 0.0})
                    e['fetch_kb' if counter == 'FETCH_SIZE' else 'write_kb'] += float(row['Counter_Value']) / steps
            if per_kernel is not None and counter == 'FETCH_SIZE':
                trow = [row for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True) for row in csv.DictReader(open(f))]
                for row in steady(trow, lambda q: int(q['Start_Timestamp'])):
                    e = per_kernel.setdefault(short(row['Kernel_Name']), {'calls': 0.0, 'us': 0.0, 'fetch_kb': 0.0, 'write_kb': 0.0})
                    e['calls'] += 1.0 / steps
                    e['us'] += (float(row['End_Timestamp']) - float(row['Start_Timestamp'])) / 1e3 / steps
            out[counter] = total / steps
            launches = len(seen) // steps
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out['FETCH_SIZE'], out['WRITE_SIZE'], launches


# non-conv kernel families of one forward (roofline.non_conv): label -> (kernel-name prefixes, what bounds it)
NON_CONV_FAMILIES = [
    ('stem_pack', ('stem_pack',), 'hbm'),
    ('max_pool', ('maxpool3x3s2',), 'hbm'),
    ('fpn top-down add', ('upsample_add', 'subsample2'), 'hbm'),
    ('split-K reductions', ('splitk_reduce',), 'hbm'),
    ('rpn scores', ('rpn_score',), 'hbm'),
    ('proposal: top-6000 selection', ('tk_hist', 'tk_compact', 'tk_rank', 'gather_decode', 'intersect_pad'), 'latency'),
    ('proposal: NMS mask', ('pair_mask',), 'hbm'),
    ('proposal: NMS greedy scan', ('greedy_scan',), 'latency'),
    ('roi_align', ('pyramid_roi_align',), 'hbm'),
    ('head tails', ('box_head_tail', 'kpts_tail'), 'hbm'),
    ('decode + class NMS', ('decode_detections', 'class_select_sort', 'map_keep'), 'latency'),
]
HBM_ACHIEVABLE = 6.3e12          # B/s: MI355X_MICROARCH.md (float4 copy; the peak the guide quotes against)


def non_conv_algorithmic_mb(plan):
    """Compulsory bytes (MB) of the streaming non-conv families from the plan's shapes: every input element read once, every output
    written once, 4 B each (both activation formats)."""
    N, B = plan.N, plan.B
    sh, sw = plan.stem_hw
    ph, pw = plan.c1_hw
    hw = plan.layer_hw
    tops = [hw[3], hw[2], hw[1]]
    fpn = sum(N * 256 * 4 * (t[0] * t[1] + 2 * l[0] * l[1]) for t, l in zip(tops, [hw[2], hw[1], hw[0]]))
    fpn += N * 256 * 4 * (plan.rpn_shapes[4][0] * plan.rpn_shapes[4][1]) * 2                      # P6 = subsampled P5
    rpn = sum(max(n, 1) * B * a * b * 24 * 4 for n, (a, b) in zip(plan.rpn_nparts, plan.rpn_shapes)) + B * plan.A * 8 * 4
    G = plan.kp_logits.shape[1]
    return {'max_pool': N * 64 * 4 * (sh * sw + ph * pw) / 1e6, 'fpn top-down add': fpn / 1e6, 'rpn scores': rpn / 1e6,
            'stem_pack': (2 * B * 3 * plan.H * plan.W * 4 + N * (plan.H + 6) * (plan.W + 8) * 16) / 1e6,
            'head tails': (plan.R * G * G * 6 * 4 + plan.R * (4 * G + 2 * G) * 4 + 2 * plan.R * plan.w.fc.cout * 4) / 1e6}


def non_conv_table(per_kernel, alg_mb=None):
    """roofline.non_conv from measure_traffic_live's per-kernel rows: per family the launches, the time alone on the chip, the
    HBM-side bytes the PMC counters saw (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction), GB/s and the fraction of 6.3 TB/s."""
    rows, used = [], set()
    for label, prefixes, bound in NON_CONV_FAMILIES:
        ks = [k for k in per_kernel if any(k.startswith(p) for p in prefixes)]
        if not ks:
            continue
        used.update(ks)
        us = sum(per_kernel[k]['us'] for k in ks)
        by = sum(2.0 * per_kernel[k]['fetch_kb'] + per_kernel[k]['write_kb'] for k in ks) * 1024.0
        alg = (alg_mb or {}).get(label)
        use = alg * 1e6 if alg else by                      # the roofline figure uses the compulsory bytes where the shapes give them
        rows.append({'family': label, 'kernels': sorted(ks), 'launches': round(sum(per_kernel[k]['calls'] for k in ks), 1), 'us': round(us, 1),
                     'algorithmic_mb': (round(alg, 2) if alg else None), 'hbm_side_mb_pmc': round(by / 1e6, 2),
                     'gb_per_s': round(use / max(us, 1e-9) / 1e3, 1),
                     'frac_of_6.3_tb_s': round(use / max(us, 1e-9) * 1e6 / HBM_ACHIEVABLE, 4), 'bound': bound})
    other = [k for k in per_kernel if k not in used and not any(c in k for c in CONV_KERNELS)]
    if other:
        us = sum(per_kernel[k]['us'] for k in other)
        by = sum(2.0 * per_kernel[k]['fetch_kb'] + per_kernel[k]['write_kb'] for k in other) * 1024.0
        rows.append({'family': 'other (runtime fills / copies)', 'kernels': sorted(other), 'launches': round(sum(per_kernel[k]['calls'] for k in other), 1),
                     'us': round(us, 1), 'algorithmic_mb': None, 'hbm_side_mb_pmc': round(by / 1e6, 2), 'gb_per_s': round(by / max(us, 1e-9) / 1e3, 1),
                     'frac_of_6.3_tb_s': round(by / max(us, 1e-9) * 1e6 / HBM_ACHIEVABLE, 4), 'bound': 'latency'})
    return {'rows': sorted(rows, key=lambda r: -r['us']), 'total_us': round(sum(r['us'] for r in rows), 1),
            'note': 'every non-conv kernel of one step (forward + decode + class NMS), each launch ALONE on the chip (one pair at a time, in-situ '
                    'plans: rocprofv3 --kernel-trace --pmc children of this run, marker-bracketed steady-state steps).  gb_per_s = compulsory bytes '
                    '(`algorithmic_mb`: every input read once, every output written once) / time where the shapes give them, else the HBM-side '
                    'traffic the PMC counters saw (`hbm_side_mb_pmc` = 2 x FETCH_SIZE + WRITE_SIZE; the x 2 is calibrated for 16-byte-per-lane '
                    'streaming reads only, MI355X_MICROARCH.md).  `latency` families are serial / few-workgroup kernels whose time is not their '
                    'bytes; split-K reductions exist in this execution only (the shipped plans of the headline split K less often)'}


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: become the driver's own launch line
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>), one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def dry_run(args, rank, world):
    """No GPU: exercises exactly the multi-process plumbing of the benchmark -- env contract, process group, the packed
    detection records of `--gather-every` steps in one all_gather, barrier + max-over-ranks timing -- over gloo, and the HOST
    side of the full-3-D-flow leg as every rank runs it: solver-thread budget from LOCAL_WORLD_SIZE, CPU pinning, the 4-DoF
    Newton-CG solves of a synthetic detection record in the library's host build (no device call), results gathered."""
    import numpy as np
    from stereo_rcnn_amd import _lib
    from stereo_rcnn_amd import distributed as sdist
    use_dist = 'RANK' in os.environ
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    G = max(1, args.gather_every)
    buf = torch.zeros((G, 301, sdist.REC_COLS))
    seen = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        buf[k % G, 0, 0] = 1.0
        buf[k % G, 1, 0] = float(rank)
        if k % G == G - 1 or k == args.steps - 1:
            out, work = sdist.gather_detections(buf)
            if work is not None:
                work.wait()
            assert out.shape[0] == world and sorted(out[:, 0, 1, 0].tolist()) == [float(r) for r in range(world)]
            seen += 1
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if use_dist:
        dist.barrier()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # ---- host side of the full-3-D leg: this rank's share of the cores, pinned, 4-DoF solves of 24 synthetic detections
    try:
        mine = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        mine = list(range(os.cpu_count() or 1))
    lw = sdist.local_world_size()
    share = mine[rank % lw::lw] or mine                       # no GPU here to ask for a NUMA node: an even split of the mask
    pin = sdist.pin_to_gpu_numa(0, cpus=share)
    threads = sdist.host_solver_threads()
    rng = np.random.default_rng(7 + rank)
    n = 24
    rec = np.zeros((301, sdist.REC_COLS), np.float32)
    rec[0, 0] = n
    z = rng.uniform(8, 40, n)
    x = rng.uniform(-6, 6, n)
    u = 721.5377 * x / z + 609.5593
    half = 721.5377 * 1.0 / z
    rec[1:n + 1, 0] = 0.9
    rec[1:n + 1, 1], rec[1:n + 1, 3] = u - half, u + half
    rec[1:n + 1, 2], rec[1:n + 1, 4] = 172.854 - 0.2 * half, 172.854 + 1.3 * half
    disp = 721.5377 * 0.54 / z
    rec[1:n + 1, 5], rec[1:n + 1, 7] = u - half - disp, u + half - disp
    rec[1:n + 1, 6], rec[1:n + 1, 8] = rec[1:n + 1, 2], rec[1:n + 1, 4]
    rec[1:n + 1, 9:12] = (1.6, 1.5, 3.9)
    rec[1:n + 1, 12], rec[1:n + 1, 13] = 0.1, 0.99
    rec[1:n + 1, 14] = u
    rec[1:n + 1, 15] = 1
    rec[1:n + 1, 16] = 0.9
    rec[1:n + 1, 17], rec[1:n + 1, 18] = u - half, u + half
    rt = torch.from_numpy(rec)
    state = torch.zeros((300, 4), dtype=torch.float64)
    ts = time.perf_counter()
    _lib.check(_lib.lib().srcnn_solve_4dof_records_host(rt.data_ptr(), 300, sdist.REC_COLS, 375, 1242, 721.5377, 609.5593, 172.854,
                                                        44.85728 + 339.5242, 0.05, state.data_ptr(), threads),
               "srcnn_solve_4dof_records_host")
    solve_ms = (time.perf_counter() - ts) * 1e3
    solved = int((rt[1:n + 1, 20] > 0).sum())
    mine_rec = torch.tensor([float(rank), float(threads), float(len(pin and share or mine)), float(solved), solve_ms])
    allr, work = sdist.gather_detections(mine_rec)
    if work is not None:
        work.wait()
    if rank == 0:
        assert world == args.gpus, "launched with %d ranks for --gpus %d" % (world, args.gpus)
        print(json.dumps({'metric': 'stereo pairs/sec @1242x375 ResNet-101', 'value': 0.0, 'unit': 'stereo pairs/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 0.0,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none', 'data': 'synthetic',
                          'config': {'workload': 'DRY RUN (no GPU): launcher + gloo all_gather of %d fake record batches' % seen,
                                     'parallelism': 'pairs sharded 1/rank, one all_gather of the detection records per %d steps' % G,
                                     'full_3d_flow_host_side': [{'rank': int(r[0]), 'host_solver_threads': int(r[1]), 'cpus_after_pinning': int(r[2]),
                                                                 'solved_of_24': int(r[3]), 'solve_ms': round(float(r[4]), 2)}
                                                                for r in allr.tolist()]},
                          'roofline': None, 'dry_run': True}), flush=True)
    if use_dist:
        dist.destroy_process_group()


CONFIG3_TEXT = ('BASELINE configs[3]: KITTI val list (%(n)d ids, data/kitti/splits/val.txt) replayed over %(d)d synthetic %(w)dx%(h)d PNG stereo '
                'pairs (no dataset offline: throughput only, AP not checkable), ids sharded i mod N over %(world)d GPU(s); per frame: PNG '
                'decode (host threads) -> H2D -> fused preprocessing -> ResNet-101 FPN forward -> decode + class NMS -> keypoints of the '
                'kept detections -> borders -> 4-DoF Newton-CG (host C threads) -> dense alignment -> 3-DoF Newton-CG -> KITTI result '
                'file (stereo_rcnn_amd.test_net.run_split); per-frame records gathered by one all_gather at the end')


def _fake_objects_from_pixels(left):
    """Dry run only (no GPU): deterministic fake detections from the decoded frame, so that the writer and the gather carry data."""
    import numpy as np
    rng = np.random.default_rng(int(left[0, 0, 0]) + 1)
    objs = []
    for i in range(int(left[0, 0, 0]) % 3 + 1):
        x1, y1 = rng.uniform(0, 300), rng.uniform(20, 80)
        objs.append({'score': float(rng.uniform(0.1, 1)), 'box_left': np.array([x1, y1, x1 + 40, y1 + 30], np.float32),
                     'box_right': np.array([x1 - 8, y1, x1 + 32, y1 + 30], np.float32), 'dim': np.array([1.6, 1.5, 4.0]),
                     'alpha': 0.3 * i, 'kpts': np.array([x1 + 5, 1, 0.9, x1, x1 + 40], np.float32),
                     'xyz_init': rng.uniform(-5, 30, 3), 'theta_init': 0.1, 'xyz': rng.uniform(-5, 30, 3), 'theta': 0.2 + i,
                     'aligned': True, 'disparity': float(rng.uniform(5, 60)), 'roi_index': i})
    return objs


def run_config3(args, rank, local_rank, world, use_dist):
    """BASELINE configs[3] (the reference's test_net.py:109-136,217-225,329-334 over data/kitti/splits/val.txt): one step = one
    frame of the rank's shard through stereo_rcnn_amd.test_net.run_split -- everything between the PNG files and the KITTI result
    files -- followed by the gather of the per-frame records.  Same JSON contract; `config.host_ms_per_pair` splits the host side
    (decode / H2D issue / solves / result files / waiting for the GPU) and `config.host_saturation` says at how many pairs/s one
    rank's host budget (cores / LOCAL_WORLD_SIZE) is used up -- the scaling limiter SURVEY 8(e) names."""
    import shutil
    import tempfile
    from stereo_rcnn_amd import fixture, test_net
    from stereo_rcnn_amd import distributed as sdist
    dry = bool(args.dry_run)
    if not dry:
        from stereo_rcnn_amd import serving
        serving.before_hip()
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo' if dry else 'nccl')
    n_ids = fixture.KITTI_VAL_IDS
    all_idx = sdist.shard_indices(n_ids, rank, world)
    K = len(all_idx) if args.steps is None else int(args.steps)
    W = max(0, int(args.warmup))
    distinct = 4 if dry else 8
    h, w = (40, 120) if dry else (args.height, args.width)
    base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
    root = tempfile.mkdtemp(prefix='srcnn_val_r%d_' % rank, dir=base)
    res_dir = os.path.join(root, 'results')
    try:
        # every rank replays its own copy of the tree (tmpfs); only the ids it owns (and their pool files) are ever opened
        need = sorted(set(all_idx[i % len(all_idx)] for i in range(max(K, 1))) | set(all_idx[:max(W, 8)]))
        ids = fixture.write_kitti_tree(root, n_ids, distinct, h, w)
        mine = [ids[all_idx[i % len(all_idx)]] for i in range(K)]
        warm = [ids[i] for i in all_idx[:max(W, 8)]]
        del need
        lw = sdist.local_world_size()
        threads = sdist.host_solver_threads()
        try:
            cpus = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cpus = os.cpu_count() or 1
        budget = max(1, cpus // lw)
        prefetch = int(os.environ.get('SRCNN_DECODE_THREADS', max(4, min(16, budget))))     # PNG decode threads (PIL releases the GIL while inflating)
        timers, ptimers = {}, {}
        records = []
        if dry:
            dev, model, detect = None, None, (lambda frames: (_fake_objects_from_pixels(f[0]) for f in frames))
            slots, roof, numa = 0, None, None
            test_net.run_split(None, root, warm[:2], res_dir, None, detect_stream=detect, prefetch=2)
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            test_net.run_split(None, root, mine, res_dir, None, detect_stream=detect, prefetch=2, records=records, timers=timers)
        else:
            from stereo_rcnn_amd import _lib, engine, pipeline
            from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
            torch.cuda.set_device(local_rank)
            dev = torch.device('cuda', local_rank)
            _lib.lib()
            numa = sdist.pin_to_gpu_numa(local_rank) if use_dist and world > 1 else None
            pipeline.HOST_SOLVER_THREADS = threads
            model = resnet(('__background__', 'Car'), 101, pretrained=False)
            model.create_architecture()
            model.load_state_dict(fixture.make_state_dict(3))
            model.cuda()
            model.eval()
            model.precision = args.precision
            model.use_program = not args.no_program
            slots = max(1, args.streams) if args.streams > 0 else 4
            test_net.run_split(model, root, warm, res_dir, dev, solver='host', slots=slots, prefetch=prefetch)   # calibration, tuning, programs
            # algorithmic conv work of one frame as this flow runs it (keypoint tower on the kept detections only): one eager frame
            # with the counter on; the row-limited launches are counted at the share of their rows that ran
            from stereo_rcnn_amd.model.utils import kitti_utils
            lu = torch.from_numpy(test_net.read_png_rgb(os.path.join(root, 'image_2', warm[0] + '.png'))).to(dev)
            ru = torch.from_numpy(test_net.read_png_rgb(os.path.join(root, 'image_3', warm[0] + '.png'))).to(dev)
            calib = kitti_utils.read_obj_calibration(os.path.join(root, 'calib', warm[0] + '.txt'))
            prog, model.use_program = model.use_program, False
            engine.FlopCounter.enabled, engine.FlopCounter.flops, engine.FlopCounter.launches, engine.FlopCounter.bytes = True, 0.0, 0, 0.0
            engine.FlopCounter.rows = []
            objs0 = pipeline.detect_3d_images(model, lu, ru, calib, slot=slots + 1)
            torch.cuda.synchronize()
            rows, engine.FlopCounter.rows, engine.FlopCounter.enabled = engine.FlopCounter.rows, None, False
            model.use_program = prog
            kept = int(pipeline._stage(300, dev, slots + 1).rec_host[0, 0])       # detections class NMS kept in that frame
            flops_pair = sum(r['flops'] * ((kept / 300.0) if r['name'].startswith('kpts') else 1.0) for r in rows)
            roof = {'flops_pair': flops_pair, 'kept': kept, 'objects': len(objs0)}
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            pipeline.TIMERS = ptimers
            t0 = time.perf_counter()
            test_net.run_split(model, root, mine, res_dir, dev, solver='host', slots=slots, prefetch=prefetch, records=records,
                               timers=timers)
            pipeline.TIMERS = None
        t_split = time.perf_counter() - t0
        # the gather of the split: this rank's K records (frame j of the job lives on rank j % world), ONE all_gather
        recs = records if dry else [r.to(dev) for r in records]
        full = sdist.gather_split_records(recs, K * world, rank, world)
        if not dry:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed, t_split], dtype=torch.float64, device=dev if (use_dist and not dry) else 'cpu')
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el[0])
        n_files = len(os.listdir(os.path.join(res_dir, 'data')))
        assert int(full.shape[0]) == K * world and n_files >= min(K, len(all_idx)), (tuple(full.shape), n_files)
        objects = int(sum(float(r[0, 0]) for r in records))
        per = lambda key, src: round(src.get(key, 0.0) * 1e3 / max(K, 1), 3)
        # the loop thread's own blocking: waiting for a pair's worker (async host phases: 'main_wait_s') or, in round 5's arrangement,
        # in front of the device itself ('gpu_wait_s', which the workers accumulate now)
        async_host = 'main_wait_s' in ptimers                  # (the dry run drives an injected detector: neither timer exists)
        main_wait = ptimers.get('main_wait_s', 0.0) if async_host else ptimers.get('gpu_wait_s', 0.0)
        main_busy = max(0.0, timers.get('loop_s', 0.0) - main_wait)
        host_ms = {'png_decode_and_calib_parse': per('decode_s', timers), 'png_decode_threads': prefetch,
                   'png_decode_in': 'worker processes (one per prefetch thread, stereo_rcnn_amd/png_worker.py)' if (not dry and test_net.DECODE_PROCESSES and test_net.ZERO_COPY_IMAGES) else 'threads of the loop process',
                   'result_files_on': 'a writer thread' if test_net.WRITER_THREAD else 'the loop thread',
                   'h2d_issue': per('h2d_s', timers), 'newton_cg_solves_wall': per('solve_s', ptimers), 'host_solver_threads': threads,
                   'result_files_and_record': per('write_s', timers), 'waiting_for_the_gpu': round(main_wait * 1e3 / max(K, 1), 3),
                   'workers_waiting_for_the_gpu': per('gpu_wait_s', ptimers), 'async_host_phases': bool(async_host),
                   'main_thread_busy': round(main_busy * 1e3 / max(K, 1), 3),
                   'note': 'ms per pair on THIS rank; decode is the wall time of a frame\'s two PNGs + calibration summed over the prefetch threads (each sleeps on its '
                           'worker process meanwhile; they run ahead of the loop); with async_host_phases the Newton-CG solves and the waits in '
                           'front of a pair\'s device stages run on one worker thread per pair in flight; '
                           'main_thread_busy = loop wall time minus the time the loop thread was blocked = launching + Python'}
        dec_rate = prefetch * 1e3 / max(host_ms['png_decode_and_calib_parse'], 1e-6)
        main_rate = 1e3 / max(host_ms['main_thread_busy'], 1e-6)
        cpu_ms = host_ms['png_decode_and_calib_parse'] + host_ms['main_thread_busy'] + host_ms['newton_cg_solves_wall'] * max(threads - 1, 0)
        saturation = {'cores_budget_per_rank': budget, 'LOCAL_WORLD_SIZE': lw,
                      'decode_bound_pairs_per_s': round(dec_rate, 1), 'main_thread_bound_pairs_per_s': round(main_rate, 1),
                      'core_seconds_bound_pairs_per_s': round(budget * 1e3 / max(cpu_ms, 1e-6), 1),
                      'pairs_per_s_at_which_the_host_saturates': round(min(dec_rate, main_rate, budget * 1e3 / max(cpu_ms, 1e-6)), 1),
                      'note': 'one rank: decode threads x 1 / decode ms; the single-threaded loop (launches + solves + files); and the '
                              'rank\'s core budget / CPU ms per pair (solver threads counted as busy for the solve wall time) -- the smallest '
                              'is where the host side of this rank stops scaling'}
        if rank == 0:
            nh, nw_ = (0, 0)
            roofline = None
            if not dry:
                from stereo_rcnn_amd import engine as _e
                nh, nw_, _ = _e.preprocess_size(h, w, 600)
                ach = roof['flops_pair'] * (K * world / elapsed) / world / 1e12          # per GPU
                roofline = {'bound': 'mfma', 'kernel': ENGINE_DESC[args.precision], 'achieved': round(ach, 2), 'peak': PEAKS[args.precision],
                            'unit': 'TFLOP/s', 'frac': round(ach / PEAKS[args.precision], 4), 'traffic': None,
                            'algorithmic_gflop_per_step': round(roof['flops_pair'] / 1e9, 1),
                            'execution': 'whole-flow mode: algorithmic conv FLOPs of one frame as this flow runs it (keypoint tower on the %d '
                                         'detections class NMS kept of 300 rois) x pairs/s per GPU; the kernel-level figure (HIP events per '
                                         'launch, PMC traffic) is `python bench.py` (--config 1), same kernels' % roof['kept']}
            res = {'metric': 'stereo pairs/sec @1242x375 ResNet-101', 'value': round(K * world / elapsed, 3), 'unit': 'stereo pairs/s',
                   'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(elapsed / max(K, 1) * 1e3, 3), 'higher_is_better': True,
                   'scaling': 'weak' if args.steps is not None else 'strong', 'vs_baseline': None,
                   'dtype': 'none' if dry else ('f32' if args.precision == 'f32' else 'f32 result via 3xf16 split MFMA (f32 accumulate)'),
                   'data': 'synthetic',
                   'config': {'workload': ('DRY RUN (no GPU, injected detector): ' if dry else '') + CONFIG3_TEXT
                              % {'n': n_ids, 'd': distinct, 'w': w, 'h': h, 'world': world},
                              'baseline_config_index': 3, 'pairs_per_step': 1, 'frames_per_rank': K, 'val_ids': n_ids,
                              'network_input': [nh, nw_], 'pairs_in_flight': slots, 'solver': 'host (C threads, bit-identical to scipy)',
                              'keypoints_on_kept_detections_only': True, 'objects_written_rank0': objects,
                              'result_files_rank0': n_files, 'records_gathered': [int(v) for v in full.shape],
                              'split_ms_per_pair_before_gather': round(float(el[1]) / max(K, 1) * 1e3, 3),
                              'gather_ms_total': round((elapsed - float(el[1])) * 1e3, 2),
                              'host_ms_per_pair': host_ms, 'host_saturation': saturation, 'numa_pinning': numa,
                              'weights': 'seeded random init, reference state_dict schema',
                              'parallelism': ('ids sharded i mod %d, weights replicated, no data-path collective; one all_gather of the '
                                              'per-frame records at the end' % world) if use_dist else 'single GPU',
                              'scaling_note': 'default --steps = the rank\'s whole shard of the %d ids (total work fixed: strong); '
                                              '--steps K = K frames per rank (weak)' % n_ids},
                   'roofline': roofline}
            if dry:
                res['dry_run'] = True
            elif not args.no_cpu_baseline and world == 1:
                res['cpu_baseline'] = cpu_baseline(2, h, w)
            print(json.dumps(res), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()


WORKLOADS = {
    1: dict(layers=101, batch=1, streams=4, flow='2d',
            text='BASELINE configs[1]: ResNet-101 FPN, batch=1 stereo pair per GPU, %(w)dx%(h)d synthetic (network input %(nw)dx%(nh)d), '
                 '300 proposals, forward + decode + class NMS, no dense-align'),
    2: dict(layers=101, batch=8, streams=2, flow='3d',
            text='BASELINE configs[2]: ResNet-101 FPN + stereo RPN + ROIAlign + dense_align, batch=8 stereo pairs per forward, '
                 '%(w)dx%(h)d synthetic (network input %(nw)dx%(nh)d), full pipeline per image: decode + class NMS + borders + 4-DoF '
                 'Newton-CG (host C threads) + dense alignment + 3-DoF Newton-CG'),
    4: dict(layers=50, batch=4, streams=2, flow='2d',
            text='BASELINE configs[4]: ResNet-50 backbone, 2x input resolution (%(w)dx%(h)d synthetic -> network input %(nw)dx%(nh)d), '
                 'batch=4 stereo pairs per forward, 300 proposals per image, forward + decode + class NMS (HBM-bound stress)'),
}


def make_batch(cfg_id, rank, height, width, dev):
    """Synthetic, preprocessed inputs of one step of the workload, resident in HBM: (im_left, im_right, im_info) of B pairs."""
    from stereo_rcnn_amd import fixture
    B = WORKLOADS[cfg_id]['batch']
    if cfg_id == 4:
        parts = []
        for b in range(B):
            lu, ru = fixture.synthetic_pair(5 + rank * B + b, 750, 2484)
            tl, sc = fixture.preprocess(lu, 1200, max_size=1 << 30)
            tr, _ = fixture.preprocess(ru, 1200, max_size=1 << 30)
            parts.append((tl, tr, torch.tensor([[tl.shape[2], tl.shape[3], sc]], dtype=torch.float32)))
    else:
        parts = [fixture.make_inputs(3 + rank * B + b, height, width) for b in range(B)]
    return [torch.cat([q[k] for q in parts], 0).to(dev) for k in range(3)]


def demo_calib():
    """KITTI object calibration of the reference's demo pair (demo/calib.txt): the 3-D stage's camera."""
    import numpy as np
    from stereo_rcnn_amd.model.utils import kitti_utils
    calib = kitti_utils.FrameCalibrationData()
    calib.p2 = np.array([721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884]).reshape(3, 4)
    calib.p3 = np.array([721.5377, 0, 609.5593, -339.5242, 0, 721.5377, 172.854, 2.199936, 0, 0, 1, 0.002729905]).reshape(3, 4)
    calib.t_cam2_cam0 = np.array([calib.p2[0, 3] / calib.p2[0, 0], 0, 0])
    return calib


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        relaunch_under_torchrun(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    use_dist = 'RANK' in os.environ          # launched by torch.distributed.run (any world size, incl. 1)
    if args.config == 3:
        return run_config3(args, rank, local_rank, world, use_dist)
    if args.steps is None:
        args.steps = 100
    if args.dry_run:
        return dry_run(args, rank, world)
    assert world == args.gpus or not use_dist, "launched with %d ranks for --gpus %d" % (world, args.gpus)
    # every forward in flight on a hardware queue of its own: GPU_MAX_HW_QUEUES is read once, when HIP starts
    # (stereo_rcnn_amd/serving.py -- the same regime set-up the product's streamed entry points use)
    from stereo_rcnn_amd import serving
    from stereo_rcnn_amd import streams as sstreams
    queues_ok = serving.before_hip()
    if use_dist and world > 1:
        # first multi-GPU runs must be diagnosable: under torchrun every rank is its own process and reads the variable itself
        assert queues_ok and int(os.environ.get('GPU_MAX_HW_QUEUES', '0')) >= sstreams.HW_QUEUES, \
            'rank %d: GPU_MAX_HW_QUEUES=%r before HIP init (needs >= %d: one hardware queue per forward in flight)' \
            % (rank, os.environ.get('GPU_MAX_HW_QUEUES'), sstreams.HW_QUEUES)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from stereo_rcnn_amd import _lib, engine, fixture, layer_table, pipeline
    from stereo_rcnn_amd import distributed as sdist
    from stereo_rcnn_amd import postprocess as hpost
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

    _lib.lib()    # fail loudly right here if the HIP library is missing
    # `value` is measured with the keypoint branch computed for ALL 300 rois of every forward, as the reference's forward
    # does; the product's default (pipeline.LAZY_KPTS: keypoints for the detections that survive class NMS only -- identical
    # detections) is reported beside it as `config.keypoints_on_kept_detections_only`, never as `value`
    lazy_default = pipeline.LAZY_KPTS
    pipeline.LAZY_KPTS = False
    numa = sdist.pin_to_gpu_numa(local_rank) if use_dist and world > 1 else None    # host solver threads stay near their GPU
    wl = WORKLOADS[args.config]
    B = wl['batch']
    model = resnet(('__background__', 'Car'), wl['layers'], pretrained=False)
    model.create_architecture()
    model.load_state_dict(fixture.make_state_dict(3) if wl['layers'] == 101 else fixture.make_state_dict(5, layers=fixture.R50))
    model.cuda()
    model.eval()
    use_graph = bool(args.graph) and not args.no_graph
    model.use_graph = use_graph
    model.use_program = not args.no_program and not use_graph
    model.precision = args.precision
    # every rank works on its own synthetic pairs (weak scaling: per-GPU work is fixed)
    im_l, im_r, im_info = make_batch(args.config, rank, args.height, args.width, dev)
    src_h, src_w = (750, 2484) if args.config == 4 else (args.height, args.width)
    calib = demo_calib()
    gather_stream = torch.cuda.Stream() if use_dist else None

    S = max(1, args.streams if args.streams > 0 else wl['streams'])
    # 'auto' = isolated: timing a launch beside copies of itself picks plans that lose the real mix (profiles/tune_objective_r04.txt);
    # the multi-stream regime is tuned by the measured step itself instead (stereo_rcnn_amd/tune.py -> the shipped plan file)
    tune_mode = args.tune if args.tune != 'auto' else 'isolated'
    engine.set_tune_mode(tune_mode, S)
    # the serving regime (S > 1: branches stay on the forwards' main streams, shipped throughput-tuned plans adopted): the SAME
    # call pipeline.detect_3d_stream / test_net.py make -- the benchmark measures what the product runs
    want_plans = not args.no_shipped_plans and not args.plans and tune_mode == 'isolated' and args.precision == 'f16x3'
    if not want_plans:
        serving.USE_SHIPPED_PLANS = False          # ... nor may a later leg (the 3-D flow enters the regime itself) adopt them
    regime = serving.enter(S, device=dev, plans=want_plans)
    shipped = regime['shipped_plans'] if want_plans else 0
    plans_loaded = bool(args.plans) and os.path.exists(args.plans) and engine.load_plans(args.plans) > 0
    if args.pmc_child:              # counter-collection child of measure_traffic_live(): forwards only, one at a time
        assert plans_loaded, "--pmc-child needs the parent's tuned plans"
        model.use_program = False
        with torch.no_grad():
            Lc = _lib.lib()
            Lc.srcnn_debug_null_launches.argtypes, Lc.srcnn_debug_null_launches.restype = [ctypes.c_int, ctypes.c_void_p], ctypes.c_int
            for k in range(args.pmc_child + 2):
                if k >= 2:                                       # two warm-up forwards (calibration, in-situ tuning), then marker-bracketed ones
                    Lc.srcnn_debug_null_launches(1, _lib.stream())
                o = model(im_l, im_r, im_info)
                det = hpost.decode_detections(o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], im_info[0:1])     # the step's tail: decode + class NMS
                hpost.class_nms_device(det, 1, 0.05)
                torch.cuda.synchronize()
            Lc.srcnn_debug_null_launches(1, _lib.stream())
            torch.cuda.synchronize()
        return
    streams = sstreams.main_streams(S) if S > 1 else [None]

    # detections of G consecutive steps (B images each) are packed into one buffer and gathered by ONE RCCL all_gather
    # (two buffers alternate so that a gather in flight on the side stream never races the next writes)
    G = max(1, args.gather_every)
    n_rec = 300 + 1
    rec_bufs = [torch.zeros((G * B, n_rec, sdist.REC_COLS), device=dev) for _ in range(2)] if use_dist else None
    gather_done = [None, None]
    gather_spans = []                         # (start, end) events of every all_gather on the gather stream (diagnostics)
    st = {'k': 0}
    pending = {}                              # flow '3d': slot -> handles of the batch still in flight on that slot

    def compute_streams():
        cur = torch.cuda.current_stream()
        return [cur] + [x for x in streams if x is not None and x != cur]

    def flush(b):
        for cs in compute_streams():
            gather_stream.wait_stream(cs)
        with torch.cuda.stream(gather_stream):          # xGMI gather overlaps the following pairs' forwards
            g0 = torch.cuda.Event(enable_timing=True)
            g0.record(gather_stream)
            sdist.gather_detections(rec_bufs[b])
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(gather_stream)
            gather_done[b] = ev
            gather_spans.append((g0, ev))

    def gather_rows(gather):
        """(buffer, first row) for this step's B records, or None; waits for the buffer's previous gather"""
        if not (use_dist and gather):
            return None
        k = st['k']
        b, row = (k // G) % 2, (k % G) * B
        if row == 0 and gather_done[b] is not None:
            for cs in compute_streams():
                cs.wait_event(gather_done[b])
        return b, row

    def gathered(gr):
        if gr is not None:
            st['k'] += 1
            if st['k'] % G == 0:
                flush(gr[0])

    def step(slot=0, gather=True):
        """one pass of the hot path over one batch.  gather=False: no collective (the rank-0-only roofline pass must not enter
        an all_gather alone)"""
        gr = gather_rows(gather)
        if wl['flow'] == '3d':
            # batch k's forward + 3-D stage are enqueued before batch k-S's host phases and results are collected: the
            # Newton-CG solves on the host overlap the next forward on the GPU
            old = pending.pop(slot, None)
            if old is not None:
                pipeline.collect_3d_batch(old)
            hs = pipeline.launch_3d_batch(model, im_l, im_r, im_info, [calib] * B, [(src_h, src_w, 3)] * B, slot=slot, solver='host')
            pending[slot] = hs
            if gr is not None:
                for b in range(B):
                    rec_bufs[gr[0]][gr[1] + b].copy_(hs[b].rec, non_blocking=True)
            gathered(gr)
            return
        out = model(im_l, im_r, im_info, slot=slot, alias_outputs=True)     # views of the slot's result buffers: consumed right below, in stream order
        for b in range(B):
            o = pipeline.image_outputs(out, b) if B > 1 else out
            det = hpost.decode_detections(o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], im_info[b:b + 1])
            keep_idx, num = hpost.class_nms_device(det, 1, 0.05)
            if gr is not None:
                sdist.pack_records_device(det, keep_idx, num, 1, out=rec_bufs[gr[0]][gr[1] + b])
        gathered(gr)

    def drain():
        """flow '3d': collect every batch still in flight (inside the timed region)"""
        for slot in sorted(pending):
            pipeline.collect_3d_batch(pending.pop(slot))

    def finish_gathers():
        """gather the partially filled buffer of the last steps"""
        if use_dist and st['k'] % G:
            flush((st['k'] // G) % 2)
            st['k'] += G - st['k'] % G

    with torch.no_grad():
        def run_steps(n):
            """n passes of the hot path, one batch each, round-robin over the in-flight slots/streams"""
            for k in range(n):
                if S == 1:
                    step(0)
                else:
                    with torch.cuda.stream(streams[k % S]):
                        step(k % S)
            drain()

        for slot in range(S):                       # first touch: autotune + graph capture per slot, serially
            if S == 1:
                step(0)
            else:
                with torch.cuda.stream(streams[slot]):
                    step(slot)
            drain()
            torch.cuda.synchronize()
        run_steps(max(args.warmup, 1))
        finish_gathers()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        finish_gathers()
        host_enqueue_ms = (time.perf_counter() - t0) * 1e3 / args.steps   # host time to launch one step (not a GPU time)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        # per-rank diagnostics for the first multi-GPU run: every rank's own ms per step, and the all_gather's time on its side
        # stream (overlapped with the forwards; its share of the step says how close it is to becoming the critical path)
        multi_gpu = None
        if use_dist:
            spans = gather_spans[-max(1, (args.steps + G - 1) // G):]
            gms = sum(a.elapsed_time(b) for a, b in spans) / max(args.steps, 1) if spans else 0.0
            mine = torch.tensor([elapsed / args.steps * 1e3, gms, host_enqueue_ms], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            multi_gpu = {'per_rank_ms_per_step': [round(float(t[0]), 3) for t in allr],
                         'per_rank_gather_ms_per_step': [round(float(t[1]), 4) for t in allr],
                         # N Python processes share the host: a rank whose enqueue time approaches its step time is host-bound
                         'per_rank_host_enqueue_ms_per_step': [round(float(t[2]), 3) for t in allr],
                         'GPU_MAX_HW_QUEUES_per_rank': os.environ.get('GPU_MAX_HW_QUEUES'),
                         'gather_share_of_step': round(max(float(t[1]) for t in allr) / max(float(t[0]) for t in allr), 4),
                         'note': 'each rank times its own K steps (barrier before and after); value uses the max; the all_gather of the '
                                 'detection records runs on a side stream, one per %d steps, overlapped with the following forwards' % G}
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el[0])

        # ---- `sustained`: the same headline loop once the chip is warm -- a 2 s soak, then >= 300 steps (the 20-step default region
        #      is 0.13 s: boost clocks; profiles/bench_r04_f16x3_60steps.json put the warm figure ~3 % lower)
        sustained = None
        if not args.no_sustained and world == 1 and wl['flow'] != '3d':
            ts = time.perf_counter()
            soak_steps = 0
            while time.perf_counter() - ts < 2.0:
                run_steps(4 * S)
                soak_steps += 4 * S
                torch.cuda.synchronize()
            ns = max(300, args.steps)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run_steps(ns)
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts
            sustained = {'value': round(ns * B / dts, 3), 'unit': 'stereo pairs/s', 'steps': ns, 'ms_per_step': round(dts / ns * 1e3, 3),
                         'soak': '%d steps over 2 s before the timed %d' % (soak_steps, ns),
                         'note': 'the headline loop (same regime, same plans) after a warm soak; `value` is the driver\'s K steps'}

        def serial_step():
            step(0, gather=False)
            drain()

        # host cost of enqueueing ONE step on an idle GPU (queues empty: no back-pressure from the runtime in the number)
        enq = []
        for _ in range(5):
            torch.cuda.synchronize()
            te = time.perf_counter()
            step(0, gather=False)
            enq.append((time.perf_counter() - te) * 1e3)
            drain()
        torch.cuda.synchronize()
        host_enqueue_idle_ms = sorted(enq)[len(enq) // 2]

        # the same K steps strictly one batch at a time (reported next to the headline when S > 1)
        single = None
        insitu = None                                  # the in-situ (latency) plan set, when the headline ran on the shipped one
        if S > 1:
            engine.set_tune_mode('isolated')       # one batch at a time runs on the plans tuned for that (the first step re-tunes / re-records)
            sstreams.set_pairs_in_flight(1)        # ... with the independent branches on side streams (latency mode)
            headline_plans = dict(engine._TUNED)
            if shipped:                            # ... which are the in-situ tuner's own picks, not the throughput-tuned file
                engine._TUNED.clear()
                engine.PLAN_EPOCH += 1
            serial_step()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(0)
                drain()
            finish_gathers()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            e1 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(e1, op=dist.ReduceOp.MAX)
            single = {'value': round(args.steps * B * world / float(e1[0]), 3), 'unit': 'stereo pairs/s',
                      'ms_per_step': round(float(e1[0]) / args.steps * 1e3, 3), 'plans': 'in-situ tuner, every launch timed alone on the chip'}
            engine.set_tune_mode(tune_mode, S)     # back to the headline's plan set (already tuned: nothing is timed again)
            sstreams.set_pairs_in_flight(S)
            if shipped:
                insitu = dict(engine._TUNED)
                engine._TUNED.clear()
                engine._TUNED.update(insitu)       # shapes only the serial leg met keep their in-situ plan
                engine._TUNED.update(headline_plans)
                engine.PLAN_EPOCH += 1

        # ---- the product's default flow of the same step: keypoint branch after class NMS, on the kept detections only
        lazy_fig = None
        if lazy_default and args.precision == 'f16x3' and not use_graph:
            def lazy_step(slot):
                if wl['flow'] == '3d':
                    return step(slot, gather=False)
                out = model(im_l, im_r, im_info, slot=slot, kpts=False, alias_outputs=True)
                plan = model._get_plan(B, int(im_l.shape[2]), int(im_l.shape[3]), slot)
                for b in range(B):
                    o = pipeline.image_outputs(out, b) if B > 1 else out
                    det = hpost.decode_detections(o[0], o[1], o[2], o[3], o[4], None, None, None, im_info[b:b + 1])
                    keep_idx, num = hpost.class_nms_device(det, 1, 0.05)
                    plan.kpts_for_kept(o[0][0].contiguous(), keep_idx, num, im_info[b:b + 1].contiguous(), det['kpts'], args.precision)
            pipeline.LAZY_KPTS = True
            nl = max(6, min(args.steps, 60))
            def run_lazy(n):
                for k in range(n):
                    if S == 1:
                        lazy_step(0)
                    else:
                        with torch.cuda.stream(streams[k % S]):
                            lazy_step(k % S)
                drain()
            run_lazy(2 * S)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            t5 = time.perf_counter()
            run_lazy(nl)
            torch.cuda.synchronize()
            e5 = torch.tensor([time.perf_counter() - t5], dtype=torch.float64, device=dev)
            if use_dist:
                dist.barrier()
                dist.all_reduce(e5, op=dist.ReduceOp.MAX)
            pipeline.LAZY_KPTS = False
            lazy_fig = {'value': round(nl * B * world / float(e5[0]), 3), 'unit': 'stereo pairs/s', 'steps': nl,
                        'ms_per_step': round(float(e5[0]) / nl * 1e3, 3),
                        'note': 'the same step with the keypoint branch run AFTER class NMS on the detections that survive it only '
                                '(stereo_rcnn_amd.pipeline default; device-side keep count bounds the launches, no host read-back): '
                                'the same detections (keypoint values within the engine\'s plan-to-plan rounding, 1e-5): every roi\'s keypoints are independent of the other rois and the '
                                'reference\'s scripts read no other row (demo.py:196-257); NOT `value`, which computes the branch '
                                'for all 300 rois as the reference\'s forward does'}

        # ---- roofline of the dominant kernel (the conv engine).  Two executions, both reported, each next to ITS OWN step time:
        #  * kernel level (`roofline.achieved`): HIP events recorded by the library on the launch stream around every conv
        #    launch, eager launches on ONE stream with the side-stream branches folded in, so that each launch is timed alone
        #    on the chip -- this is the `one_pair_at_a_time` execution, and conv_ms_per_step <= one_pair_at_a_time.ms_per_step;
        #  * headline mode (`roofline.headline`): the same algorithmic conv FLOPs per step over the headline's measured wall
        #    time per step (several pairs in flight share the chip, so per-launch events would time the sharing, not the kernel).
        #  `roofline.layers`: the per-layer table (every launch against ITS OWN bound), 15 worst groups; --layers-out = all.
        roofline = None
        engines = None
        if rank == 0:
            L = _lib.lib()
            model.use_graph = False
            use_program = model.use_program
            model.use_program = False               # the library's per-launch events are taken on eagerly issued launches
            for pl in model._plans.values():        # one stream: every conv launch is timed alone on the chip
                pl.overlap = False
            # ... on the plans tuned for a launch that IS alone on the chip (the in-situ set of the one_pair_at_a_time leg, also what
            # the committed rocprofv3 summary of `--streams 1 --plans` shows): the shipped set trades latency for joules -- fat tiles on
            # the small-M layers that lose 15 % alone and win the several-in-flight step -- and is judged by `roofline.headline`
            plans_of_headline = None
            if shipped and insitu:
                plans_of_headline = dict(engine._TUNED)
                engine._TUNED.clear()
                engine._TUNED.update(insitu)
                engine.PLAN_EPOCH += 1
            serial_step()
            torch.cuda.synchronize()
            nprof = min(args.steps, 5)
            t2 = time.perf_counter()
            for _ in range(nprof):
                serial_step()
            torch.cuda.synchronize()
            serial_ms = (time.perf_counter() - t2) * 1e3 / nprof        # same execution, events off
            engine.FlopCounter.enabled, engine.FlopCounter.flops, engine.FlopCounter.launches, engine.FlopCounter.bytes = True, 0.0, 0, 0.0
            L.srcnn_prof_enable(1)
            for _ in range(nprof):
                serial_step()
            torch.cuda.synchronize()
            ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            L.srcnn_prof_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt))
            L.srcnn_prof_enable(0)
            engine.FlopCounter.enabled = False
            alg = engine.FlopCounter.flops
            achieved = alg / (ms.value * 1e-3) / 1e12
            peak = PEAKS[args.precision]
            issued = 3.0 if args.precision == 'f16x3' else 1.0     # MFMA flops issued per algorithmic flop
            launches = int(cnt.value // nprof)
            alg_step = alg / nprof
            alg_bytes_launch = engine.FlopCounter.bytes / max(engine.FlopCounter.launches, 1)   # compulsory bytes per conv launch
            # HBM-side bytes per conv launch: PMC counters cannot be read from inside this process, so they come from the
            # newest committed rocprofv3 --pmc summary -- but ONLY if that file was measured on these kernel sources
            # (it carries the sha256 of stereo_rcnn_amd/csrc/*) and for this workload; otherwise traffic is null, never stale.
            traffic, traffic_note = None, 'no profiles/pmc_*_traffic.json next to bench.py'
            import glob
            tj = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_*_traffic.json')))
            live = None
            per_kernel = {}
            if args.config == 1 and world == 1 and not args.no_pmc and args.precision == 'f16x3':
                import tempfile
                pf = args.plans if plans_loaded else os.path.join(tempfile.gettempdir(), 'srcnn_bench_plans_%d.json' % os.getpid())
                if not plans_loaded:
                    engine.save_plans(pf)
                torch.cuda.synchronize()
                live = measure_traffic_live(pf, per_kernel=per_kernel)
            if live is not None and live[2] == launches:
                traffic = round((2.0 * live[0] + live[1]) * 1024.0 / max(launches, 1))
                traffic_note = ('bytes per conv launch, HBM side, MEASURED IN THIS RUN on these sources: (2 x FETCH_SIZE + WRITE_SIZE) of '
                                'the conv-engine launches of one step / launches, two child runs of this script under rocprofv3 '
                                '--kernel-trace --pmc (one counter per pass; FETCH_SIZE doubled as the gfx950 note prescribes -- '
                                'uncorrected it is %.1f MB); compulsory bytes of the same launches (every operand element read once, '
                                'every result written once, 4 B each): `algorithmic_bytes_per_launch`'
                                % ((live[0] + live[1]) * 1024.0 / max(launches, 1) / 1e6))
            elif args.config != 1:
                traffic_note = 'PMC traffic is collected for the headline workload (--config 1) only'
            elif tj and args.precision == 'f16x3':
                with open(tj[-1]) as f:
                    txt = f.read().strip()
                pm = json.loads(txt) if txt else {}
                if pm.get('csrc_sha256') != csrc_hash():
                    traffic_note = ('%s was not measured on the current kernel sources (csrc hash differs or is absent): '
                                    'traffic withheld' % os.path.basename(tj[-1]))
                else:
                    traffic = round((2.0 * pm['conv_fetch_size_kb_per_step'] + pm['conv_write_size_kb_per_step']) * 1024.0
                                    / max(launches, 1))
                    traffic_note = ('bytes per conv launch, HBM side: (2 x FETCH_SIZE + WRITE_SIZE) of the conv-engine launches of '
                                    'one step / launches, separate rocprofv3 --pmc passes on these sources (%s; FETCH_SIZE doubled '
                                    'as the gfx950 note prescribes -- uncorrected it is %.1f MB); compulsory bytes of the same launches '
                                    '(every operand element read once, every result written once, 4 B each): `algorithmic_bytes_per_launch`'
                                    % (os.path.basename(tj[-1]),
                                       (pm['conv_fetch_size_kb_per_step'] + pm['conv_write_size_kb_per_step']) * 1024.0 / max(launches, 1) / 1e6))
            head_ms = elapsed / args.steps * 1e3
            rows = layer_table.measure(serial_step, reps=3, precision=args.precision)
            rows_headline_plans = rows
            if plans_of_headline is not None:       # back to the headline's plan set; its launches alone on the chip (for the in-mix table)
                engine._TUNED.clear()
                engine._TUNED.update(plans_of_headline)
                engine.PLAN_EPOCH += 1
                rows_headline_plans = layer_table.measure(serial_step, reps=3, precision=args.precision)
            if args.layers_out:
                with open(args.layers_out, 'w') as f:
                    f.write(layer_table.format_table(rows, 'bench.py --config %d, conv engine %s, MI355X' % (args.config, args.precision)) + '\n')
            roofline = {'bound': 'mfma', 'kernel': ENGINE_DESC[args.precision],
                        'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': round(achieved / peak, 4), 'traffic': traffic,
                        'traffic_note': traffic_note,
                        'algorithmic_bytes_per_launch': round(alg_bytes_launch),
                        'traffic_over_algorithmic': (round(traffic / alg_bytes_launch, 3) if traffic else None),
                        'issued_mfma_frac': round(achieved * issued / peak, 4),
                        'launches_per_step': launches,
                        'avg_launch_ms': round(ms.value / max(cnt.value, 1), 5),
                        'algorithmic_gflop_per_step': round(alg_step / 1e9, 1),
                        'conv_ms_per_step': round(ms.value / nprof, 3),
                        'execution': 'one batch at a time, one stream, every conv launch alone on the chip'
                                     + (', on the in-situ (latency) plans of one_pair_at_a_time; the headline runs on the shipped '
                                        'throughput plans, whose launches alone on the chip take %.3f ms per step'
                                        % (sum(r['us'] for r in rows_headline_plans) / 1e3) if plans_of_headline is not None else ''),
                        'step_ms_of_this_execution': round(serial_ms, 3),
                        'headline': {'execution': '%d batches of %d pairs in flight (the mode `value` is measured in)' % (S, B),
                                     'achieved': round(alg_step / (head_ms * 1e-3) / 1e12, 2),
                                     'frac': round(alg_step / (head_ms * 1e-3) / 1e12 / peak, 4),
                                     'issued_mfma_frac': round(alg_step / (head_ms * 1e-3) / 1e12 * issued / peak, 4),
                                     'step_ms_of_this_execution': round(head_ms, 3)},
                        'sustained_peak': {'value': layer_table.SUSTAINED_MFMA_PEAK[args.precision] / 1e12, 'unit': 'TFLOP/s issued',
                                           'frac_of_it': round(achieved * issued * 1e12 / layer_table.SUSTAINED_MFMA_PEAK[args.precision], 4),
                                           'headline_frac_of_it': round(alg_step / (head_ms * 1e-3) * issued / layer_table.SUSTAINED_MFMA_PEAK[args.precision], 4),
                                           'source': layer_table.SUSTAINED_SOURCE if args.precision == 'f16x3' else 'fp32 MFMA: the spec peak is sustained (MI355X_MICROARCH.md)',
                                           'note': '`frac` above stays against the 2.5 PF dense peak; this is what the same instruction stream sustains on this power-limited chip'},
                        'non_conv': (non_conv_table(per_kernel, non_conv_algorithmic_mb(model._get_plan(B, int(im_l.shape[2]), int(im_l.shape[3]), 0))) if (live is not None and per_kernel) else None),
                        'backbone': layer_table.backbone(rows, args.precision),
                        'layers_summary': layer_table.summary(rows),
                        'layers_note': 'per conv launch alone on the chip: own bound = max(MFMA flops issued / dense MFMA peak, compulsory '
                                       'bytes / 6.3 TB/s achievable HBM); groups pool launches of one layer shape; sorted by time lost',
                        'layers': layer_table.top_for_json(rows, 15)}
            assert ms.value / nprof <= serial_ms * 1.02, "conv time exceeds the step time of its own execution"
            for pl in model._plans.values():
                pl.overlap = None                     # back to the regime's own choice
            # ---- the exact-fp32 engine as a first-class figure of the same line (short: fewer steps, its own plans)
            engines = {args.precision: {'value': round(args.steps * B * world / elapsed, 3), 'ms_per_step': round(head_ms, 3),
                                        'pairs_in_flight': S * B}}
            other = 'f32' if args.precision == 'f16x3' else 'f16x3'
            if not args.no_f32_leg and world == 1 and args.config == 1:
                model.precision = other
                nf = max(3, min(args.steps, 12))
                def run_other(k):
                    if S == 1:
                        step(0, gather=False)
                    else:
                        with torch.cuda.stream(streams[k % S]):
                            step(k % S, gather=False)
                for slot in range(S):                   # first touch per slot: autotunes this engine's plans
                    run_other(slot)
                    torch.cuda.synchronize()
                t3 = time.perf_counter()
                for k in range(nf):
                    run_other(k)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t3
                engines[other] = {'value': round(nf / dt, 3), 'ms_per_step': round(dt / nf * 1e3, 3), 'pairs_in_flight': S,
                                  'steps': nf, 'peak': PEAKS[other],
                                  'frac_headline_mode': round(alg_step / (dt / nf) / 1e12 / PEAKS[other], 4)}
                model.precision = args.precision
            model.use_graph = use_graph
            model.use_program = use_program
            # ---- per-layer evidence of the regime `value` is measured in (rocprofv3 serialises the queues; HIP events around a
            #      launch would time the sharing): marginal in-mix cost per conv group and workgroup residency from the stamps
            if (not args.no_mix_layers and S > 1 and world == 1 and args.config == 1 and args.precision == 'f16x3' and model.use_program
                    and roofline is not None):
                from stereo_rcnn_amd import mix_table

                class _MixRunner(object):
                    def measure(self, steps, repeats=3):
                        run_steps(S)                    # re-records the launch programs after a plan-epoch bump
                        torch.cuda.synchronize()
                        ts = []
                        for _ in range(repeats):
                            tm = time.perf_counter()
                            run_steps(steps)
                            torch.cuda.synchronize()
                            ts.append((time.perf_counter() - tm) * 1e3 / steps)
                        return sorted(ts)[len(ts) // 2]
                try:
                    mr_ = _MixRunner()
                    mbase, marg = mix_table.marginal(mr_, rows_headline_plans, steps=24)
                    sms, resid = mix_table.residency(mr_, S, steps=24)
                    roofline['headline']['layers'] = mix_table.for_json(mbase, marg, resid)
                    roofline['headline']['layers']['step_ms_with_stamps'] = round(sms, 3)
                    if args.mix_out:
                        with open(args.mix_out, 'w') as f:
                            f.write(mix_table.format_marginal(mbase, marg, S) + '\n\n' + mix_table.format_residency(sms, resid, S, base_ms=mbase) + '\n')
                except Exception as e:                  # a measurement aid must never cost the benchmark line
                    roofline['headline']['layers'] = {'error': repr(e)[:300]}
                finally:
                    engine.REPEAT = []
                    engine.PLAN_EPOCH += 1

    # ---- the whole 3-D flow of the same pair (the metric's "3D box" half; BASELINE configs[2] minus the batch): short,
    #      outside the timed region, reported beside the headline -- never as `value`.  EVERY rank runs it (barrier + max over
    #      ranks, aggregate pairs/s), so that on an 8-GPU node the host-solver side of the flow is measured under the load of
    #      all ranks: solver threads sized by LOCAL_WORLD_SIZE and pinned to the GPU's NUMA node (distributed.py).
    full3d = None
    if args.config == 1 and not args.no_3d_leg:
        if use_dist:
            dist.barrier()
        frame = (im_l, im_r, im_info, calib, (args.height, args.width, 3), float(im_info[0, 2]))
        nfr = max(8, min(8 * args.steps, 160))          # (40-48 frames were 10 % pipeline fill and drain: profiles/flow3d_breakdown_r06.txt)
        pipeline.HOST_SOLVER_THREADS = sdist.host_solver_threads()
        full3d = {'pairs_in_flight': 4, 'frames_per_rank': nfr, 'ranks': world, 'host_solver_threads_per_rank': pipeline.HOST_SOLVER_THREADS,
                  'numa_pinning': numa}
        for solver in ('host', 'device', 'host+keypoints_on_kept_only', 'device+keypoints_on_kept_only'):
            pipeline.LAZY_KPTS = solver.endswith('kept_only') and lazy_default
            key = solver
            solver = solver.split('+')[0]
            list(pipeline.detect_3d_stream(model, [frame] * 8, slots=4, solver=solver))
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            t4 = time.perf_counter()
            outs = list(pipeline.detect_3d_stream(model, [frame] * nfr, slots=4, solver=solver))
            torch.cuda.synchronize()
            e4 = torch.tensor([time.perf_counter() - t4], dtype=torch.float64, device=dev)
            if use_dist:
                dist.barrier()
                dist.all_reduce(e4, op=dist.ReduceOp.MAX)
            dt = float(e4[0])
            full3d[key] = {'value': round(nfr * world / dt, 3), 'ms_per_pair': round(dt / nfr * 1e3, 3), 'objects_per_pair': len(outs[0]),
                              'aligned_per_pair': int(sum(o['aligned'] for o in outs[0]))}
        pipeline.LAZY_KPTS = False
        full3d['note'] = ("forward + decode + class NMS + borders + 4-DoF Newton-CG + dense alignment + 3-DoF Newton-CG per pair, whole job "
                          "over all ranks (max over ranks of the elapsed time); 'host' = solves in C on the host between the device stages "
                          "(bit-identical to the reference's scipy path), 'device' = solves as kernels; '+keypoints_on_kept_only' = the pipeline's "
                          "default: keypoint branch after class NMS on the surviving detections (same objects, keypoint values within 1e-5)")

    if rank == 0:
        pairs = args.steps * B * world
        nh, nw = int(im_l.shape[2]), int(im_l.shape[3])
        res = {
            'metric': 'stereo pairs/sec @1242x375 ResNet-101' if args.config != 4 else 'stereo pairs/sec @2484x750 ResNet-50',
            'value': round(pairs / elapsed, 3),
            'unit': 'stereo pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'f32' else 'f32 result via 3xf16 split MFMA (f32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': wl['text'] % {'w': src_w, 'h': src_h, 'nw': nw, 'nh': nh},
                       'baseline_config_index': args.config, 'pairs_per_step': B,
                       'weights': 'seeded random init, reference state_dict schema', 'hipgraph': use_graph,
                       'native_launch_program': bool(model.use_program),
                       'host_enqueue_ms_per_step': round(host_enqueue_ms, 3),
                       'host_enqueue_ms_per_step_idle_gpu': round(host_enqueue_idle_ms, 3), 'plans_preloaded': plans_loaded,
                       'conv_engine': args.precision, 'pairs_in_flight': S * B, 'batches_in_flight': S,
                       'stream_placement': {'main_streams': sstreams.MAIN_KIND, 'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
                                            'branch_side_streams': sstreams.SIDE_KIND,
                                            'note': 'each forward in flight on a HIP stream with a hardware queue of its own; a plan forks its '
                                                    'independent branches onto side streams only while ONE forward is in flight '
                                                    '(stereo_rcnn_amd/streams.py, profiles/queue_mapping_r04.txt)'},
                       'shipped_plans': ('%d conv plans from stereo_rcnn_amd/plans/mi355x.json: tuned on an MI355X with the measured %d-in-flight step '
                                         'as objective (stereo_rcnn_amd/tune.py); one_pair_at_a_time runs on the in-situ tuner\'s own picks' % (shipped, S)) if shipped else None,
                       'tuner_objective': ('concurrent: every conv plan timed with %d copies of the launch in flight on %d HIP streams (the regime '
                                           '`value` is measured in)' % (S, S)) if tune_mode == 'concurrent' and S > 1 else 'isolated: every conv plan timed alone on the chip',
                       'input': 'the same synthetic pair(s) every step (resident in HBM; the forward has no data-dependent control flow on the host)',
                       'one_pair_at_a_time': single, 'engines': engines,
                       'keypoints_on_kept_detections_only': lazy_fig,
                       'full_3d_flow': full3d,
                       'multi_gpu': multi_gpu,
                       'parallelism': ('pairs sharded %d/GPU per step, one RCCL all_gather of the detection records per %d steps' % (B, G)) if use_dist else 'single GPU'},
            'roofline': roofline,
        }
        if args.config != 1:
            res['value_ms_note'] = 'ms_per_step is per batch of %d pairs; value = pairs/s' % B
        if sustained is not None:
            res['sustained'] = sustained
        if not args.no_parity and world == 1 and args.config == 1:
            import bench_parity
            res['parity'] = bench_parity.parity_block(dev, args.precision)
        if not args.no_cpu_baseline and world == 1:          # the CPU path timed beside it: rank 0 at N = 1 only
            res['cpu_baseline'] = cpu_baseline(args.config, src_h, src_w)
        print(json.dumps(res), flush=True)
    if args.plans and not plans_loaded and rank == 0:
        engine.save_plans(args.plans)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
