"""Per-layer table of the regime the headline is measured in: several batch-1 forwards in flight.

`rocprofv3 --kernel-trace` serialises the hardware queues, and HIP events around a launch time the sharing,
not the kernel -- so the in-flight mix had no per-layer evidence.  Two measurements that do work there:

1. MARGINAL STEP TIME (`marginal`).  A layer group's conv launches are issued n more times right behind themselves (engine.REPEAT;
   outputs rewritten, nothing else changes) and the several-in-flight step is measured again: (t_n - t_0) / n is what that group
   costs the MIX -- CU-time, power, queue slots, everything -- per forward.  Groups whose marginal cost is far below their time
   alone on the chip are absorbed by the other forwards (small-M layers that leave CUs free); groups at or above it own the step.
   The marginals of all conv groups plus the non-conv remainder add up to the step time when costs are additive.
2. WORKGROUP RESIDENCY (`residency`).  The conv kernel's own stamps (s_memrealtime, chip-wide 100 MHz; cycle counters per phase;
   csrc/conv_f16s.hip p.stamp) written by every workgroup of every conv launch of the forwards in flight, each launch into a region
   of its own (the library's stamp arena): per layer the workgroup-seconds it holds, the CU share those workgroups occupy, first
   start to last end of a launch inside the mix, the K loop's share of a workgroup's life, and the MFMA time its tiles need at
   the clock the chip runs at -- i.e. how busy the matrix pipe is WHILE the layer's workgroups sit on their CUs.
Product-side measurement helper (bench.py, tools/mix_layers.py); no oracle.
"""
import ctypes
import re

import torch

from . import _lib, engine

GROUPS = [          # (label, regex over the conv launch names of plan.py)
    ('stem', r'stem$'),
    ('layer1', r'layer1\.'),
    ('layer2', r'layer2\.'),
    ('layer3.conv1', r'layer3\.\d+\.conv1$'),
    ('layer3.conv2', r'layer3\.\d+\.conv2$'),
    ('layer3.conv3', r'layer3\.\d+\.conv3'),
    ('layer4', r'layer4\.'),
    ('fpn.laterals+top', r'fpn\.(lateral|toplayer)'),
    ('fpn.smooth', r'fpn\.smooth'),
    ('layer3.chain', r'layer3\.\d+\.chain'),          # SRCNN_BOTTLENECK_CHAIN=1: [conv2, conv3, next conv1] per launch
    ('rpn_conv.P2', r'rpn_conv(\+head)?\.P2'),         # (also the five-level grouped launch, SRCNN_RPN_GROUP=all)
    ('rpn_conv.P3-P6', r'rpn_conv(\+head)?\.P[3-6]'),  # (one grouped launch by default: 'rpn_conv+head.P3+4+5+6')
    ('rpn_head', r'rpn_head\.'),
    ('box head', r'box\.'),
    ('kpts.0-10', r'kpts\.\d+$'),
    ('kpts.deconv+class', r'kpts\.(deconv|class)'),
]


def group_of(name):
    for label, rx in GROUPS:
        if re.match(rx, name):
            return label
    return 'other'


def marginal(runner, alone_rows, steps=24, target_us=400.0, log=None):
    """runner: tune.StepRunner-like (measure(steps) -> ms per step of the several-in-flight step; re-records after a plan-epoch
    bump).  alone_rows: layer_table.measure() rows (time of every launch alone on the chip).  Returns (base_ms, rows): per group
    the launches, flops, time alone, extra launches used, and the marginal in-mix time per forward."""
    say = log or (lambda *a: None)
    alone = {}
    for r in alone_rows:
        g = alone.setdefault(group_of(r['name']), {'launches': 0, 'us': 0.0, 'flops': 0.0})
        g['launches'] += 1
        g['us'] += r['us']
        g['flops'] += r['flops']
    out = []
    base0 = runner.measure(steps)
    bases = [base0]
    for label, rx in GROUPS:
        a = alone.get(label)
        if not a:
            continue
        n = int(max(1, min(6, round(target_us / max(a['us'], 1.0)))))      # small groups are repeated more often: signal above the step's noise
        engine.REPEAT = [(re.compile(rx), n)]
        engine.PLAN_EPOCH += 1
        t = runner.measure(steps)
        engine.REPEAT = []
        engine.PLAN_EPOCH += 1
        b = runner.measure(steps)                                           # the incumbent again: clocks / temperature drift
        ref = 0.5 * (bases[-1] + b)
        bases.append(b)
        m_us = (t - ref) * 1e3 / n
        out.append({'group': label, 'launches': a['launches'], 'gflop': a['flops'] / 1e9, 'alone_us': a['us'], 'extra': n,
                    'marginal_us': m_us, 'marginal_over_alone': m_us / a['us'] if a['us'] > 0 else 0.0,
                    'in_mix_tflops': a['flops'] / m_us / 1e6 if m_us > 0 else float('inf')})
        say('  %-20s x%d extra: %.3f ms vs %.3f -> %.1f us per forward in the mix (alone %.1f us)' % (label, n, t, ref, m_us, a['us']))
    base = sum(bases) / len(bases)
    return base, out


def format_marginal(base_ms, rows, S):
    tot = sum(r['marginal_us'] for r in rows)
    lines = ['marginal cost of every conv group INSIDE the %d-in-flight mix (step %.3f ms = %.1f pairs/s): the group\'s launches issued n more times, '
             '(t_n - t_0) / n per forward' % (S, base_ms, 1e3 / base_ms),
             '%-20s %4s %8s %9s %6s %11s %8s %9s %7s' % ('group', 'n', 'GFLOP', 'alone us', 'extra', 'in-mix us', 'mix/alone', 'mix TF/s', 'of step')]
    for r in sorted(rows, key=lambda r: -r['marginal_us']):
        lines.append('%-20s %4d %8.1f %9.1f %6d %11.1f %8.2f %9.1f %6.1f%%'
                     % (r['group'], r['launches'], r['gflop'], r['alone_us'], r['extra'], r['marginal_us'], r['marginal_over_alone'],
                        r['in_mix_tflops'], 100 * r['marginal_us'] / (base_ms * 1e3)))
    lines.append('%-20s %4s %8.1f %9.1f %6s %11.1f %8.2f %9s %6.1f%%   <- all conv groups; the rest of the step (%.1f us) is the non-conv '
                 'launches and whatever is not additive'
                 % ('sum', '', sum(r['gflop'] for r in rows), sum(r['alone_us'] for r in rows), '', tot,
                    tot / max(sum(r['alone_us'] for r in rows), 1e-9), '', 100 * tot / (base_ms * 1e3), base_ms * 1e3 - tot))
    return '\n'.join(lines)


# ------------------------------------------------------------------------------------------------ workgroup residency
CLOCK_GHZ_IN_MIX = 1.9          # profiles/clocks_under_bench_r04.txt: 1.89-1.91 GHz in the headline regime
MFMA_FLOP_PER_CLK_PER_CU = 2.5e15 / 256 / 2.4e9       # dense f16 peak at the 2.4 GHz peak clock -> 4069 flop / clk / CU


def cu_share(lds_bytes, threads):
    """Share of a CU one workgroup occupies (model): the 256x256 tile's 8 waves hold 256 registers each -- the whole CU; otherwise
    the larger of its LDS share (160 KB) and its wave-slot share at 128 registers per wave (16 waves per CU)."""
    if threads == 512 and lds_bytes >= 131072:
        return 1.0
    return min(1.0, max(lds_bytes / 163840.0, threads / 1024.0))


def residency(runner, S, launches_per_forward=None, steps=24, arena_mb=384):
    """Arms the stamp arena, re-records the launch programs (every recorded conv launch gets its own stamp region), runs the
    several-in-flight step, and reads back the LAST execution of every conv launch of every slot.  Returns (ms with stamps, rows)."""
    import numpy as np
    L = _lib.lib()
    L.srcnn_debug_set_stamp_arena.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.srcnn_debug_set_stamp_arena.restype = None
    L.srcnn_debug_stamp_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.srcnn_debug_stamp_log.restype = ctypes.c_int
    words = arena_mb * (1 << 20) // 8
    arena = torch.zeros(words, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    L.srcnn_debug_set_stamp_arena(arena.data_ptr(), words)
    try:
        engine.PLAN_EPOCH += 1
        ms = runner.measure(steps)
        torch.cuda.synchronize()
        cap = 1 << 16
        logbuf = np.zeros((cap, 8), np.int64)
        n = L.srcnn_debug_stamp_log(logbuf.ctypes.data, cap)
        log = logbuf[:min(n, cap)]
    finally:
        L.srcnn_debug_set_stamp_arena(None, 0)
        engine.PLAN_EPOCH += 1
    st = arena.cpu().numpy()
    del arena
    # the log holds, per slot, the eager warm-up pass of Plan._record_program (stamped once, alone-ish) and then the recorded
    # pass (re-stamped by every replay): blocks of equal length alternate; keep the recorded ones
    per = len(log) // (2 * S) if launches_per_forward is None else launches_per_forward
    assert per > 0 and len(log) >= 2 * S * per, (len(log), S, per)
    rows = {}
    for slot in range(S):
        blk = log[(2 * slot + 1) * per:(2 * slot + 2) * per]
        for off, wgs, tag, lds, threads, M, N, K in blk.tolist():
            a = st[off:off + wgs * 16].reshape(wgs, 16)
            a = a[(a[:, 7] & 1) == 1]
            if a.shape[0] == 0:
                continue
            name = engine.TAG_NAMES.get(tag + 1, 'tag %d' % tag)
            g = rows.setdefault(group_of(name), {'launches': 0, 'wgs': 0, 'wg_us': 0.0, 'cu_us': 0.0, 'span_us': 0.0, 'life_clk': 0.0,
                                                 'kloop_clk': 0.0, 'vm_clk': 0.0, 'bar_clk': 0.0, 'mfma_us': 0.0})
            res = (a[:, 9] - a[:, 8]).astype(np.float64) / 100.0                       # us per workgroup (100 MHz clock)
            g['launches'] += 1
            g['wgs'] += int(a.shape[0])
            g['wg_us'] += float(res.sum())
            g['cu_us'] += float(res.sum()) * cu_share(lds, threads)
            g['span_us'] += float(a[:, 9].max() - a[:, 8].min()) / 100.0
            g['life_clk'] += float((a[:, 5] - a[:, 0]).sum())
            g['kloop_clk'] += float((a[:, 3] - a[:, 1]).sum())
            g['vm_clk'] += float(a[:, 10].sum())
            g['bar_clk'] += float(a[:, 11].sum())
            # MFMA time the launch's tiles need on one CU at the clock of the mix: 3 products, tile = lds / (stages * 128) rows
            g['mfma_us'] += 3.0 * 2.0 * M * N * K / (MFMA_FLOP_PER_CLK_PER_CU * CLOCK_GHZ_IN_MIX * 1e9) * 1e6
    out = []
    for label, g in rows.items():
        k = float(S)                                                                   # per forward: averaged over the S slots
        out.append({'group': label, 'launches': g['launches'] / k, 'wgs': g['wgs'] / k, 'wg_us': g['wg_us'] / k, 'cu_us': g['cu_us'] / k,
                    'span_us': g['span_us'] / k, 'kloop_share': g['kloop_clk'] / max(g['life_clk'], 1.0),
                    'vmwait_share': g['vm_clk'] / max(g['life_clk'], 1.0), 'barrier_share': g['bar_clk'] / max(g['life_clk'], 1.0),
                    'mfma_cu_us': g['mfma_us'] / k, 'mfma_busy_while_resident': g['mfma_us'] / max(g['cu_us'], 1e-9)})
    return ms, out


def format_residency(ms, rows, S, base_ms=None):
    avail = 256.0 * ms * 1e3
    tot_cu = sum(r['cu_us'] for r in rows)
    tot_mfma = sum(r['mfma_cu_us'] for r in rows)
    lines = ['workgroup residency of every conv launch INSIDE the %d-in-flight mix, from the kernel\'s own stamps (step with stamps %.3f ms%s); per forward, '
             'mean of the %d slots\' last executions' % (S, ms, '' if base_ms is None else ', %.3f without' % base_ms, S),
             'CU-us = workgroup-us x the CU share a workgroup occupies (256x256 tile: 1; else max(LDS / 160 KB, threads / 1024)); the chip offers '
             '256 CUs x %.3f ms = %.0f CU-us per forward' % (ms, avail),
             '%-20s %5s %7s %10s %10s %8s %9s %7s %7s %7s %10s %9s' % ('group', 'n', 'WGs', 'WG-us', 'CU-us', 'of chip', 'span us', 'K loop', 'vmwait', 'barrier',
                                                                       'MFMA CU-us', 'MFMA busy')]
    for r in sorted(rows, key=lambda r: -r['cu_us']):
        lines.append('%-20s %5.0f %7.0f %10.0f %10.0f %7.1f%% %9.1f %6.0f%% %6.0f%% %6.0f%% %10.0f %8.0f%%'
                     % (r['group'], r['launches'], r['wgs'], r['wg_us'], r['cu_us'], 100 * r['cu_us'] / avail, r['span_us'], 100 * r['kloop_share'],
                        100 * r['vmwait_share'], 100 * r['barrier_share'], r['mfma_cu_us'], 100 * r['mfma_busy_while_resident']))
    lines.append('%-20s %5s %7s %10s %10.0f %7.1f%% %9s %7s %7s %7s %10.0f %8.0f%%   <- conv workgroups hold %.1f %% of the chip\'s CU-time; the matrix pipe '
                 'is busy %.1f %% of ALL CU-time (%.0f %% of the time conv workgroups are resident)'
                 % ('sum', '', '', '', tot_cu, 100 * tot_cu / avail, '', '', '', '', tot_mfma, 100 * tot_mfma / max(tot_cu, 1e-9), 100 * tot_cu / avail,
                    100 * tot_mfma / avail, 100 * tot_mfma / max(tot_cu, 1e-9)))
    return '\n'.join(lines)


def for_json(base_ms, marg, resid=None):
    res = {}
    for r in marg:
        res[r['group']] = {'launches': r['launches'], 'gflop': round(r['gflop'], 1), 'alone_us': round(r['alone_us'], 1),
                           'in_mix_marginal_us': round(r['marginal_us'], 1), 'in_mix_tflops': round(min(r['in_mix_tflops'], 9999.0), 1)}
    for r in resid or []:
        res.setdefault(r['group'], {}).update({'cu_us': round(r['cu_us']), 'span_us': round(r['span_us'], 1),
                                               'kloop_share': round(r['kloop_share'], 3), 'mfma_busy_while_resident': round(r['mfma_busy_while_resident'], 3)})
    return {'step_ms': round(base_ms, 3), 'groups': res,
            'note': 'in_mix_marginal_us: the step-time increase per extra execution of the group\'s launches with the other forwards in flight '
                    '(stereo_rcnn_amd/mix_table.py); cu_us / span / K-loop share / MFMA busy: from the conv kernel\'s own per-workgroup stamps'}
