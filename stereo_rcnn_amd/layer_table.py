"""Per-layer roofline table of the conv engine (measurement helper of bench.py and tools/layer_table.py).

For every conv launch of one forward: shape, algorithmic FLOPs, compulsory bytes (engine.FlopCounter), the measured time of
the launch alone on the chip (HIP events taken by the library on the launch stream, srcnn_prof_read_launches) and the
launch's OWN bound  max(issued MFMA flops / MFMA peak, compulsory bytes / achievable HBM rate)  -- so that layers can be
ranked by the time they lose against what the hardware allows for THEIR shape, instead of against one chip-wide peak.
Peaks: MI355X_MICROARCH.md (dense f16 MFMA 2.5 PFLOP/s; fp32 MFMA 157.3 TFLOP/s; HBM 6.3 TB/s achievable of 8 TB/s).
"""
import ctypes
import re

import torch

from . import _lib, engine

MFMA_PEAK = {'f16x3': 2.5e15, 'f32': 157.3e12}
ISSUED = {'f16x3': 3.0, 'f32': 1.0}          # MFMA flops issued per algorithmic flop
HBM_ACHIEVABLE = 6.3e12


def measure(step, reps=5, precision='f16x3'):
    """`step()` = one forward issued eagerly on ONE stream (caller's responsibility: launch program / side streams off).
    Returns one dict per conv launch, in launch order, with the mean time over `reps` steps."""
    L = _lib.lib()
    step()
    torch.cuda.synchronize()
    engine.FlopCounter.enabled, engine.FlopCounter.rows = True, []
    saved = (engine.FlopCounter.flops, engine.FlopCounter.launches, engine.FlopCounter.bytes)
    L.srcnn_prof_enable(1)
    step()
    torch.cuda.synchronize()
    rows = engine.FlopCounter.rows
    engine.FlopCounter.rows = None
    cap = len(rows) + 8
    buf = (ctypes.c_float * cap)()
    tot = [0.0] * len(rows)
    done = 0
    for r in range(reps):
        n = L.srcnn_prof_read_launches(buf, cap)
        if n < 0:
            _lib.check(n, "srcnn_prof_read_launches")
        assert n == len(rows), "launch list changed between steps (%d vs %d)" % (n, len(rows))
        for i in range(n):
            tot[i] += buf[i]
        done += 1
        if r + 1 < reps:
            L.srcnn_prof_enable(1)          # resets the recording
            step()
            torch.cuda.synchronize()
    L.srcnn_prof_enable(0)
    engine.FlopCounter.enabled = False
    engine.FlopCounter.flops, engine.FlopCounter.launches, engine.FlopCounter.bytes = saved
    peak, issued = MFMA_PEAK[precision], ISSUED[precision]
    for r, t in zip(rows, tot):
        us = t / done * 1e3
        mr, nr, waves, stages, splits = r.get('plan', (0, 0, 0, 0, 0))
        r['us'] = us
        r['wgs'] = (-(-r['M'] // (64 * mr)) * -(-r['N'] // (64 * nr)) * max(splits, 1)) if mr and nr else 0
        r['mfma_us'] = issued * r['flops'] / peak * 1e6
        r['hbm_us'] = r['bytes'] / HBM_ACHIEVABLE * 1e6
        r['bound_us'] = max(r['mfma_us'], r['hbm_us'])
        r['bound'] = 'mfma' if r['mfma_us'] >= r['hbm_us'] else 'hbm'
        r['frac_of_own_bound'] = r['bound_us'] / us if us > 0 else 0.0
        r['lost_us'] = us - r['bound_us']
        r['tflops'] = r['flops'] / us / 1e6 if us > 0 else 0.0
    return rows


def grouped(rows):
    """Launches of the same layer shape (e.g. the 22 identical layer3 blocks) pooled; sorted by lost time."""
    groups = {}
    for r in rows:
        key = (re.sub(r'layer(\d)\.\d+\.', r'layer\1.*.', r['name']), r['M'], r['N'], r['K'], r.get('plan'))
        g = groups.setdefault(key, {'name': key[0], 'M': r['M'], 'N': r['N'], 'K': r['K'], 'plan': r.get('plan'), 'wgs': r['wgs'],
                                    'launches': 0, 'us': 0.0, 'bound_us': 0.0, 'flops': 0.0, 'bytes': 0.0, 'bound': r['bound']})
        g['launches'] += 1
        for k in ('us', 'bound_us', 'flops', 'bytes'):
            g[k] += r[k]
    out = []
    for g in groups.values():
        g['lost_us'] = g['us'] - g['bound_us']
        g['frac_of_own_bound'] = g['bound_us'] / g['us'] if g['us'] > 0 else 0.0
        g['tflops'] = g['flops'] / g['us'] / 1e6 if g['us'] > 0 else 0.0
        out.append(g)
    return sorted(out, key=lambda g: -g['lost_us'])


def summary(rows):
    us = sum(r['us'] for r in rows)
    bound = sum(r['bound_us'] for r in rows)
    return {'launches': len(rows), 'conv_us': round(us, 1), 'sum_of_own_bounds_us': round(bound, 1),
            'frac_of_own_bounds': round(bound / us, 4) if us > 0 else None,
            'mfma_bound_launches': sum(1 for r in rows if r['bound'] == 'mfma'),
            'hbm_bound_launches': sum(1 for r in rows if r['bound'] == 'hbm'),
            'time_in_hbm_bound_launches_us': round(sum(r['us'] for r in rows if r['bound'] == 'hbm'), 1)}


# Sustained rate of the engine's own MFMA pattern on random data with the K loop's LDS fragment reads, all 256 CUs, 2.5 s
# (profiles/mfma_sustained_r04.txt, tools/mfma_sustained.py): the chip is power-limited there (~1.29 kW at 1.72 GHz) -- the
# same pattern on zero data runs at 2.39 GHz / 2.2 PF.  Register-resident without the LDS reads: 1681 TF (random), 2459 (zeros).
SUSTAINED_MFMA_PEAK = {'f16x3': 1.504e15, 'f32': 157.3e12}
SUSTAINED_SOURCE = 'profiles/mfma_sustained_r04.txt: 3 x v_mfma_f32_32x32x16_f16 per accumulator + 8 ds_read_b128 per 12 MFMAs, random data, 256 CUs, 1.72 GHz at 1.28 kW'


def backbone(rows, precision='f16x3'):
    """The north star's MFMA target is stated on the backbone: the trunk's conv launches (stem + layer1..4) pooled -- their
    algorithmic GFLOP over their summed time (each launch alone on the chip), as achieved TFLOP/s, as issued-MFMA fraction of
    the 2.5 PF dense peak, and against the sustained peak above."""
    tr = [r for r in rows if r['name'] == 'stem' or r['name'].startswith('layer')]
    if not tr:
        return None
    us = sum(r['us'] for r in tr)
    fl = sum(r['flops'] for r in tr)
    ach = fl / us / 1e6
    return {'launches': len(tr), 'algorithmic_gflop': round(fl / 1e9, 1), 'conv_us': round(us, 1), 'achieved_tflops': round(ach, 1),
            'frac_of_peak_algorithmic': round(ach * 1e12 / MFMA_PEAK[precision], 4),
            'issued_mfma_frac_of_peak': round(ach * 1e12 * ISSUED[precision] / MFMA_PEAK[precision], 4),
            'issued_mfma_frac_of_sustained_peak': round(ach * 1e12 * ISSUED[precision] / SUSTAINED_MFMA_PEAK[precision], 4),
            'note': 'trunk conv launches (stem, layer1-4), each alone on the chip; the max-pool between stem and layer1 is not a conv launch'}


def top_for_json(rows, n=15):
    res = []
    for g in grouped(rows)[:n]:
        res.append({'layer': g['name'], 'launches': g['launches'], 'M': g['M'], 'N': g['N'], 'K': g['K'],
                    'plan': list(g['plan']) if g['plan'] else None, 'workgroups': g['wgs'], 'us': round(g['us'], 1),
                    'own_bound_us': round(g['bound_us'], 1), 'bound': g['bound'], 'frac_of_own_bound': round(g['frac_of_own_bound'], 3),
                    'lost_us': round(g['lost_us'], 1), 'tflops': round(g['tflops'], 1)})
    return res


def format_table(rows, title=''):
    lines = [title] if title else []
    s = summary(rows)
    lines.append('%d conv launches, %.1f us in total; sum of the launches\' own bounds max(3 x flops / 2.5 PF, bytes / 6.3 TB/s) = %.1f us '
                 '-> %.1f %% of own bounds; %d launches MFMA-bound by shape, %d HBM-bound (%.1f us spent in those)'
                 % (s['launches'], s['conv_us'], s['sum_of_own_bounds_us'], 100 * s['frac_of_own_bounds'], s['mfma_bound_launches'],
                    s['hbm_bound_launches'], s['time_in_hbm_bound_launches_us']))
    lines.append('')
    lines.append('grouped by layer shape, sorted by time lost against the own bound')
    hdr = '%-26s %3s %7s %5s %6s %-16s %5s %8s %8s %5s %6s %8s %7s' % ('layer', 'n', 'M', 'N', 'K', 'plan', 'WGs', 'us', 'bound us', 'by', 'frac', 'lost us', 'TF/s')
    lines.append(hdr)
    for g in grouped(rows):
        lines.append('%-26s %3d %7d %5d %6d %-16s %5d %8.1f %8.1f %5s %6.3f %8.1f %7.1f'
                     % (g['name'][:26], g['launches'], g['M'], g['N'], g['K'], str(g['plan']), g['wgs'], g['us'], g['bound_us'], g['bound'],
                        g['frac_of_own_bound'], g['lost_us'], g['tflops']))
    lines.append('')
    lines.append('every launch, in launch order')
    lines.append('%4s %-26s %7s %5s %6s %-16s %5s %8s %8s %8s %8s %5s %6s' % ('#', 'layer', 'M', 'N', 'K', 'plan', 'WGs', 'GFLOP', 'MB', 'us', 'bound us', 'by', 'frac'))
    for i, r in enumerate(rows):
        lines.append('%4d %-26s %7d %5d %6d %-16s %5d %8.2f %8.1f %8.1f %8.1f %5s %6.3f'
                     % (i, r['name'][:26], r['M'], r['N'], r['K'], str(r.get('plan')), r['wgs'], r['flops'] / 1e9, r['bytes'] / 1e6, r['us'],
                        r['bound_us'], r['bound'], r['frac_of_own_bound']))
    return '\n'.join(lines)
