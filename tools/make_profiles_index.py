"""Writes profiles/INDEX.json: for every file of profiles/ its round, family, whether it is CURRENT (no same-family file of a later round),
what superseded it, the command that produces the family, and -- for files of the current round -- the sha256 of the kernel sources they
were measured on.  (hygiene item 13.)     python tools/make_profiles_index.py [current round tag, default r06]"""
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')
CUR = sys.argv[1] if len(sys.argv) > 1 else 'r06'
ORDER = ['r01', 'r02', 'r03', 'r04', 'r05', 'r05b', 'r06']

COMMANDS = [      # (family regex, how the family is produced)
    (r'^bench_f16x3\.json$', 'python bench.py   (the driver\'s command; tools/refresh_profiles_<round>.sh)'),
    (r'^bench_config(\d)\.json$', 'python bench.py --config N [--no-pmc --steps K]   (refresh script)'),
    (r'^bench_', 'python bench.py with the flags in the name (refresh script of that round)'),
    (r'^layers_config\d\.txt$', 'python bench.py --config N --layers-out <file>: every conv launch alone on the chip against its own bound'),
    (r'^mix_layers', 'python bench.py --mix-out <file> / tools/mix_layers.py: marginal in-mix cost per conv group + workgroup residency from stamps'),
    (r'f16x3_bench_kernel_stats\.csv$', 'rocprofv3 --kernel-trace --stats -- python bench.py --no-pmc --steps 10 --warmup 3 --streams 1 --plans <tuned> (one pair at a time, steady state)'),
    (r'f16x3_bench_conv_avg\.txt$', 'tools/stats_avg.py on the kernel_stats.csv of the same call + that run\'s roofline.avg_launch_ms'),
    (r'^timeline_', 'tools/trace_analyze.py on the same rocprofv3 kernel trace: one steady-state step, kernel by kernel'),
    (r'^stage_times_', 'python tools/time_forward.py f16x3'),
    (r'3d_stage_kernels\.txt$', 'rocprofv3 --kernel-trace --stats of bench.py\'s full_3d_flow leg, 3-D-stage kernels only (refresh script)'),
    (r'^pmc_', 'rocprofv3 --pmc passes (tools/pmc_passes.sh, tools/pmc_sum.py); since round 4 bench.py measures roofline.traffic live'),
    (r'^measured_tolerances', 'pytest tests -m gpu: tests/conftest.py writes the largest deviations the parity tests observed (tests/tolerances.py)'),
    (r'^chain_ab', 'python tools/chain_ab.py --streams S --configs ...   (chained bottleneck launches off / on per layer and tile)'),
    (r'^rpn_group_ab', 'SRCNN_RPN_GROUP={0,small,all} python tools/chain_ab.py --streams 4 / 1 --configs off'),
    (r'^skip_probe', 'python tools/skip_probe.py'),
    (r'^energy_per_kernel', 'python tools/energy_probe.py'),
    (r'^tune_', 'tools/tune_headline.py / tools/tune_from_shipped.py'),
    (r'^conv_microbench', 'SWEEP=1 python tools/conv_bench.py f16s | f32'),
]


def split(name):
    m = re.search(r'(?:^|_)(r0\d b?)(?:_|\.|$)'.replace(' ', ''), name)
    if not m:
        return None, name
    tag = m.group(1)
    fam = re.sub(r'(^|_)' + tag + r'(_|\.|$)', lambda q: ('_' if q.group(1) and q.group(2) == '_' else (q.group(2) if q.group(2) == '.' else '')), name, count=1)
    return tag, fam.lstrip('_')


def csrc_hash():
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'stereo_rcnn_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(base, '*.hip')) + glob.glob(os.path.join(base, '*.h'))):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


files = sorted(f for f in os.listdir(P) if os.path.isfile(os.path.join(P, f)) and f not in ('INDEX.json', 'README.md'))
fams = {}
for f in files:
    tag, fam = split(f)
    fams.setdefault(fam, []).append((ORDER.index(tag) if tag in ORDER else -1, tag, f))
sha = csrc_hash()
out = []
for fam, lst in sorted(fams.items()):
    lst.sort()
    for i, (o, tag, f) in enumerate(lst):
        newer = lst[i + 1][2] if i + 1 < len(lst) else None
        cmd = next((c for rx, c in COMMANDS if re.search(rx, fam)), 'experiment log / notes of that round (see docs/HISTORY.md or DESIGN.md where it is cited)')
        e = {'file': f, 'round': tag, 'family': fam, 'current': newer is None, 'superseded_by': newer, 'command': cmd}
        if tag == CUR:
            e['kernel_sources_sha256'] = sha
        out.append(e)
json.dump({'current_round': CUR, 'kernel_sources_sha256_of_current_round': sha,
           'note': 'current = no file of the same family from a later round; files of earlier rounds are kept as the evidence DESIGN.md / docs/HISTORY.md cite',
           'files': out}, open(os.path.join(P, 'INDEX.json'), 'w'), indent=1)
print('%d files, %d families, %d current' % (len(out), len(fams), sum(1 for e in out if e['current'])))
