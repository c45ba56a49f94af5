"""Energy per launch, kernel by kernel (round 5): the headline regime sits at the package power limit (1.37 kW of 1.4 kW,
profiles/clocks_under_bench_r04.txt), where throughput is pairs per JOULE.  This probe loops ONE conv shape at a time for ~1.5 s
while `rocm-smi --showpower --showclocks` is sampled, and prints average package power, clock, time and energy per launch,
algorithmic TFLOP/s and picojoules per ISSUED MFMA flop (3 issued per algorithmic one) -- with the idle power (measured first)
shown separately, since it is charged per second whatever runs.  Which layers are energy-inefficient (joules per issued flop
far above the big kernels') is what the power-limited mix pays for.
    python tools/energy_probe.py [seconds per shape = 1.5]  > profiles/energy_per_kernel_r05.txt
"""
import os
import re
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
dev = torch.device('cuda:0')
engine.PRECISION = 'f16x3'


def sampler(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
            pw = re.search(r'Power \(W\): ([0-9.]+)', out)
            if pw:
                rows.append((int(sclk.group(1)) if sclk else 0, float(pw.group(1))))
        except (subprocess.TimeoutExpired, OSError):
            pass
        time.sleep(0.05)


def measure(fn, secs):
    """loop fn() for `secs`; returns (launches, seconds, median W, median MHz)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sampler, args=(stop, rows))
    th.start()
    time.sleep(0.25)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    rows = rows[2:] or rows
    pw = sorted(r[1] for r in rows)
    ck = sorted(r[0] for r in rows)
    return n, dt, (pw[len(pw) // 2] if pw else float('nan')), (ck[len(ck) // 2] if ck else 0)


# (name, B, H, W, cin, cout, k, stride, pad, plan or None = the in-situ tuner's pick, residual?)
SHAPES = [
    ('rpn.conv 3x3 256->512 P2   256x256', 2, 150, 497, 256, 512, 3, 1, 1, (4, 4, 8, 2, 1), False),
    ('kpts 3x3 256 (300x14x14)   256x256', 300, 14, 14, 256, 256, 3, 1, 1, (4, 4, 8, 2, 1), False),
    ('fpn.smooth3 3x3 256 P2     256x256', 2, 150, 497, 256, 256, 3, 1, 1, (4, 4, 8, 2, 1), False),
    ('l3.conv2 3x3 256           128x128', 2, 38, 125, 256, 256, 3, 1, 1, (2, 2, 8, 4, 1), False),
    ('l3.conv2 3x3 256           256x128', 2, 38, 125, 256, 256, 3, 1, 1, (4, 2, 8, 3, 1), False),
    ('l3.conv1 1x1 1024->256     128x128', 2, 38, 125, 1024, 256, 1, 1, 0, (2, 2, 8, 4, 1), False),
    ('l3.conv3 1x1 256->1024+res 128x128', 2, 38, 125, 256, 1024, 1, 1, 0, (2, 2, 8, 2, 1), True),
    ('l3.conv3 1x1 256->1024+res 256x256', 2, 38, 125, 256, 1024, 1, 1, 0, (4, 4, 8, 2, 1), True),
    ('l2.conv2 3x3 128           128x64 ', 2, 75, 249, 128, 128, 3, 1, 1, (2, 1, 4, 2, 1), False),
    ('l2.conv3 1x1 128->512+res  64x64  ', 2, 75, 249, 128, 512, 1, 1, 0, (1, 1, 4, 2, 1), True),
    ('l1.conv2 3x3 64            128x64 ', 2, 150, 497, 64, 64, 3, 1, 1, (2, 1, 4, 2, 1), False),
    ('l1.conv3 1x1 64->256+res   128x128', 2, 150, 497, 64, 256, 1, 1, 0, (2, 2, 8, 2, 1), True),
    ('l1.conv1 1x1 256->64       128x64 ', 2, 150, 497, 256, 64, 1, 1, 0, (2, 1, 4, 2, 1), False),
    ('fpn.lateral3 1x1 256->256  128x128', 2, 150, 497, 256, 256, 1, 1, 0, (2, 2, 4, 2, 1), False),
    ('box.top0 GEMM 300x25088x2048      ', 300, 1, 1, 25088, 2048, 1, 1, 0, (2, 2, 4, 2, 8), False),
]

rows, stop = [], threading.Event()
th = threading.Thread(target=sampler, args=(stop, rows))
th.start()
time.sleep(2.0)
stop.set()
th.join()
idle = sorted(r[1] for r in rows)[len(rows) // 2] if rows else float('nan')
print('MI355X, conv engine f16x3 (SPLIT16 operands), one shape at a time in a loop of %.1f s, rocm-smi sampled at ~10 Hz (median); idle package power before the '
      'first launch: %.0f W' % (SECS, idle))
print('%-38s %8s %8s %7s %7s %9s %10s %12s %14s' % ('shape / tile', 'us', 'TF/s', 'W', 'MHz', 'mJ/launch', 'of it idle', 'pJ/issued fl', 'pJ/fl above idle'))
for name, B, H, W, cin, cout, k, s, p, plan, res in SHAPES:
    x = engine.act_convert(torch.randn(B, H, W, cin, device=dev), 0, 1)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    cw = engine.prep_conv(w, torch.zeros(cout), s, p, True, device=dev)
    OH, OW = engine.conv_out_hw(H, W, k, k, s, p)
    y = torch.empty(B, OH, OW, cout, device=dev)
    r = engine.act_convert(torch.randn(B, OH, OW, cout, device=dev), 0, 1) if res else None
    kw = dict(x_fmt=1, y_fmt=1, plan=plan)
    if res:
        kw.update(residual=r, res_fmt=1)
    fn = lambda: engine.conv2d(cw, x, B, H, W, y, OH, OW, **kw)
    n, dt, watts, mhz = measure(fn, SECS)
    us = dt / n * 1e6
    fl = 2.0 * B * OH * OW * cout * cin * k * k
    mj = watts * dt / n * 1e3
    print('%-38s %8.1f %8.1f %7.0f %7d %9.3f %9.0f%% %12.3f %14.3f'
          % (name, us, fl / us / 1e6, watts, mhz, mj, 100 * idle / watts, watts * (dt / n) / (3 * fl) * 1e12, (watts - idle) * (dt / n) / (3 * fl) * 1e12), flush=True)
