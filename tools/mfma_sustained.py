"""Sustained MFMA rate of the conv engine's instruction pattern with clock / power samples (dev tool).

Runs tools/probes/mfma_sustained.hip (built on first use with hipcc) for a few seconds per case while a thread samples
`rocm-smi --showclocks --showpower` -- register-resident MFMAs only, then the same with the K loop's LDS fragment reads, each on
zero data and on random data (hi halves of order 1, lo halves 2^-11 of that, like the SPLIT16 operands).
    usage: python tools/mfma_sustained.py [seconds per case = 2.5]  > profiles/mfma_sustained_r04.txt
"""
import os
import re
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'probes', 'mfma_sustained.hip')
EXE = os.path.join(HERE, 'probes', '_build', 'mfma_sustained')


def build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', SRC, '-o', EXE])


def sample(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
            pw = re.search(r'Power \(W\): ([0-9.]+)', out)
            if sclk:
                rows.append((int(sclk.group(1)), float(pw.group(1)) if pw else float('nan')))
        except (subprocess.TimeoutExpired, OSError):
            pass
        time.sleep(0.1)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    cases = [(0, 'zero'), (0, 'rand'), (1, 'zero'), (1, 'rand')]
    if len(sys.argv) > 2 and sys.argv[2] == 'order':      # the instruction-order question only: random data, three orders, twice
        cases = [(0, 'rand'), (2, 'rand'), (3, 'rand'), (0, 'rand'), (2, 'rand'), (3, 'rand')]
    build()
    print('Sustained rate of the conv engine\'s MFMA pattern (3 x v_mfma_f32_32x32x16_f16 per accumulator, 4 accumulators per wave, '
          '8 waves per CU), %.1f s per case, rocm-smi sampled meanwhile (sclk MHz / package W: median [min..max] over the samples)' % secs)
    for variant, data in cases:
        if True:
            rows, stop = [], threading.Event()
            th = threading.Thread(target=sample, args=(stop, rows))
            th.start()
            time.sleep(0.3)
            idle = list(rows)
            del rows[:]
            r = subprocess.run([EXE, str(variant), data, str(secs)], capture_output=True, text=True)
            stop.set()
            th.join()
            line = (r.stdout.strip().splitlines() or ['(no output) ' + r.stderr.strip()[-300:]])[-1]
            # drop the first and last sample (ramp / after the end)
            mid = rows[1:-1] if len(rows) > 4 else rows
            if mid:
                cl = sorted(x[0] for x in mid)
                pw = sorted(x[1] for x in mid)
                stat = 'sclk %d [%d..%d] MHz, power %.0f [%.0f..%.0f] W, %d samples' % (cl[len(cl) // 2], cl[0], cl[-1], pw[len(pw) // 2], pw[0], pw[-1], len(mid))
            else:
                stat = 'no rocm-smi samples'
            print(line)
            print('    ' + stat, flush=True)


if __name__ == '__main__':
    main()
