"""dev tool: per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py stereo_rcnn_amd/csrc/conv_f16s.hip [name filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-fvisibility=hidden', '-c', src,
       '-o', '/tmp/_kr.o', '-Rpass-analysis=kernel-resource-usage'] + os.environ.get('EXTRA', '').split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.split('\n'):
    m = re.search(r'remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (.*?) \[-Rpass', l)
    if not m:
        if 'error' in l:
            print(l)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print('%-70s %5s %5s %5s %8s %4s %6s %6s' % ('kernel', 'SGPR', 'VGPR', 'AGPR', 'scratch', 'occ', 'sSpill', 'vSpill'))
for r in rows:
    if flt in r['name']:
        n = re.sub(r'^void srcnn::', '', r['name']).replace('(srcnn::ConvArgs)', '')
        print('%-70s %5s %5s %5s %8s %4s %6s %6s' % (n[:70], r.get('TotalSGPRs'), r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'),
                                                   r.get('Occupancy'), r.get('SGPRs Spill'), r.get('VGPRs Spill')))
