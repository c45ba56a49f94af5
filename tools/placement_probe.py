"""Where do the workgroups of a CU-masked stream run?  (dev tool; srcnn_probe_placement)   usage: python tools/placement_probe.py"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import _lib, streams

dev = torch.device('cuda:0')
L = _lib.lib()
n_cus = torch.cuda.get_device_properties(0).multi_processor_count
print('device CUs:', n_cus)


def probe(stream, blocks=2048):
    xcc = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    hw = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    _lib.check(L.srcnn_probe_placement(blocks, xcc.data_ptr(), hw.data_ptr(), stream.cuda_stream), 'probe')
    stream.synchronize()
    xcc, hw = xcc.cpu().tolist(), hw.cpu().tolist()
    per_xcc = collections.Counter(xcc)
    cus = collections.defaultdict(set)
    for x, h in zip(xcc, hw):
        cus[x].add(((h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15))      # (se_id, sh_id, cu_id)
    mod8 = sum(1 for b, x in enumerate(xcc) if x == b % 8)
    print('   blocks per XCC:', dict(sorted(per_xcc.items())), ' block b on XCC b%%8: %d / %d' % (mod8, blocks))
    for x in sorted(cus):
        print('   XCC %d: %2d distinct (se, sh, cu): %s' % (x, len(cus[x]), sorted(cus[x])))
    return {x: len(c) for x, c in cus.items()}


print('unmasked dedicated stream:')
probe(streams.new_stream('dedicated'))
for parts in (4, 2):
    for k, m in enumerate(streams.partition_masks(parts, n_cus)):
        if k in (0, parts - 1):
            print('partition %d of %d, mask %s:' % (k, parts, ' '.join('%08x' % w for w in m)))
            probe(streams.masked_stream(m))

# which enumeration do the mask bits follow?  partition 0 of 4 with bit 1 cleared: one CU less in XCD 1 if bit b belongs to XCD b % 8
# (interleaved), in XCD 0 if bits 0..31 are XCD 0's CUs (XCD-major).  Safe either way: every XCD keeps 7-8 CUs.
m = streams.partition_masks(4, n_cus)[0]
m[0] &= ~2
print('partition 0 of 4 without bit 1:')
c = probe(streams.masked_stream(m))
short = [x for x, v in c.items() if v == min(c.values())]
if len(set(c.values())) == 2 and short == [1]:
    print('LAYOUT interleaved')
elif len(set(c.values())) == 2 and short == [0]:
    print('LAYOUT xcd-major')
else:
    print('LAYOUT unknown', c)
