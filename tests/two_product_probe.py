"""Item 7(ii) probe: what a two-product conv (w rounded to f16 = the a*w_lo product dropped) costs in output accuracy.
Emulated on the CPU oracle (fp32), small network input 192x640; layer groups chosen by state_dict key."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import net as onet
from stereo_rcnn_amd import fixture
torch.set_num_threads(8)
sd = fixture.make_state_dict(3)
l, r, info = fixture.make_inputs(3, 120, 400, target_short=192)
t0 = time.time(); ref = onet.forward(sd, l, r, info); print('forward %.1f s' % (time.time() - t0)); print(sorted(ref.keys()))
keys4 = [k for k in sd if sd[k].dim() == 4]
groups = {
 'kpts tower (6x 3x3)': [k for k in keys4 if 'kpts' in k.lower()],
 'rpn conv': [k for k in keys4 if 'RPN_Conv' in k],
 'fpn smooth': [k for k in keys4 if 'smooth' in k],
 'trunk conv2 (3x3)': [k for k in keys4 if '.conv2.' in k],
 'all 3x3': [k for k in keys4 if sd[k].shape[-1] == 3],
 'everything': keys4,
}
def cmp(a, b, same_rois):
    out = []
    for k in ('bbox_pred', 'cls_prob', 'kpts_prob', 'left_prob', 'right_prob', 'dim_orien_pred', 'rois_left'):
        if k in a and k in b and a[k].shape == b[k].shape:
            out.append('%s %.1e' % (k, float((a[k] - b[k]).abs().max())))
    return '  '.join(out)
for name, ks in groups.items():
    if not ks: print(name, 'no keys'); continue
    sd2 = dict(sd)
    for k in ks: sd2[k] = sd[k].half().float()
    o = onet.forward(sd2, l, r, info)
    same = torch.equal(o['rois_left'], ref['rois_left'])
    print('%-22s %3d tensors: rois identical=%s  %s' % (name, len(ks), same, cmp(ref, o, same)))
