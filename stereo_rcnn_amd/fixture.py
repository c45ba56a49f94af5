"""Seeded random-init weights (reference state_dict key schema) and synthetic
stereo pairs.  Product-side utility: bench.py, smoke() and the tests all draw
their weights/inputs from here; the oracle never feeds the product path.

The reference ships no checkpoint (README.md:48-52,80), so parity runs on a
seeded state_dict.  Key names/shapes follow resnet.py:228-286 and
stereo_rpn.py:32-40 (`RCNN_layerN.0.<blk>.convK.weight`, ...).  The reference's
own init recipe (resnet.py:123-129, stereo_rcnn.py:47-85) is meant for training
from ImageNet weights; applied to a frozen-BN random network it makes the
activations explode through 33 residual blocks, which would turn an absolute
1e-4 tolerance into noise.  The fixture therefore draws fan-in-scaled conv
weights and non-trivial frozen-BN statistics (so BN folding is exercised) that
keep activations O(1) - a pure function of the seed, identical on every machine.
"""
import math
from collections import OrderedDict

import torch

R101 = (3, 4, 23, 3)
R50 = (3, 4, 6, 3)     # extension (BASELINE config 5); the reference hard-codes R101 (resnet.py:229)


def _conv(g, cout, cin, k, gain=2.0):
    std = math.sqrt(gain / (cin * k * k))
    return torch.randn(cout, cin, k, k, generator=g) * std


def _bn(g, c, gamma_lo, gamma_hi):
    return {
        'weight': torch.rand(c, generator=g) * (gamma_hi - gamma_lo) + gamma_lo,
        'bias': torch.randn(c, generator=g) * 0.1,
        'running_mean': torch.randn(c, generator=g) * 0.1,
        'running_var': torch.rand(c, generator=g) * 0.4 + 0.8,
    }


def make_state_dict(seed=3, layers=R101, n_classes=2, rpn_cls_std=0.01, cls_std=0.005):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def put_bn(prefix, c, lo=0.8, hi=1.2):
        for k, v in _bn(g, c, lo, hi).items():
            sd[prefix + '.' + k] = v

    # stem: conv1 7x7/2 + bn1 (resnet.py:109-110, wrapped by RCNN_layer0 :236)
    sd['RCNN_layer0.0.weight'] = _conv(g, 64, 3, 7, gain=2.0) / 60.0  # inputs are mean-subtracted 0..255 pixels
    put_bn('RCNN_layer0.1', 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for b in range(nblk):
            p = 'RCNN_layer%d.0.%d' % (li, b)
            sd[p + '.conv1.weight'] = _conv(g, planes, inplanes, 1)
            put_bn(p + '.bn1', planes)
            sd[p + '.conv2.weight'] = _conv(g, planes, planes, 3)
            put_bn(p + '.bn2', planes)
            sd[p + '.conv3.weight'] = _conv(g, planes * 4, planes, 1)
            put_bn(p + '.bn3', planes * 4, 0.2, 0.4)      # small residual-branch gain
            if b == 0:
                sd[p + '.downsample.0.weight'] = _conv(g, planes * 4, inplanes, 1, gain=1.0)
                put_bn(p + '.downsample.1', planes * 4)
            inplanes = planes * 4

    def conv_b(name, cout, cin, k, gain=1.0, wstd=None):
        sd[name + '.weight'] = _conv(g, cout, cin, k, gain) if wstd is None else \
            torch.randn(cout, cin, k, k, generator=g) * wstd
        sd[name + '.bias'] = torch.randn(cout, generator=g) * 0.05

    conv_b('RCNN_toplayer', 256, 2048, 1)
    for i in (1, 2, 3):
        conv_b('RCNN_smooth%d' % i, 256, 256, 3)
    for i, cin in ((1, 1024), (2, 512), (3, 256)):
        conv_b('RCNN_latlayer%d' % i, 256, cin, 1)
    conv_b('RCNN_top.0', 2048, 512, 7, gain=2.0)
    conv_b('RCNN_top.3', 2048, 2048, 1, gain=2.0)
    for i in (0, 2, 4, 6, 8, 10):
        conv_b('RCNN_kpts.%d' % i, 256, 256, 3, gain=2.0)
    # ConvTranspose2d(256,256,2,2): weight (in, out, kh, kw) (resnet.py:278)
    sd['RCNN_kpts.12.weight'] = torch.randn(256, 256, 2, 2, generator=g) * math.sqrt(2.0 / 256)
    sd['RCNN_kpts.12.bias'] = torch.randn(256, generator=g) * 0.05
    for name, nout, std in (('RCNN_cls_score', n_classes, cls_std),
                            ('RCNN_bbox_pred', 6 * n_classes, 0.002),
                            ('RCNN_dim_orien_pred', 5 * n_classes, 0.002)):
        sd[name + '.weight'] = torch.randn(nout, 2048, generator=g) * std
        sd[name + '.bias'] = torch.randn(nout, generator=g) * 0.05
    conv_b('kpts_class', 6, 256, 1, wstd=0.02)
    conv_b('RCNN_rpn.RPN_Conv', 512, 256, 3, gain=2.0)
    conv_b('RCNN_rpn.RPN_cls_score', 6, 1024, 1, wstd=rpn_cls_std)
    conv_b('RCNN_rpn.RPN_bbox_pred_left_right', 18, 1024, 1, wstd=0.004)
    return sd


# --------------------------------------------------------------------------- inputs
PIXEL_MEANS_BGR = (102.9801, 115.9465, 122.7717)      # config.py:170


def _smooth_noise(rng, h, w, cells):
    """Bilinear-upsampled coarse uniform noise (deterministic numpy arithmetic only)."""
    import numpy as np
    gh, gw = h // cells + 2, w // cells + 2
    coarse = rng.random((gh, gw, 3))
    ys = np.arange(h) / cells
    xs = np.arange(w) / cells
    y0 = ys.astype(np.int64); fy = (ys - y0)[:, None, None]
    x0 = xs.astype(np.int64); fx = (xs - x0)[None, :, None]
    a = coarse[y0][:, x0]; b = coarse[y0][:, x0 + 1]
    c = coarse[y0 + 1][:, x0]; d = coarse[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def synthetic_pair(seed=3, height=375, width=1242):
    """Seeded synthetic stereo pair, uint8 RGB (H, W, 3) x2.

    left = multi-octave smooth texture; right = left shifted left by a per-row
    disparity in [5, 60] px plus small noise, so photometric terms are non-degenerate.
    """
    import numpy as np
    rng = np.random.default_rng(seed)
    tex = (0.55 * _smooth_noise(rng, height, width + 64, 48)
           + 0.30 * _smooth_noise(rng, height, width + 64, 12)
           + 0.15 * _smooth_noise(rng, height, width + 64, 3))
    disp = np.linspace(5.0, 60.0, height)            # larger disparity towards the bottom rows
    cols = np.arange(width)
    left = tex[:, :width]
    right = np.empty_like(left)
    for r in range(height):
        src = cols + disp[r]
        s0 = np.floor(src).astype(np.int64)
        f = (src - s0)[:, None]
        right[r] = tex[r, s0] * (1 - f) + tex[r, s0 + 1] * f
    right = right + rng.normal(0.0, 2.0 / 255.0, right.shape)
    to_u8 = lambda a: np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)
    return to_u8(left), to_u8(right)


def demo_state_dict(seed=3):
    """Weights of the demo-pair parity case (BASELINE configs[0], tests/golden/reference_demo_pair_*): make_state_dict(seed)
    with the RPN objectness layer scaled by 1/8.  On the reference's natural demo image the unscaled random init drives
    the pair-softmax to EXACTLY 1.0 for 11 607 of the 298 476 anchors, so which 6000 enter the NMS would be a property
    of the sort implementation's tie order (torch 0.3 CUDA sort vs torch 2.x CPU sort vs a stable sort), not of the
    algorithm under test; scaled by 1/8 the 6000th score is 0.906 and only float32-density ties remain."""
    sd = make_state_dict(seed)
    for k in ('RCNN_rpn.RPN_cls_score.weight', 'RCNN_rpn.RPN_cls_score.bias'):
        sd[k] = sd[k] * 0.125
    return sd


def preprocess(img_rgb_u8, target_short=600, max_size=2484, device='cpu'):
    """demo.py:103-129 / blob.py:39-64: RGB->BGR, -PIXEL_MEANS, bilinear resize so the
    short side is `target_short` (OpenCV INTER_LINEAR = half-pixel centres, scale 1/fx),
    HWC -> 1x3xHxW float32.  Returns (tensor, im_scale)."""
    import numpy as np
    import torch.nn.functional as F
    im = img_rgb_u8[:, :, ::-1].astype(np.float32)
    im = im - np.asarray(PIXEL_MEANS_BGR, np.float64).reshape(1, 1, 3)   # float32 - float64 -> float64
    im = im.astype(np.float32)
    scale = float(target_short) / float(min(im.shape[0], im.shape[1]))
    t = torch.from_numpy(np.ascontiguousarray(im)).permute(2, 0, 1).unsqueeze(0).to(device)
    if scale != 1.0:
        t = F.interpolate(t, scale_factor=scale, mode='bilinear', align_corners=False,
                          recompute_scale_factor=False)
    if t.shape[3] > max_size:
        t = t[:, :, :, :max_size]
    return t.contiguous(), scale


def make_inputs(seed=3, height=375, width=1242, device='cpu', target_short=600):
    """(im_left, im_right, im_info) ready for `_StereoRCNN.forward` (demo.py:122-135).
    `target_short` != 600 is only for reduced-size parity cases."""
    l, r = synthetic_pair(seed, height, width)
    tl, s = preprocess(l, target_short, device=device)
    tr, _ = preprocess(r, target_short, device=device)
    info = torch.tensor([[tl.shape[2], tl.shape[3], s]], dtype=torch.float32, device=device)
    return tl, tr, info


# --------------------------------------------------------------------------- a KITTI object tree of synthetic frames
KITTI_VAL_IDS = 3769     # ids of the reference's data/kitti/splits/val.txt (BASELINE.json words it "3712 pairs"; the file has 3769)

DEMO_CALIB_ROWS = (
    ('P0', (721.5377, 0, 609.5593, 0, 0, 721.5377, 172.854, 0, 0, 0, 1, 0)),
    ('P1', (721.5377, 0, 609.5593, -387.5744, 0, 721.5377, 172.854, 0, 0, 0, 1, 0)),
    ('P2', (721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884)),
    ('P3', (721.5377, 0, 609.5593, -339.5242, 0, 721.5377, 172.854, 2.199936, 0, 0, 1, 0.002729905)),
    ('R0_rect', (1, 0, 0, 0, 1, 0, 0, 0, 1)),
    ('Tr_velo_to_cam', (1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)),
)


def write_kitti_tree(root, n_ids=KITTI_VAL_IDS, distinct=16, height=375, width=1242, seed0=3, split_name='val.txt'):
    """A KITTI object `training/` tree (image_2/, image_3/, calib/ -- the layout lib/datasets/kitti.py:60-75 reads, the one
    test_net.run_split takes) whose `n_ids` frames replay `distinct` seeded synthetic PNG pairs: the real split cannot be
    shipped (no dataset offline), its LENGTH and per-frame host work (PNG decode of two 375x1242 images, calibration parse,
    result file) can.  Frame i is a symlink to pair i % distinct, so the tree costs `distinct` x 2 PNGs of disk (tmpfs) and
    every frame still goes through a real file open + PNG decode.  Returns the id list (also written as <root>/<split_name>)."""
    import os
    from PIL import Image
    for d in ('image_2', 'image_3', 'calib', '_pool'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    calib = os.path.join(root, '_pool', 'calib.txt')
    with open(calib, 'w') as fh:
        fh.write('\n'.join('%s: %s' % (k, ' '.join('%.12e' % v for v in vals)) for k, vals in DEMO_CALIB_ROWS) + '\n')
    for p in range(distinct):
        left, right = synthetic_pair(seed0 + p, height, width)
        Image.fromarray(left).save(os.path.join(root, '_pool', 'l_%03d.png' % p))
        Image.fromarray(right).save(os.path.join(root, '_pool', 'r_%03d.png' % p))
    ids = ['%06d' % i for i in range(n_ids)]
    for i, frame in enumerate(ids):
        for d, src in (('image_2', 'l_%03d.png' % (i % distinct)), ('image_3', 'r_%03d.png' % (i % distinct)), ('calib', 'calib.txt')):
            dst = os.path.join(root, d, frame + ('.txt' if d == 'calib' else '.png'))
            if not os.path.lexists(dst):
                os.symlink(os.path.join('..', '_pool', src), dst)
    with open(os.path.join(root, split_name), 'w') as fh:
        fh.write('\n'.join(ids) + '\n')
    return ids
