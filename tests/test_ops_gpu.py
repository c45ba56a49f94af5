"""Op-level parity: HIP kernels (through the C ABI) vs the CPU oracle on identical inputs.
Bit-exact for index outputs (NMS keep lists) and for ROIAlign (same float/double op order);
conv within a float32 accumulation-order tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as oops
import tolerances as tol_

pytestmark = pytest.mark.gpu


def _rand_dets(rng, n, w=1987.0, h=600.0, cluster=True):
    if cluster:   # clusters of overlapping boxes so that suppression actually happens
        nc = max(1, n // 12)
        cx = rng.uniform(0, w, nc); cy = rng.uniform(0, h, nc); s = rng.uniform(16, 300, nc)
        idx = rng.integers(0, nc, n)
        x = cx[idx] + rng.normal(0, 0.15, n) * s[idx]; y = cy[idx] + rng.normal(0, 0.15, n) * s[idx]
        bw = s[idx] * rng.uniform(0.7, 1.4, n); bh = s[idx] * rng.uniform(0.5, 1.2, n)
    else:
        x = rng.uniform(0, w, n); y = rng.uniform(0, h, n); bw = rng.uniform(1, 400, n); bh = rng.uniform(1, 300, n)
    b = np.stack([x - bw / 2, y - bh / 2, x + bw / 2, y + bh / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, w - 1); b[:, 1::2] = np.clip(b[:, 1::2], 0, h - 1)
    sc = np.sort(rng.uniform(0, 1, n))[::-1]
    return np.concatenate([b, sc[:, None]], 1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 300, 1000, 4097, 6000, 16384])     # 16384 = the documented maximum
@pytest.mark.parametrize("thresh", [0.7, 0.3])
def test_nms_bit_exact(dev, n, thresh):
    from stereo_rcnn_amd.model.nms.nms_wrapper import nms
    rng = np.random.default_rng(n * 7 + int(thresh * 10))
    dets = _rand_dets(rng, n)
    ref = oops.nms(dets, thresh)
    got = nms(torch.from_numpy(dets).to(dev), thresh)
    assert got.dtype == torch.int32 and got.dim() == 2 and got.shape[1] == 1
    assert np.array_equal(got.view(-1).cpu().numpy(), ref)


def test_nms_edge_cases(dev):
    from stereo_rcnn_amd.model.nms.nms_wrapper import nms
    assert nms(torch.zeros((0, 5), device=dev), 0.7) == []
    # degenerate / identical / zero-area boxes
    d = np.array([[10, 10, 10, 10, .9], [10, 10, 10, 10, .8], [0, 0, 0, 0, .7], [5, 5, 4, 4, .6],
                  [0, 0, 1986, 599, .5], [0, 0, 1986, 599, .4]], np.float32)
    for th in (0.0, 0.3, 0.7, 1.0):
        assert np.array_equal(nms(torch.from_numpy(d).to(dev), th).view(-1).cpu().numpy(), oops.nms(d, th))
    # all identical boxes: only the first survives
    d = np.tile(np.array([[3, 4, 50, 60, 0.5]], np.float32), (200, 1))
    assert nms(torch.from_numpy(d).to(dev), 0.7).view(-1).tolist() == [0]
    # one box more than the documented maximum: refused loudly, never truncated
    big = torch.zeros((16385, 5), device=dev)
    with pytest.raises(RuntimeError):
        nms(big, 0.7)


def test_nms_legacy_symbol(dev):
    from stereo_rcnn_amd import _lib
    rng = np.random.default_rng(5)
    dets = _rand_dets(rng, 777)
    t = torch.from_numpy(dets).to(dev)
    keep = torch.zeros(777, dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().nms_cuda(keep.data_ptr(), t.data_ptr(), num.data_ptr(), 777, 5, 0.7, _lib.stream())
    assert rc == 1
    ref = oops.nms(dets, 0.7)
    assert int(num[0]) == len(ref) and np.array_equal(keep[:len(ref)].cpu().numpy(), ref)


@pytest.mark.parametrize("a", [8, 15])
@pytest.mark.parametrize("shape", [(1, 8, 38, 125), (2, 16, 19, 63)])
def test_roi_align_legacy_bit_exact(dev, a, shape):
    from stereo_rcnn_amd.model.roi_align.functions.roi_align import RoIAlignFunction
    rng = np.random.default_rng(a * 100 + shape[2])
    feat = rng.normal(0, 1, shape).astype(np.float32)
    n = 40
    x1 = rng.uniform(-20, 1900, n); y1 = rng.uniform(-20, 560, n)
    rois = np.stack([rng.integers(0, shape[0], n).astype(np.float64), x1, y1, x1 + rng.uniform(0, 500, n),
                     y1 + rng.uniform(0, 300, n)], 1).astype(np.float32)
    rois[0, 1:] = 0                      # the zero-padded proposal
    rois[1, 1:] = [1900, 500, 2100, 700]  # sticks out of the image
    scale = shape[2] / 600.0
    ref = oops.roi_align_forward(feat, rois, a, a, scale)
    got = RoIAlignFunction(a, a, scale)(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev))
    assert np.array_equal(got.cpu().numpy(), ref)


def test_roi_align_bad_roi_shape_is_noop(dev):
    from stereo_rcnn_amd.model.roi_align.functions.roi_align import RoIAlignFunction
    feat = torch.ones((1, 4, 8, 8), device=dev)
    out = RoIAlignFunction(3, 3, 1.0)(feat, torch.zeros((5, 4), device=dev))   # roi_align_cuda.c:19-22
    assert float(out.abs().sum()) == 0.0


def test_roi_align_avg_module(dev):
    from stereo_rcnn_amd.model.roi_align.modules.roi_align import RoIAlignAvg
    rng = np.random.default_rng(11)
    feat = rng.normal(0, 1, (1, 32, 38, 125)).astype(np.float32)
    rois = np.array([[0, 100, 50, 400, 300], [0, 0, 0, 0, 0], [0, 1500, 10, 1986, 599]], np.float32)
    ref = oops.roi_align_avg(feat, rois, 7, 7, 38 / 600.0)
    got = RoIAlignAvg(7, 7, 1 / 16.0)(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), 38 / 600.0)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-6)
    # the 2x2 / stride-1 reductions are library launches (srcnn_pool2x2_s1): bit-equal to ATen's poolings of the same lattice
    from stereo_rcnn_amd.model.roi_align.modules.roi_align import RoIAlign, RoIAlignMax
    lattice = RoIAlign(8, 8, 1 / 16.0)(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), 38 / 600.0)
    assert torch.equal(got.cpu(), F.avg_pool2d(lattice.cpu(), kernel_size=2, stride=1))
    gmax = RoIAlignMax(7, 7, 1 / 16.0)(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), 38 / 600.0)
    assert torch.equal(gmax.cpu(), F.max_pool2d(lattice.cpu(), kernel_size=2, stride=1))


@pytest.mark.parametrize("A,pad_c", [(7, 0), (14, 0), (7, 4), (14, 4)])
def test_pyramid_roi_align_fused(dev, A, pad_c):
    """Fused NHWC kernel == per-level legacy op + avg-pool + level routing of the oracle.  pad_c = 0: 8-channel-group
    kernel (output strides multiples of 8, what the forward uses); pad_c = 4: the per-channel kernel behind it."""
    import ctypes
    from stereo_rcnn_amd import _lib
    from oracle import net as onet
    rng = np.random.default_rng(A)
    C, B = 64, 2
    hw = [(150, 497), (75, 249), (38, 125), (19, 63)]
    maps = [rng.normal(0, 1, (B, C, h, w)).astype(np.float32) for h, w in hw]
    n = 120
    cx = rng.uniform(0, 1987, n); cy = rng.uniform(0, 600, n)
    sz = np.exp(rng.uniform(np.log(8), np.log(900), n)); ar = rng.uniform(0.5, 2.0, n)
    bw, bh = sz * np.sqrt(ar), sz / np.sqrt(ar)
    rois = np.stack([rng.integers(0, B, n).astype(np.float64), np.clip(cx - bw / 2, 0, 1986), np.clip(cy - bh / 2, 0, 599),
                     np.clip(cx + bw / 2, 0, 1986), np.clip(cy + bh / 2, 0, 599)], 1).astype(np.float32)
    rois[:5, 1:] = 0
    im_info = torch.tensor([[600.0, 1987.0, 1.6]])
    # oracle: pyramid_roi_feat handles only one batch index per call through rois[:,0]; it passes rois through
    ref = onet.pyramid_roi_feat([torch.from_numpy(m) for m in maps], torch.from_numpy(rois), im_info, kpts=(A == 14))
    tm = [torch.from_numpy(m).to(dev).permute(0, 2, 3, 1).contiguous() for m in maps]
    CS, CO = 2 * C + pad_c, C + pad_c                        # output channel stride / offset
    out = torch.zeros((n, A, A, CS), device=dev)
    ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in tm])
    mh = (ctypes.c_int * 4)(*[h for h, _ in hw]); mw = (ctypes.c_int * 4)(*[w for _, w in hw])
    tr = torch.from_numpy(rois).to(dev)
    _lib.check(_lib.lib().srcnn_pyramid_roi_align(ptrs, mh, mw, C, 600.0, tr.data_ptr(), n, A, out.data_ptr(), CS, CO,
                                                  0, 0, None, _lib.stream()))
    got = out[:, :, :, CO:].permute(0, 3, 1, 2).cpu().numpy()
    lv_dev = onet.roi_levels(torch.from_numpy(rois))
    assert float(out[:, :, :, :CO].abs().sum()) == 0.0         # other channel slice untouched
    assert np.array_equal(got, ref.numpy()), float(np.abs(got - ref.numpy()).max())
    # device-side roi limit (the keypoint head on the kept detections): the first rois as before, later ones not touched
    lim = torch.tensor([37], dtype=torch.int32, device=dev)
    out2 = torch.zeros((n, A, A, CS), device=dev)
    _lib.check(_lib.lib().srcnn_pyramid_roi_align(ptrs, mh, mw, C, 600.0, tr.data_ptr(), n, A, out2.data_ptr(), CS, CO,
                                                  0, 0, lim.data_ptr(), _lib.stream()))
    assert torch.equal(out2[:37], out[:37]) and float(out2[37:].abs().sum()) == 0.0


def _conv_case(dev, B, H, W, cin, cout, k, stride, pad, relu, res, bn, seed, precision='f32'):
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = None if bn else torch.randn(cout, generator=g)
    bnp = None
    if bn:
        bnp = {'weight': torch.rand(cout, generator=g) + 0.5, 'bias': torch.randn(cout, generator=g),
               'running_mean': torch.randn(cout, generator=g) * 0.1, 'running_var': torch.rand(cout, generator=g) + 0.5}
    ref = F.conv2d(x, w, b, stride, pad)
    if bn:
        ref = F.batch_norm(ref, bnp['running_mean'], bnp['running_var'], bnp['weight'], bnp['bias'], False, 0.0, 1e-5)
    OH, OW = ref.shape[2:]
    r = torch.randn(B, cout, OH, OW, generator=g) if res else None
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    cw = engine.prep_conv(w, b, stride, pad, relu, bn=bnp, device=dev)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous()
    rd = r.to(dev).permute(0, 2, 3, 1).contiguous() if res else None
    y = torch.empty((B, OH, OW, cout), device=dev)
    if precision == 'f16s':      # f16x3 arithmetic, SPLIT16 activations in HBM, both operands DMA'd to LDS
        fo = 1 if cout % 8 == 0 else 0
        xs = engine.act_convert(xd, 0, 1)
        rs = engine.act_convert(rd, 0, 1) if res else None
        engine.conv2d(cw, xs, B, H, W, y, OH, OW, residual=rs, precision='f16x3', x_fmt=1, y_fmt=fo, res_fmt=1 if res else 0)
        y = engine.act_convert(y, fo, 0) if fo else y
    else:
        engine.conv2d(cw, xd, B, H, W, y, OH, OW, residual=rd, precision=precision)
    got = y.permute(0, 3, 1, 2).cpu()
    err = float((got - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("case", [
    # B, H, W, cin, cout, k, stride, pad, relu, res, bn
    (1, 19, 63, 64, 64, 1, 1, 0, True, False, True),
    (2, 38, 125, 256, 256, 3, 1, 1, True, False, True),      # layer3 conv2 shape
    (2, 38, 125, 256, 1024, 1, 1, 0, True, True, True),      # layer3 conv3 + residual
    (2, 75, 249, 512, 256, 1, 2, 0, True, False, True),      # stride-2 1x1 (layer3.0.conv1)
    (1, 150, 497, 64, 64, 3, 1, 1, True, False, True),       # layer1 conv2, big M
    (1, 10, 32, 256, 512, 3, 1, 1, True, False, False),      # RPN conv on P6 (bias, no bn)
    (1, 19, 63, 1024, 24, 1, 1, 0, False, False, False),     # fused RPN heads, Cout=24 (N guard)
    (1, 7, 9, 2048, 512, 1, 1, 0, False, False, False),      # small M, long K -> split-K
    (3, 14, 14, 256, 256, 3, 1, 1, True, False, False),      # kpts tower
])
@pytest.mark.parametrize("precision", ['f32', 'f16x3', 'f16s'])
def test_conv_engine_vs_torch_cpu(dev, case, precision):
    """Both engines must meet the SAME tolerance (the f16x3 split is fp32-class by construction)."""
    _conv_case(dev, *case, seed=hash(case) % 1000, precision=precision)


@pytest.mark.parametrize("plan", [
    # (tile_mr, tile_nr, waves, stages, splits): every SPLIT16 kernel instantiation, with and without split-K
    (1, 1, 4, 2, 1), (1, 1, 4, 4, 1), (1, 1, 4, 4, 3), (2, 1, 4, 2, 1), (2, 1, 4, 3, 2), (1, 2, 4, 2, 1), (1, 2, 4, 3, 1),
    (2, 2, 4, 2, 2), (2, 2, 8, 2, 1), (2, 2, 8, 4, 1), (2, 2, 8, 4, 9), (4, 2, 8, 3, 1), (4, 2, 8, 3, 4),
    (4, 4, 8, 2, 1), (4, 4, 8, 2, 3),       # 256x256 on 8 waves of 64x128 (in-place B fragments), two-pass epilogue
])
@pytest.mark.parametrize("case", [
    (2, 23, 37, 96, 200, 3, 1, 1, True, True, True),      # ragged M (1702) and N (200) tails, image-border taps
    (1, 9, 11, 64, 264, 1, 1, 0, False, False, False),    # K = 2 tiles: shorter than the deepest DMA ring
    (1, 12, 40, 32, 64, 1, 1, 0, True, False, True),      # K = 1 tile
])
def test_conv_f16s_every_plan(dev, plan, case):
    """Forces each launch plan of the DMA-ring kernel (csrc/conv_f16s.hip) instead of the autotuned one."""
    from stereo_rcnn_amd import engine
    saved_tuned, saved_flag = dict(engine._TUNED), engine.AUTOTUNE

    class Forced(dict):
        def get(self, key, default=None):
            return plan
    engine._TUNED = Forced()
    try:
        _conv_case(dev, *case, seed=11, precision='f16s')
    finally:
        engine._TUNED, engine.AUTOTUNE = saved_tuned, saved_flag


@pytest.mark.parametrize("plan", [None, (1, 1, 4, 2, 1), (1, 1, 4, 4, 1), (2, 1, 4, 3, 1), (1, 2, 4, 2, 1), (2, 2, 4, 2, 1), (2, 2, 8, 4, 1),
                                  (4, 2, 8, 3, 1), (2, 2, 8, 2, 3), (4, 4, 8, 2, 1)])
@pytest.mark.parametrize("case", [
    # B, OH, OW, cin (conv3 input), cin2 (block input), cout, stride2
    (2, 19, 31, 64, 64, 256, 1),         # layer1.0: same grid, ragged M (1178)
    (2, 12, 21, 128, 256, 512, 2),       # layer2.0 .. layer4.0: the block input at twice the resolution (odd sizes)
    (1, 7, 9, 32, 32, 200, 2),           # one K tile per input, ragged N
])
def test_conv_with_projection_shortcut_as_second_operand(dev, plan, case):
    """srcnn_conv_desc.x2 (engine.prep_conv_shortcut): relu(bn3(conv3(t)) + bn_d(downsample(x))) of a bottleneck's first block
    (resnet.py:86-100) as ONE launch over K = [channels of t | channels of x] against torch on the CPU, both BNs folded; every
    tile plan that may carry it (a split-K or 256x256 request falls back / is ignored: the result must still be right)."""
    from stereo_rcnn_amd import engine
    B, OH, OW, cin, cin2, cout, s2 = case
    g = torch.Generator().manual_seed(cin + cout + s2)
    H2, W2 = OH * s2 - (s2 - 1), OW * s2 - (s2 - 1)          # smallest input whose stride-s2 grid is (OH, OW)
    t = torch.randn(B, cin, OH, OW, generator=g)
    x = torch.randn(B, cin2, H2, W2, generator=g)
    w3 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    wd = torch.randn(cout, cin2, 1, 1, generator=g) / cin2 ** 0.5
    bn = lambda: {'weight': torch.rand(cout, generator=g) + 0.5, 'bias': torch.randn(cout, generator=g),
                  'running_mean': torch.randn(cout, generator=g) * 0.1, 'running_var': torch.rand(cout, generator=g) + 0.5}
    bn3, bnd = bn(), bn()
    fbn = lambda v, b: F.batch_norm(v, b['running_mean'], b['running_var'], b['weight'], b['bias'], False, 0.0, 1e-5)
    ref = F.relu(fbn(F.conv2d(t, w3), bn3) + fbn(F.conv2d(x, wd, None, s2), bnd))
    cw = engine.prep_conv_shortcut(w3, bn3, wd, bnd, s2, device=dev)
    ts = engine.act_convert(t.to(dev).permute(0, 2, 3, 1).contiguous(), 0, 1)
    xs = engine.act_convert(x.to(dev).permute(0, 2, 3, 1).contiguous(), 0, 1)
    y = torch.empty((B, OH, OW, cout), device=dev)
    engine.conv2d(cw, ts, B, OH, OW, y, OH, OW, precision='f16x3', x_fmt=1, y_fmt=1, x2=xs, H2=H2, W2=W2, plan=plan)
    got = engine.act_convert(y, 1, 0).permute(0, 3, 1, 2).cpu()
    err = float((got - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("plans", [
    [(2, 2, 4, 2, 1), (4, 2, 8, 3, 1), (4, 4, 8, 2, 1), (4, 4, 4, 2, 1), (4, 2, 4, 2, 1), (4, 2, 4, 3, 1), (2, 4, 4, 2, 1), (2, 4, 4, 3, 1)],       # per-wave tile >= 2x2: one accumulator chain
    [(2, 1, 4, 2, 1), (2, 1, 4, 3, 1), (1, 2, 4, 3, 1), (2, 2, 8, 2, 1), (2, 2, 8, 4, 1)],   # two chains
    [(1, 1, 4, 2, 1), (1, 1, 4, 4, 1)],                                         # three chains
])
def test_conv_f16s_plans_of_one_order_class_give_the_same_bits(dev, plans):
    """Without split-K, the tile plans of the SPLIT16 kernel whose waves keep the three split products of a K step in the
    same number of accumulator chains add a row's products in the same order: such a plan changes WHICH workgroup computes
    a row, not its bits -- and a device-side row limit never changes the bits of the rows it leaves.  (Across classes, and
    with split-K, results differ in the last bits: the plan-to-plan rounding the lazy keypoint head's tolerance refers to.)"""
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(5)
    B, H, W, cin, cout = 3, 14, 28, 256, 256
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 48.0
    cw = engine.prep_conv(w, torch.randn(cout, generator=g), 1, 1, True, device=dev)
    xs = engine.act_convert(x, 0, 1)
    lim = torch.tensor([1], dtype=torch.int32, device=dev)
    first = None
    for plan in plans:
        y = torch.zeros((B, H, W, cout), device=dev)
        engine.conv2d(cw, xs, B, H, W, y, H, W, precision='f16x3', x_fmt=1, y_fmt=1, plan=plan)
        z = torch.zeros((B, H, W, cout), device=dev)
        engine.conv2d(cw, xs, B, H, W, z, H, W, precision='f16x3', x_fmt=1, y_fmt=1, plan=plan, m_limit=lim, m_limit_mul=H * W)
        y, z = y.view(torch.int32).cpu(), z.view(torch.int32).cpu()
        assert torch.equal(z[0], y[0]), plan                 # the row-limited launch: same bits for the rows it computes
        first = y if first is None else first
        assert torch.equal(y, first), plan


@pytest.mark.parametrize("precision,W", [('f32', 131), ('f16x3', 131), ('f16x3+split16', 131), ('f16x3+split16', 130)])
def test_conv_stem_vs_torch_cpu(dev, precision, W):
    """7x7/2 stem through the packed NHWC4 image; '+split16' = the packed image in SPLIT16 form read by the DMA engine
    (odd and even padded widths: the SPLIT16 groups are aligned per row)."""
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(1)
    B, H = 2, 75
    xfmt = 1 if precision.endswith('+split16') else 0
    precision = precision.split('+')[0]
    x = torch.randn(B, 3, H, W, generator=g) * 50
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.01
    bnp = {'weight': torch.rand(64, generator=g) + 0.5, 'bias': torch.randn(64, generator=g),
           'running_mean': torch.randn(64, generator=g) * 0.1, 'running_var': torch.rand(64, generator=g) + 0.5}
    ref = F.relu(F.batch_norm(F.conv2d(x, w, None, 2, 3), bnp['running_mean'], bnp['running_var'], bnp['weight'],
                              bnp['bias'], False, 0.0, 1e-5))
    refp = F.max_pool2d(ref, 3, 2, 0, ceil_mode=True)
    cw = engine.prep_stem(w, bnp, device=dev)
    packed = torch.empty((B, H + 6, W + 8, 4), device=dev)
    engine.stem_pack(x.to(dev), packed, out_fmt=xfmt)
    OH, OW = ref.shape[2:]
    y = torch.empty((B, OH, OW, 64), device=dev)
    engine.conv2d(cw, packed, B, H + 6, W + 8, y, OH, OW, x_cstride=4, precision=precision, x_fmt=xfmt)
    assert float((y.permute(0, 3, 1, 2).cpu() - ref).abs().max()) < 1e-4
    PH, PW = refp.shape[2:]
    p = torch.empty((B, PH, PW, 64), device=dev)
    engine.maxpool3x3s2_ceil(y, B, OH, OW, 64, p, PH, PW)
    assert float((p.permute(0, 3, 1, 2).cpu() - refp).abs().max()) < 1e-4


def test_act_convert_roundtrip(dev):
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(3, 5, 7, 64, generator=g) * torch.tensor([1e-2, 1.0, 300.0, 30.0]).repeat(16)).to(dev)
    s = engine.act_convert(x, 0, 1)
    back = engine.act_convert(s, 1, 0)
    # hi + lo reproduces the value to ~2^-22 relative while lo is a normal f16 (|x| >~ 0.1); below that
    # lo is subnormal-limited to an ABSOLUTE error of 2^-25 ~ 3e-8
    err = (back - x).abs()
    assert float((err / x.abs().clamp(min=0.125)).max()) < 5e-7, float((err / x.abs().clamp(min=0.125)).max())
    assert torch.equal(engine.act_convert(engine.act_convert(back, 0, 1), 1, 0), back)   # idempotent


@pytest.mark.parametrize("precision", ['f32', 'f16x3', 'f16s'])
def test_deconv2x2_vs_torch_cpu(dev, precision):
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, 256, 14, 14, generator=g)
    w = torch.randn(256, 256, 2, 2, generator=g) / 16
    b = torch.randn(256, generator=g)
    ref = F.relu(F.conv_transpose2d(x, w, b, 2))
    cw = engine.prep_deconv2x2(w, b, device=dev)
    y = torch.empty((5, 28, 28, 256), device=dev)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous()
    if precision == 'f16s':
        engine.conv2d(cw, engine.act_convert(xd, 0, 1), 5, 14, 14, y, 14, 14, precision='f16x3', x_fmt=1, y_fmt=1)
        y = engine.act_convert(y, 1, 0)
    else:
        engine.conv2d(cw, xd, 5, 14, 14, y, 14, 14, precision=precision)
    assert float((y.permute(0, 3, 1, 2).cpu() - ref).abs().max()) < 2e-5


def test_upsample_add_subsample_layout(dev):
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(3)
    top = torch.randn(2, 256, 19, 63, generator=g); lat = torch.randn(2, 256, 38, 125, generator=g)
    ref = F.interpolate(top, size=(38, 125), mode='bilinear', align_corners=True) + lat
    td = engine.nchw_to_nhwc(top.to(dev)); ld = engine.nchw_to_nhwc(lat.to(dev))
    assert torch.equal(td.cpu(), top.permute(0, 2, 3, 1).contiguous())
    y = torch.empty_like(ld)
    engine.upsample_add(td, 19, 63, ld, 2, 38, 125, 256, y)
    got = engine.nhwc_to_nchw(y).cpu()
    assert float((got - ref).abs().max()) < 5e-6
    s = torch.empty((2, 10, 32, 256), device=dev)
    engine.subsample2(td, 2, 19, 63, 256, s, 10, 32)
    assert torch.equal(engine.nhwc_to_nchw(s).cpu(), top[:, :, ::2, ::2])


@pytest.mark.parametrize("pre_nms,quant", [(500, 0.01), (1000, 0.001), (6000, 0.05), (300, 1.0)])
def test_proposal_layer_ties_and_selection(dev, pre_nms, quant):
    """Top-K with massive score ties (scores quantised so that the K-th value is shared by many anchors):
    the stable (score desc, index asc) order must match the oracle's stable sort exactly."""
    from oracle import proposal as oprop
    from stereo_rcnn_amd.model.rpn.proposal_layer import _ProposalLayer
    g = torch.Generator().manual_seed(pre_nms)
    shapes = [[38, 125], [19, 63], [10, 32], [5, 16], [3, 8]]
    A = sum(3 * h * w for h, w in shapes)
    sc = torch.rand(1, A, generator=g)
    sc = torch.round(sc / quant) * quant if quant < 1.0 else torch.full_like(sc, 0.75)     # quant=1: ALL tied
    probs = torch.stack((1 - sc, sc), 2).contiguous()
    deltas = (torch.randn(1, A, 6, generator=g) * 0.3).contiguous()
    info = torch.tensor([[600.0, 1987.0, 1.6]])
    rl_ref, rr_ref, extra = oprop.proposal_layer(probs, deltas, info, shapes, pre_nms_top_n=pre_nms, post_nms_top_n=120)
    layer = _ProposalLayer(16, [0.5, 1, 2])
    rl, rr = layer.run(probs.to(dev), deltas.to(dev), info.to(dev), shapes, pre_nms, 120, 0.7)
    assert int(layer.last_num_valid[0]) == min(120, len(extra['keep'][0]))
    assert tol_.observe('proposal_isolated_px', (rl.cpu() - rl_ref).abs().max()) < tol_.PROPOSAL_ISOLATED_PX
    assert tol_.observe('proposal_isolated_px', (rr.cpu() - rr_ref).abs().max()) < tol_.PROPOSAL_ISOLATED_PX


@pytest.mark.parametrize("hw,short", [((375, 1242), 600), ((370, 1224), 600), ((374, 1238), 600), ((376, 1241), 600),
                                      ((120, 400), 192), ((37, 53), 100), ((64, 48), 31)])
def test_preprocess_kernel_bit_exact_vs_opencv_restatement(dev, hw, short):
    """A0: device preprocessing == oracle/preprocess.py (OpenCV's float INTER_LINEAR path restated operation by operation)
    BIT FOR BIT, on every KITTI frame size (370..376 x 1224..1242), an up-scale with a non-dyadic factor and a down-scale."""
    from oracle import preprocess as opre
    from stereo_rcnn_amd import engine, fixture
    l, _ = fixture.synthetic_pair(7, hw[0], hw[1])
    ref, s_ref = opre.prepare_image(l, short)
    got, s = engine.preprocess(torch.from_numpy(l).to(dev), short)
    assert s == s_ref and tuple(got.shape) == tuple(ref.shape)
    assert torch.equal(got.cpu(), torch.from_numpy(ref))
    # and close to the torch restatement the synthetic fixtures were generated with (ATen evaluates the tap position in
    # float32: up to 2e-3 away for non-dyadic scales, 2e-5 for the 1.6 of a 375-row frame)
    if round(hw[0] * s) == int(hw[0] * s) and round(hw[1] * s) == int(hw[1] * s) and s > 1:
        old, _ = fixture.preprocess(l, short)
        assert float((got.cpu() - old).abs().max()) < (2e-4 if hw[0] == 375 else 5e-3)


@pytest.mark.parametrize("fmt", [0, 1])
def test_preprocess_fused_stem_pack(dev, fmt):
    """(f)2: the packed stem input written by the fused pass == srcnn_stem_pack of the planar output, byte for byte,
    in both formats (odd padded row length: 1987 + 8)."""
    from stereo_rcnn_amd import engine, fixture
    l, _ = fixture.synthetic_pair(9, 375, 1242)
    img = torch.from_numpy(l).to(dev)
    OH, OW, _ = engine.preprocess_size(375, 1242, 600)
    packed = torch.full((1, OH + 6, OW + 8, 4), 7.0, device=dev)
    planar, _ = engine.preprocess(img, 600, packed=packed, packed_fmt=fmt)
    want = torch.full((1, OH + 6, OW + 8, 4), 7.0, device=dev)
    engine.stem_pack(planar, want, 0, out_fmt=fmt)
    torch.cuda.synchronize()
    assert torch.equal(packed.view(torch.int32), want.view(torch.int32))
    only_packed = torch.full_like(packed, 7.0)
    none, _ = engine.preprocess(img, 600, packed=only_packed, packed_fmt=fmt, planar=False)
    assert none is None and torch.equal(only_packed.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_forward_images_equals_forward_of_preprocessed(dev, precision):
    """forward_images (fused A0 -> stem) gives exactly the forward of the separately preprocessed tensors."""
    from stereo_rcnn_amd import engine, fixture
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    m = resnet(('__background__', 'Car'), 101, pretrained=False)
    m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3))
    m.cuda().eval()
    m.precision = precision
    l, r = fixture.synthetic_pair(3, 120, 400)
    lu, ru = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    with torch.no_grad():
        out_f, iml, imr, info = m.forward_images(lu, ru, target_short=192)
        out_f = [o.clone() for o in out_f[:8]]
        iml, info = iml.clone(), info.clone()
        tl, s = engine.preprocess(lu, 192)
        tr, _ = engine.preprocess(ru, 192)
        assert torch.equal(tl, iml) and float(info[0, 2]) == float(torch.tensor(s, dtype=torch.float32))
        out = m(tl, tr, info)
    for a, b in zip(out_f, out[:8]):
        assert torch.equal(a, b)


def _conv_ref64(x_nhwc, cw):
    w = cw.weight.double().permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(x_nhwc.double().permute(0, 3, 1, 2), w, cw.bias.double(), stride=cw.stride, padding=cw.pad)
    return (y.relu() if cw.relu else y).permute(0, 2, 3, 1)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e3])
@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("shifted", [False, True])
def test_split16_dynamic_range_trained_like_weights(dev, scale, k, shifted):
    """the f16x3 / SPLIT16 engine with activations scaled x1e-3 and x1e3 and a trained-like weight
    distribution (frozen-BN per-channel gains spanning 1e-2 .. 16 folded into the weights) against float64 convolution,
    through two chained layers so that the re-split SPLIT16 intermediate is exercised.  The range guard stays clear.
    shifted: every tensor stored x 2^k with the k a calibration would choose (in_shift / out_shift of
    engine.conv2d, exact power-of-two bookkeeping) -- the x1e-3 case then has the x1 case's 2e-7, not the 2e-5 of f16
    subnormal `lo` halves."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(int(k * 10 + abs(torch.log10(torch.tensor(scale)).item())))
    C = 128
    mk = lambda cout, cin, kk: torch.randn(cout, cin, kk, kk, generator=g) * (2.0 / (cin * kk * kk)) ** 0.5
    # gains 1e-2 .. 16 (1e-2 .. 2 at x1e3, where two layers of gain 16 would legitimately leave the f16 range: see the guard test)
    gain = lambda c: 10.0 ** (torch.rand(c, generator=g) * (2.3 if scale > 1 else 3.2) - 2.0)
    bn = lambda c: {'weight': gain(c), 'bias': torch.randn(c, generator=g) * 0.1 * scale,
                    'running_mean': torch.randn(c, generator=g) * 0.1 * scale, 'running_var': torch.ones(c)}
    c1 = engine.prep_conv(mk(C, C, k), None, 1, k // 2, True, bn(C), dev)
    c2 = engine.prep_conv(mk(C, C, 1), None, 1, 0, False, bn(C), dev)
    B, H, W = 1, 24, 40
    x = (torch.randn(B, H, W, C, generator=g) * scale).to(dev)
    S = _lib.FMT_SPLIT16
    engine.range_flag(reset=True)
    ref_mid = _conv_ref64(x, c1).float()
    ref = _conv_ref64(ref_mid, c2)
    import math
    pick = lambda t: int(round(math.log2(2048.0 / float(t.abs().max())))) if shifted else 0      # plan.calibrate's rule
    kx, km, ko = pick(x), pick(ref_mid), pick(ref)
    xs = engine.act_convert(x * 2.0 ** kx, 0, S)
    mid, out = torch.empty_like(xs), torch.empty_like(xs)
    engine.conv2d(c1, xs, B, H, W, mid, H, W, precision='f16x3', x_fmt=S, y_fmt=S, name='range.c1', in_shift=kx, out_shift=km)
    engine.conv2d(c2, mid, B, H, W, out, H, W, precision='f16x3', x_fmt=S, y_fmt=S, name='range.c2', in_shift=km, out_shift=ko)
    got = engine.act_convert(out, S, 0).double() * 2.0 ** -ko
    flag, name = engine.range_flag(reset=True)
    assert flag == 0, name
    err = float((got - ref).abs().max() / ref.abs().max())
    print('scale %g k=%d %s: max |err| / max |ref| = %.2e (max |ref| %.3g)'
          % (scale, k, 'tensor shifts %d/%d/%d' % (kx, km, ko) if shifted else 'unshifted', err, float(ref.abs().max())))
    # unshifted, below ~6e-5 the f16 halves are subnormal (absolute floor 3e-8); with the tensors' scales every case is the x1 case
    assert err < (2e-6 if (scale >= 1.0 or shifted) else 1e-4), err


def test_split16_range_guard_trips_and_names_the_layer(dev):
    """An activation beyond +-65504 cannot be held by SPLIT16 (hi = f16(v) = inf): the guard records the layer that produced it."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(1)
    C = 64
    cw = engine.prep_conv(torch.randn(C, C, 1, 1, generator=g), torch.zeros(C), 1, 0, False, None, dev)
    S = _lib.FMT_SPLIT16
    x = torch.randn(1, 8, 8, C, generator=g).to(dev)
    y = torch.empty_like(x)
    engine.range_flag(reset=True)
    engine.conv2d(cw, engine.act_convert(x * 100.0, 0, S), 1, 8, 8, y, 8, 8, precision='f16x3', x_fmt=S, y_fmt=S, name='guard.fine')
    assert engine.range_flag(reset=True) == (0, None)
    hot = engine.act_convert(x * 3.0e3, 0, S)              # inputs up to ~1.2e4 (fine), outputs ~8 x that (not)
    assert engine.range_flag()[0] == 0
    engine.conv2d(cw, hot, 1, 8, 8, y, 8, 8, precision='f16x3', x_fmt=S, y_fmt=S, name='guard.hot')
    flag, name = engine.range_flag(reset=False)
    assert flag > 0 and name == 'guard.hot'
    assert engine.range_flag(reset=True)[0] == flag and engine.range_flag()[0] == 0        # sticky until reset
    # F32 output of the same layer is not subject to the format's range
    engine.conv2d(cw, hot, 1, 8, 8, y, 8, 8, precision='f16x3', x_fmt=S, y_fmt=0, name='guard.f32out')
    assert engine.range_flag()[0] == 0 and torch.isfinite(y).all() and float(y.abs().max()) > 65504
    # an input that cannot be represented is caught at the conversion
    engine.act_convert(x * 1.0e5, 0, S)
    flag, name = engine.range_flag()
    assert flag == 9002 and 'input conversion' in name


def test_split16_range_guard_covers_the_split_k_reduction(dev, cout=256):
    """with an explicit split-K plan the SPLIT16 result is written by splitk_reduce_kernel, not by the conv
    epilogue -- an out-of-range sum (and a NaN that a ReLU would launder) must trip the flag there too."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(9)
    C = 256
    S = _lib.FMT_SPLIT16
    cw = engine.prep_conv(torch.randn(cout, C, 1, 1, generator=g), torch.zeros(cout), 1, 0, True, None, dev)
    x = torch.randn(1, 8, 8, C, generator=g).to(dev)
    y = torch.empty(1, 8, 8, cout, device=dev)
    plan = (1, 1, 4, 2, 4)                       # 4 K slices -> workspace reduction writes the result
    engine.range_flag(reset=True)
    engine.conv2d(cw, engine.act_convert(x * 100.0, 0, S), 1, 8, 8, y, 8, 8, precision='f16x3', x_fmt=S, y_fmt=S, plan=plan, name='sk.fine')
    assert engine.range_flag(reset=True) == (0, None)
    engine.conv2d(cw, engine.act_convert(x * 3.0e3, 0, S), 1, 8, 8, y, 8, 8, precision='f16x3', x_fmt=S, y_fmt=S, plan=plan, name='sk.hot')
    flag, name = engine.range_flag(reset=True)
    assert flag > 0 and name == 'sk.hot', (flag, name)


@pytest.mark.parametrize("plan", [(2, 2, 4, 2, 1), (4, 2, 8, 3, 1), (4, 4, 8, 2, 1), (1, 1, 4, 4, 3), (2, 1, 4, 3, 2)])
@pytest.mark.parametrize("B,H,W", [(1, 19, 31), (2, 10, 13)])
def test_conv_mode2_pair_concat_equals_two_launches(dev, plan, B, H, W):
    """Conv mode 2 (the stereo RPN's [left | right] channel concatenation, stereo_rpn.py:77-78, as ONE launch over the 2B
    images) against the two launches it replaces, same plan: bit-identical, including split-K plans (reduction kernel path)
    and tiles that straddle the left / right boundary of the batch."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(B * 100 + H)
    C, N = 64, 72
    S = _lib.FMT_SPLIT16
    w = torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5
    b = torch.randn(N, generator=g)
    cw = engine.prep_conv(w, b, 1, 1, True, None, dev)
    cw2 = engine.ConvW(cw.weight, cw.bias, 3, 3, 1, 1, True, mode=2)
    x = engine.act_convert(torch.randn(2 * B, H, W, C, generator=g).to(dev), 0, S)
    want = torch.zeros(B, H, W, 2 * N + 8, device=dev)
    got = torch.zeros_like(want)
    kw = dict(precision='f16x3', x_fmt=S, y_fmt=S, plan=plan, y_cstride=2 * N + 8)
    engine.conv2d(cw, x, B, H, W, want, H, W, y_coffset=0, **kw)
    engine.conv2d(cw, x, B, H, W, want, H, W, y_coffset=N, x_offset_elems=B * H * W * C, **kw)
    engine.conv2d(cw2, x, 2 * B, H, W, got, H, W, y_coffset=0, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert float(engine.act_convert(got, S, 0)[..., :2 * N].abs().max()) > 0.1


@pytest.mark.parametrize("plan", [(2, 2, 4, 2, 1), (4, 4, 8, 2, 1), (1, 1, 4, 4, 3), (2, 1, 4, 3, 2)])
@pytest.mark.parametrize("keep", [0, 1, 5, 37])
def test_conv_device_side_row_limit(dev, plan, keep):
    """srcnn_conv_desc.m_limit: only rows m < *m_limit * m_limit_mul are needed.  The rows below the limit equal the unlimited
    launch bit for bit; tiles that lie wholly beyond it are not touched (sentinel survives), also through the split-K
    reduction; a limit of 0 launches nothing but exits."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(3)
    S = _lib.FMT_SPLIT16
    R, s, C = 64, 14, 64
    cw = engine.prep_conv(torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5, torch.randn(C, generator=g), 1, 1, True, None, dev)
    x = engine.act_convert(torch.randn(R, s, s, C, generator=g).to(dev), 0, S)
    full = torch.empty(R, s, s, C, device=dev)
    engine.conv2d(cw, x, R, s, s, full, s, s, precision='f16x3', x_fmt=S, y_fmt=S, plan=plan)
    sentinel = 12345.0
    lim = torch.tensor([keep], dtype=torch.int32, device=dev)
    got = engine.act_convert(torch.full((R, s, s, C), sentinel, device=dev), 0, S)
    engine.conv2d(cw, x, R, s, s, got, s, s, precision='f16x3', x_fmt=S, y_fmt=S, plan=plan, m_limit=lim, m_limit_mul=s * s)
    torch.cuda.synchronize()
    rows = keep * s * s
    a, b = got.view(-1, C).view(torch.int32), full.view(-1, C).view(torch.int32)
    assert torch.equal(a[:rows], b[:rows])
    bm = 64 * plan[0]
    first_untouched = -(-rows // bm) * bm if rows else 0
    tail = engine.act_convert(got, S, 0).view(-1, C)[first_untouched:]
    assert tail.numel() == 0 or bool((tail == sentinel).all())


def test_gather_rows_and_decode_kept_kpts_vs_full_decode(dev):
    """srcnn_gather_rows / srcnn_decode_kept_kpts: the keypoint part of the decode for a keep list, fed with probabilities in
    KEPT order, writes exactly the rows the full decode writes for those rois and leaves the other rows alone."""
    from stereo_rcnn_amd import _lib
    from stereo_rcnn_amd import postprocess as hpost
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    n, G = 300, 28
    rois = torch.rand(1, n, 5, generator=g) * 300
    rois[:, :, 3:] += rois[:, :, 1:3] + 8
    rois[:, :, 0] = 0
    rois_r = rois.clone()
    cls = torch.rand(1, n, 2, generator=g)
    bp, dp = torch.randn(1, n, 12, generator=g) * 0.1, torch.randn(1, n, 10, generator=g)
    kp, lp, rp = torch.rand(n, 4 * G, generator=g), torch.rand(n, G, generator=g), torch.rand(n, G, generator=g)
    info = torch.tensor([[192.0, 640.0, 1.6]])
    t = lambda v: v.to(dev)
    det = hpost.decode_detections(t(rois), t(rois_r), t(cls), t(bp), t(dp), t(kp), t(lp), t(rp), t(info))
    keep = torch.tensor([7, 299, 0, 42, 13] + [-1] * (n - 5), dtype=torch.int32, device=dev)
    num = torch.tensor([5], dtype=torch.int32, device=dev)
    gathered = torch.empty(n, 5, device=dev)
    _lib.check(L.srcnn_gather_rows(t(rois)[0].contiguous().data_ptr(), keep.data_ptr(), n, 5, gathered.data_ptr(), _lib.stream()))
    assert torch.equal(gathered[:5].cpu(), rois[0][[7, 299, 0, 42, 13]]) and torch.equal(gathered[5:].cpu(), rois[0][[0] * (n - 5)])
    idx = keep[:5].long()
    out = torch.full((n, 5), -7.0, device=dev)
    kk, ll, rr = torch.zeros(n, 4 * G, device=dev), torch.zeros(n, G, device=dev), torch.zeros(n, G, device=dev)
    kk[:5], ll[:5], rr[:5] = t(kp)[idx], t(lp)[idx], t(rp)[idx]
    rl = t(rois)[0].contiguous()
    _lib.check(L.srcnn_decode_kept_kpts(rl.data_ptr(), kk.data_ptr(), ll.data_ptr(), rr.data_ptr(), keep.data_ptr(), num.data_ptr(),
                                        t(info).data_ptr(), n, G, out.data_ptr(), _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(out[idx], det['kpts'][idx])
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    mask[idx] = False
    assert bool((out[mask] == -7.0).all())


def test_conv_row_limit_never_reads_rows_beyond_it(dev):
    """Rows beyond the device-side limit inside the last computed tile must not be READ: they may hold another format's bits
    (a buffer last written by the fp32 engine) -- NaNs here.  The rows below the limit still equal the unlimited launch on clean
    input and the range guard stays clear."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(4)
    S = _lib.FMT_SPLIT16
    R, s, C, keep = 8, 14, 64, 3
    cw = engine.prep_conv(torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5, torch.randn(C, generator=g), 1, 1, True, None, dev)
    clean = torch.randn(R, s, s, C, generator=g).to(dev)
    xs = engine.act_convert(clean, 0, S)
    want = torch.empty(R, s, s, C, device=dev)
    engine.conv2d(cw, xs, R, s, s, want, s, s, precision='f16x3', x_fmt=S, y_fmt=S, plan=(4, 2, 8, 3, 1))
    poisoned = xs.clone()
    poisoned.view(R, -1)[keep:] = float('nan')                    # NaN bit patterns in every 16-bit half of the rows beyond
    got = torch.zeros(R, s, s, C, device=dev)
    lim = torch.tensor([keep], dtype=torch.int32, device=dev)
    engine.range_flag(reset=True)
    engine.conv2d(cw, poisoned, R, s, s, got, s, s, precision='f16x3', x_fmt=S, y_fmt=S, plan=(4, 2, 8, 3, 1), m_limit=lim,
                  m_limit_mul=s * s, name='limit.poison')
    torch.cuda.synchronize()
    assert engine.range_flag(reset=True) == (0, None)
    rows = keep * s * s
    assert torch.equal(got.view(-1, C).view(torch.int32)[:rows], want.view(-1, C).view(torch.int32)[:rows])
    assert torch.isfinite(engine.act_convert(got, S, 0)).all()


@pytest.mark.parametrize("B", [1, 3])
def test_rpn_score_of_all_levels_in_one_launch_equals_the_per_level_launches(dev, B):
    """srcnn_rpn_score_levels (one launch for the five pyramid levels) writes exactly what five srcnn_rpn_score launches write."""
    import ctypes
    from stereo_rcnn_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B)
    shapes = [(38, 125), (19, 63), (10, 32), (5, 16), (3, 8)]
    A = sum(3 * h * w for h, w in shapes)
    heads = [(torch.randn(B, h * w, 24, generator=g) * 3).to(dev).contiguous() for h, w in shapes]
    pa, da = torch.zeros(B, A, 2, device=dev), torch.zeros(B, A, 6, device=dev)
    pb, db = torch.full((B, A, 2), -1.0, device=dev), torch.full((B, A, 6), -1.0, device=dev)
    off = 0
    for (h, w), hd in zip(shapes, heads):
        _lib.check(L.srcnn_rpn_score(hd.data_ptr(), B, h * w, 24, pa.data_ptr(), da.data_ptr(), off, A, _lib.stream()))
        off += 3 * h * w
    ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in heads])
    hw = (ctypes.c_int * 5)(*[h * w for h, w in shapes])
    _lib.check(L.srcnn_rpn_score_levels(ptrs, hw, 5, B, 24, pb.data_ptr(), db.data_ptr(), A, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(da, db)
    assert L.srcnn_rpn_score_levels(ptrs, hw, 5, B, 24, pb.data_ptr(), db.data_ptr(), A + 3, _lib.stream()) != 0      # anchor count must match


def test_proposal_selection_batched_and_repeated(dev):
    """The radix select's passes are single launches whose LAST workgroup picks the digit (arrival counter per image): a batch of
    three images with different tie structures, run three times on one workspace -- counters and histograms must come back to zero
    by themselves, every run and every image equal to the oracle's stable top-K."""
    from oracle import proposal as oprop
    from stereo_rcnn_amd.model.rpn.proposal_layer import _ProposalLayer
    g = torch.Generator().manual_seed(11)
    shapes = [[38, 125], [19, 63], [10, 32], [5, 16], [3, 8]]
    A = sum(3 * h * w for h, w in shapes)
    sc = torch.rand(3, A, generator=g)
    sc[1] = torch.round(sc[1] / 0.01) * 0.01                      # heavy ties
    sc[2] = 0.5                                                   # all tied: the index tie-break decides everything
    probs = torch.stack((1 - sc, sc), 2).contiguous()
    deltas = (torch.randn(3, A, 6, generator=g) * 0.3).contiguous()
    info = torch.tensor([[600.0, 1987.0, 1.6]] * 3)
    rl_ref, rr_ref, extra = oprop.proposal_layer(probs, deltas, info, shapes, pre_nms_top_n=1000, post_nms_top_n=150)
    layer = _ProposalLayer(16, [0.5, 1, 2])
    for _ in range(3):
        rl, rr = layer.run(probs.to(dev), deltas.to(dev), info.to(dev), shapes, 1000, 150, 0.7)
        for b in range(3):
            assert int(layer.last_num_valid[b]) == min(150, len(extra['keep'][b]))
        assert tol_.observe('proposal_isolated_px', (rl.cpu() - rl_ref).abs().max()) < tol_.PROPOSAL_ISOLATED_PX
        assert tol_.observe('proposal_isolated_px', (rr.cpu() - rr_ref).abs().max()) < tol_.PROPOSAL_ISOLATED_PX


def _split16(x_nhwc):
    from stereo_rcnn_amd import _lib, engine
    return engine.act_convert(x_nhwc.contiguous(), _lib.FMT_F32, _lib.FMT_SPLIT16)


@pytest.mark.parametrize("mode,n2", [(0, 6), (0, 20), (1, 6)])
def test_fused_head_mfma_form_final_equals_two_launches(dev, mode, n2):
    """srcnn_conv_desc.head_wf, final form: a 1x1 conv (or 2x2 / 2 deconvolution) to 256-channel pixels + bias + ReLU with an
    n2-channel 1x1 head applied in its epilogue on the matrix pipe, against the two launches it replaces (activations stored as
    SPLIT16, head conv on them): same 3-term split of the same hi / lo halves, only the summation order differs.  Ragged M
    (rows beyond it in the last tile), B = 3."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(100 * mode + n2)
    B, H, W, cin = 3, 9, 13, 64                                   # M = 351: one full 256-row tile + a ragged one
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    xs = _split16(x)
    if mode == 0:
        cw = engine.prep_conv(torch.randn(256, cin, 1, 1, generator=g) * 0.2, torch.randn(256, generator=g) * 0.1, 1, 0, True, device=dev)
        OH, OW, opix = H, W, B * H * W
    else:
        cw = engine.prep_deconv2x2(torch.randn(cin, 256, 2, 2, generator=g) * 0.2, torch.randn(256, generator=g) * 0.1, True, device=dev)
        OH, OW, opix = H, W, B * 2 * H * 2 * W
    hcw = engine.prep_conv(torch.randn(n2, 256, 1, 1, generator=g) * 0.1, torch.randn(n2, generator=g) * 0.1, 1, 0, False, device=dev)
    mid = torch.zeros(opix, 256, device=dev)
    engine.conv2d(cw, xs, B, H, W, mid, OH, OW, precision='f16x3', x_fmt=1, y_fmt=1, plan=(4, 4, 8, 2, 1))
    two = torch.zeros(opix, n2, device=dev)
    side = (2 * H, 2 * W) if mode == 1 else (H, W)
    engine.conv2d(hcw, mid, B, side[0], side[1], two, side[0], side[1], precision='f16x3', x_fmt=1, y_fmt=0, plan=(1, 1, 4, 2, 1))
    one = torch.full((opix, n2), float('nan'), device=dev)
    used = engine.conv2d(cw, xs, B, H, W, None, OH, OW, precision='f16x3', x_fmt=1, head2=(hcw, one, 0))
    torch.cuda.synchronize()
    assert used[:4] == (4, 4, 8, 2)
    assert torch.isfinite(one).all()
    scale = float(two.abs().max())
    err = float((one - two).abs().max())
    print('fused head (mode %d, %d channels) vs two launches: %.2e of %.2e' % (mode, n2, err, scale))
    assert err < 3e-6 * max(scale, 1.0)
    again = torch.zeros_like(one)
    engine.conv2d(cw, xs, B, H, W, None, OH, OW, precision='f16x3', x_fmt=1, head2=(hcw, again, 0))
    torch.cuda.synchronize()
    assert torch.equal(one, again)                                 # fixed summation order


@pytest.mark.parametrize("plan", [(4, 4, 8, 2, 1), (2, 2, 8, 2, 1)])
@pytest.mark.parametrize("B,H,W", [(1, 19, 63), (2, 10, 32), (1, 3, 8)])
def test_fused_rpn_head_partial_form_equals_two_launches(dev, plan, B, H, W):
    """srcnn_conv_desc.head_wf, partial form under conv mode 2: RPN_Conv on [left images | right images] with the 24-channel stereo
    head (over [left 512 | right 512]) applied per tile; every (eye, N tile) leaves a plane of partial sums that
    srcnn_rpn_score_parts adds (+ bias) before the pair softmax.  Against: the pair launch writing the (B, h, w, 1024) tensor, the
    head as its own launch, srcnn_rpn_score_levels.  Both tile forms; B * h * w is never a multiple of the tile height, so a tile
    straddles the two eyes (its rows meet different halves of the head's weights)."""
    import ctypes
    from stereo_rcnn_amd import _lib, engine
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(2 * B, H, W, 256, generator=g).to(dev)          # [left images | right images]
    xs = _split16(x)
    wt, bs = (torch.randn(512, 256, 3, 3, generator=g) * 0.03), torch.randn(512, generator=g) * 0.1
    cw = engine.prep_conv(wt, bs, 1, 1, True, device=dev)
    pair = engine.ConvW(cw.weight, cw.bias, 3, 3, 1, 1, True, mode=2)
    hcw = engine.prep_conv(torch.randn(24, 1024, 1, 1, generator=g) * 0.05, torch.randn(24, generator=g) * 0.1, 1, 0, False, device=dev)
    cat = torch.zeros(B, H, W, 1024, device=dev)
    engine.conv2d(pair, xs, 2 * B, H, W, cat, H, W, y_cstride=1024, y_coffset=0, precision='f16x3', x_fmt=1, y_fmt=1, plan=(2, 2, 4, 2, 1))
    hd = torch.zeros(B, H * W, 24, device=dev)
    engine.conv2d(hcw, cat, B, H, W, hd, H, W, precision='f16x3', x_fmt=1, y_fmt=0, plan=(1, 1, 4, 2, 1))
    A = 3 * H * W
    pa, da = torch.zeros(B, A, 2, device=dev), torch.zeros(B, A, 6, device=dev)
    ptr1 = (ctypes.c_void_p * 1)(hd.data_ptr())
    hw1 = (ctypes.c_int * 1)(H * W)
    _lib.check(L.srcnn_rpn_score_levels(ptr1, hw1, 1, B, 24, pa.data_ptr(), da.data_ptr(), A, _lib.stream()))
    parts = torch.full((8, B * H * W, 24), float('nan'), device=dev)
    used = engine.conv2d(pair, xs, 2 * B, H, W, None, H, W, precision='f16x3', x_fmt=1, head2=(hcw, parts, 8), plan=plan)
    assert used[:2] == plan[:2]
    npl = 2 * (512 // (64 * used[1]))
    torch.cuda.synchronize()
    assert torch.isfinite(parts[:npl]).all() and torch.isnan(parts[npl:]).all()
    summed = parts[:npl].sum(0) + hcw.bias.view(1, 24)
    scale = float(hd.abs().max())
    err = float((summed.view(B, H * W, 24) - hd).abs().max())
    print('fused RPN head %s B=%d %dx%d: %d planes, max |d| %.2e of %.2e' % (plan[:2], B, H, W, npl, err, scale))
    assert err < 3e-6 * max(scale, 1.0)
    pb, db = torch.zeros(B, A, 2, device=dev), torch.zeros(B, A, 6, device=dev)
    pp = (ctypes.c_void_p * 1)(parts.data_ptr())
    np1 = (ctypes.c_int * 1)(npl)
    pl1 = (ctypes.c_longlong * 1)(parts.numel() // 8)
    _lib.check(L.srcnn_rpn_score_parts(pp, np1, pl1, hw1, 1, B, hcw.bias.data_ptr(), pb.data_ptr(), db.data_ptr(), A, _lib.stream()))
    torch.cuda.synchronize()
    assert float((pb - pa).abs().max()) < 1e-6 and float((db - da).abs().max()) < 3e-6 * max(scale, 1.0)


@pytest.mark.parametrize("plan", [(1, 1, 4, 2, 1), (2, 1, 4, 2, 1), (2, 2, 4, 2, 1), (2, 2, 8, 4, 1), (4, 2, 8, 3, 1), (4, 4, 8, 2, 1), (1, 1, 4, 4, 3),
                                  None])
@pytest.mark.parametrize("B,H,W,TH,TW,cin", [(2, 38, 125, 19, 63, 64), (2, 11, 17, 6, 9, 96), (1, 5, 1, 3, 1, 32), (1, 7, 9, 7, 9, 32)])
def test_fused_upsample_add_in_the_lateral_conv_equals_two_launches(dev, plan, B, H, W, TH, TW, cin):
    """srcnn_conv_desc.up_top (the FPN top-down addition of stereo_rcnn.py:91-108 inside the lateral 1x1 conv's epilogue) against
    the two launches it replaces -- srcnn_conv2d into a float32 lateral map, then srcnn_upsample_add: bit-identical, on every tile
    the kernel has it for (engine.conv2d turns a request for the 256x256 tile into 256x128 and a split-K plan into its unsplit form), odd map
    sizes (KITTI's 38x125 from 19x63), a one-column map, and a top map of the output's own size (all weights 1 / 0)."""
    from stereo_rcnn_amd import _lib, engine
    g = torch.Generator().manual_seed(H * 7 + W)
    S = _lib.FMT_SPLIT16
    C = 256
    w = torch.randn(C, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(C, generator=g)
    cw = engine.prep_conv(w, b, 1, 0, False, None, dev)
    x = engine.act_convert(torch.randn(B, H, W, cin, generator=g).to(dev), 0, S)
    top = engine.act_convert(torch.randn(B, TH, TW, C, generator=g).to(dev), 0, S)
    lat = torch.empty(B, H, W, C, device=dev)
    want = torch.zeros(B, H, W, C, device=dev)
    got = torch.zeros_like(want)
    used = engine.conv2d(cw, x, B, H, W, got, H, W, precision='f16x3', x_fmt=S, y_fmt=S, plan=plan, up=(top, TH, TW, S))
    assert used[4] == 1 and used[:2] != (4, 4), used
    # the two launches on the plan the fused launch actually ran with (another tile may add the K products in another order)
    used0 = engine.conv2d(cw, x, B, H, W, lat, H, W, precision='f16x3', x_fmt=S, y_fmt=0, plan=used)
    engine.upsample_add(top, TH, TW, lat, B, H, W, C, want, top_fmt=S, y_fmt=S)
    torch.cuda.synchronize()
    assert tuple(used0) == tuple(used)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (used0, used)
    ref = F.interpolate(engine.act_convert(top, S, 0).permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=True) \
        + lat.permute(0, 3, 1, 2)
    assert float((engine.act_convert(got, S, 0).permute(0, 3, 1, 2) - ref).abs().max()) < 2e-5
    # float32 top map (the exact-fp32 engine's pyramid), float32 result
    topf = engine.act_convert(top, S, 0)
    engine.upsample_add(topf, TH, TW, lat, B, H, W, C, want, top_fmt=0, y_fmt=0)
    engine.conv2d(cw, x, B, H, W, got, H, W, precision='f16x3', x_fmt=S, y_fmt=0, plan=used, up=(topf, TH, TW, 0))
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("B,H,W", [(1, 37, 131), (3, 20, 50)])
def test_stem_pack_pair_equals_one_launch_per_eye(dev, fmt, B, H, W):
    """srcnn_stem_pack_pair (both eyes of the batch behind each other in one launch) == srcnn_stem_pack of the lefts into images
    [0, B) and of the rights into [B, 2B), byte for byte, both formats, odd padded row length."""
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(B + W)
    l = torch.randn(B, 3, H, W, generator=g).to(dev)
    r = torch.randn(B, 3, H, W, generator=g).to(dev)
    want = torch.full((2 * B, H + 6, W + 8, 4), 7.0, device=dev)
    got = torch.full_like(want, 7.0)
    engine.stem_pack(l, want, 0, out_fmt=fmt)
    engine.stem_pack(r, want, B, out_fmt=fmt)
    engine.stem_pack_pair(l, r, got, out_fmt=fmt)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
