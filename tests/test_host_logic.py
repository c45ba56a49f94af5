"""CPU: host-side logic of the product package (no kernels are launched)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import config as OC
from stereo_rcnn_amd import engine, fixture
from stereo_rcnn_amd.model.utils.config import cfg


def test_cfg_matches_oracle_constants():
    assert cfg.TEST.RPN_PRE_NMS_TOP_N == OC.RPN_PRE_NMS_TOP_N and cfg.TEST.RPN_POST_NMS_TOP_N == OC.RPN_POST_NMS_TOP_N
    assert cfg.TEST.RPN_NMS_THRESH == OC.RPN_NMS_THRESH and cfg.TEST.NMS == OC.TEST_NMS
    assert list(cfg.FPN_ANCHOR_SCALES) == OC.FPN_ANCHOR_SCALES and list(cfg.FPN_FEAT_STRIDES) == OC.FPN_FEAT_STRIDES
    assert cfg.KPTS_GRID == OC.KPTS_GRID and cfg.POOLING_SIZE == OC.POOLING_SIZE
    assert tuple(cfg.TRAIN.BBOX_NORMALIZE_STDS) == OC.BBOX_NORMALIZE_STDS
    assert tuple(cfg.TRAIN.DIM_NORMALIZE_MEANS) == OC.DIM_NORMALIZE_MEANS
    assert cfg['TEST'].NMS == cfg.TEST.NMS            # key and attribute access, like easydict


def test_state_dict_schema_is_the_references():
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    m = resnet(('__background__', 'Car'), 101)
    m.create_architecture()
    keys = {k for k in m.state_dict() if not k.endswith('num_batches_tracked')}
    sd = fixture.make_state_dict(3)
    assert keys == set(sd)
    for k in ('RCNN_layer0.0.weight', 'RCNN_layer0.1.running_var', 'RCNN_layer3.0.22.conv2.weight',
              'RCNN_layer4.0.0.downsample.1.bias', 'RCNN_top.3.bias', 'RCNN_kpts.12.weight',
              'RCNN_rpn.RPN_bbox_pred_left_right.weight', 'kpts_class.bias', 'RCNN_dim_orien_pred.weight'):
        assert k in keys
    assert tuple(sd['RCNN_kpts.12.weight'].shape) == (256, 256, 2, 2)
    m.load_state_dict(sd)
    assert torch.equal(m.RCNN_layer3[0][5].conv2.weight, sd['RCNN_layer3.0.5.conv2.weight'])
    with pytest.raises(RuntimeError):
        m.load_state_dict({'bogus': torch.zeros(1)})
    with pytest.raises(RuntimeError, match='GPU'):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64), torch.tensor([[64., 64., 1.]]))
    with pytest.raises(NotImplementedError):
        m.train()


def test_r50_extension_builds():
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    m = resnet(('__background__', 'Car'), 50)
    m.create_architecture()
    assert len(m.RCNN_layer3[0]) == 6
    sd50 = fixture.make_state_dict(1, layers=fixture.R50)
    m.load_state_dict(sd50)


def test_fold_bn_equals_batch_norm():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 9, 11, generator=g)
    w = torch.randn(24, 16, 3, 3, generator=g) * 0.1
    bn = {'weight': torch.rand(24, generator=g) + 0.5, 'bias': torch.randn(24, generator=g),
          'running_mean': torch.randn(24, generator=g), 'running_var': torch.rand(24, generator=g) + 0.5}
    ref = F.batch_norm(F.conv2d(x, w, None, 1, 1), bn['running_mean'], bn['running_var'], bn['weight'], bn['bias'],
                       False, 0.0, 1e-5)
    wf, bf = engine.fold_bn(w, bn)
    assert float((F.conv2d(x, wf, bf, 1, 1) - ref).abs().max()) < 1e-5


def test_stem_relayout_is_the_7x7_conv():
    """prep_stem + the NHWC4 zero-border packing == Conv2d(3,64,7,2,3): emulate the engine's
    addressing (7 row taps of 32 contiguous floats at padded pixel (2*oh+kh, 2*ow)) with torch on CPU."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, 21, 30, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    bn = {'weight': torch.ones(64), 'bias': torch.zeros(64), 'running_mean': torch.zeros(64),
          'running_var': torch.ones(64) - 1e-5}
    cw = engine.prep_stem(w, bn, device='cpu')
    assert cw.cin == 32 and cw.kh == 7 and cw.kw == 1 and cw.alg_k == 147
    H, W = 21, 30
    packed = torch.zeros(H + 6, W + 8, 4)
    packed[3:3 + H, 3:3 + W, :3] = x[0].permute(1, 2, 0)
    OH, OW = engine.conv_out_hw(H, W, 7, 7, 2, 3)
    flat = packed.reshape(-1)
    wk = cw.weight.view(64, 7 * 32)
    out = torch.zeros(64, OH, OW)
    for oh in range(OH):
        for ow in range(OW):
            a = torch.cat([flat[((2 * oh + kh) * (W + 8) + 2 * ow) * 4:][:32] for kh in range(7)])
            out[:, oh, ow] = wk @ a
    ref = F.conv2d(x, w, None, 2, 3)[0]
    assert float((out - ref).abs().max()) < 1e-4


def test_deconv_relayout():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 8, 3, 3, generator=g)
    w = torch.randn(8, 5, 2, 2, generator=g)
    b = torch.randn(5, generator=g)
    cw = engine.prep_deconv2x2(w, b, device='cpu')
    ref = F.conv_transpose2d(x, w, b, 2)
    rows = cw.weight.view(20, 8) @ x[0].reshape(8, 9)            # (i,j,co) x pixels
    out = torch.zeros(5, 6, 6)
    for ij in range(4):
        for co in range(5):
            out[co, (ij >> 1)::2, (ij & 1)::2] = rows[ij * 5 + co].view(3, 3) + b[co]
    assert float((out - ref[0]).abs().max()) < 1e-5


def test_nms_wrapper_contract_on_cpu():
    from stereo_rcnn_amd.model.nms.nms_wrapper import nms
    assert nms(torch.zeros(0, 5), 0.7) == []                      # nms_wrapper.py:15-16
    with pytest.raises(NotImplementedError):
        nms(torch.zeros(3, 5), 0.7)                               # no CPU branch, like the reference's CUDA op
    with pytest.raises(NotImplementedError):
        nms(torch.zeros(3, 5), 0.7, force_cpu=True)


def test_roi_align_function_needs_gpu():
    from stereo_rcnn_amd.model.roi_align.functions.roi_align import RoIAlignFunction
    with pytest.raises(NotImplementedError):                     # functions/roi_align.py:28-29
        RoIAlignFunction(8, 8, 0.25)(torch.zeros(1, 4, 8, 8), torch.zeros(2, 5))


def test_fixture_is_deterministic():
    a, b = fixture.synthetic_pair(5, 64, 96), fixture.synthetic_pair(5, 64, 96)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and not np.array_equal(a[0], a[1])
    s1, s2 = fixture.make_state_dict(7), fixture.make_state_dict(7)
    assert all(torch.equal(s1[k], s2[k]) for k in s1)
    l, r, info = fixture.make_inputs(3, 375, 1242)
    assert tuple(l.shape) == (1, 3, 600, 1987) and info.tolist() == [[600.0, 1987.0, pytest.approx(1.6)]]


def test_tuned_plans_round_trip(tmp_path):
    """engine.save_plans / load_plans (bench.py --plans): keys and plans survive the JSON file unchanged."""
    saved = dict(engine._TUNED)
    try:
        engine._TUNED.clear()
        key = ('f16x3', 2, 38, 125, 38, 125, 256, 256, 3, 3, 1, 1, 0, 256, 1, 1, 0)
        engine._TUNED[key] = (2, 2, 8, 4, 1)
        engine._TUNED[('f32', 1, 8, 8, 8, 8, 32, 64, 1, 1, 1, 0, 0, 32, 0, 0, 0)] = (1, 1, 4, 2, 3)
        path = str(tmp_path / 'plans.json')
        engine.save_plans(path)
        before = dict(engine._TUNED)
        engine._TUNED.clear()
        assert engine.load_plans(path) == 2
        assert engine._TUNED == before
    finally:
        engine._TUNED.clear()
        engine._TUNED.update(saved)
