"""Chained and grouped launches of the SPLIT16 convolution engine (csrc/conv_chain.hip) against the SAME convolutions launched
one by one with the same tiles: bit-identical, whatever the tile, the tails, the residual / projection-shortcut form.
The separate launches themselves are checked against torch on the CPU in test_ops_gpu.py; the bottleneck they form is
/root/reference/lib/model/stereo_rcnn/resnet.py:82-102, the shared-weight RPN levels stereo_rpn.py:73-95."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bn(g, c):
    return {'weight': torch.rand(c, generator=g) + 0.5, 'bias': torch.randn(c, generator=g) * 0.1,
            'running_mean': torch.randn(c, generator=g) * 0.1, 'running_var': torch.rand(c, generator=g) + 0.5}


def _bits(t):
    return t.view(torch.int32).cpu()


def _bottleneck(dev, P, B, H, W, seed, shortcut=None):
    """Weights + inputs of [conv2 3x3 -> conv3 (+ residual | projection shortcut) -> conv1 of the next block], SPLIT16 inputs."""
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(seed)
    w2 = torch.randn(P, P, 3, 3, generator=g) / (9 * P) ** 0.5
    w3 = torch.randn(4 * P, P, 1, 1, generator=g) / P ** 0.5
    w1 = torch.randn(P, 4 * P, 1, 1, generator=g) / (4 * P) ** 0.5
    c2 = engine.prep_conv(w2, None, 1, 1, True, bn=_bn(g, P), device=dev)
    c1 = engine.prep_conv(w1, None, 1, 0, True, bn=_bn(g, P), device=dev)
    m1 = engine.act_convert(torch.randn(B, H, W, P, generator=g).to(dev), 0, 1)
    if shortcut is None:
        c3 = engine.prep_conv(w3, None, 1, 0, True, bn=_bn(g, 4 * P), device=dev)
        x = engine.act_convert(torch.randn(B, H, W, 4 * P, generator=g).to(dev), 0, 1)
        kw3 = dict(residual=x, res_fmt=1)
    else:
        cin2, s2 = shortcut
        H2, W2 = H * s2 - (s2 - 1), W * s2 - (s2 - 1)
        wd = torch.randn(4 * P, cin2, 1, 1, generator=g) / cin2 ** 0.5
        c3 = engine.prep_conv_shortcut(w3, _bn(g, 4 * P), wd, _bn(g, 4 * P), s2, device=dev)
        x = engine.act_convert(torch.randn(B, H2, W2, cin2, generator=g).to(dev), 0, 1)
        kw3 = dict(x2=x, H2=H2, W2=W2)
    return c2, c3, c1, m1, kw3


def _run_both(dev, P, B, H, W, tile, nphase, seed, shortcut=None):
    from stereo_rcnn_amd import engine
    mr, waves, stages, na, nb = tile
    c2, c3, c1, m1, kw3 = _bottleneck(dev, P, B, H, W, seed, shortcut)
    e = lambda c: torch.zeros((B, H, W, c), device=dev)
    nr = lambda cw: nb if cw.cout >= 64 * nb else na
    common = dict(precision='f16x3', x_fmt=1, y_fmt=1)
    # one by one
    m2a, xa, m1a = e(P), e(4 * P), e(P)
    engine.conv2d(c2, m1, B, H, W, m2a, H, W, plan=(mr, nr(c2), waves, stages, 1), **common)
    if nphase > 1:
        engine.conv2d(c3, m2a, B, H, W, xa, H, W, plan=(mr, nr(c3), waves, stages, 1), **common, **kw3)
    if nphase > 2:
        engine.conv2d(c1, xa, B, H, W, m1a, H, W, plan=(mr, nr(c1), waves, stages, 1), **common)
    # chained
    m2b, xb, m1b = e(P), e(4 * P), e(P)
    ph = [((c2, m1, B, H, W, m2b, H, W), dict(common))]
    if nphase > 1:
        ph.append(((c3, m2b, B, H, W, xb, H, W), dict(common, **kw3)))
    if nphase > 2:
        ph.append(((c1, xb, B, H, W, m1b, H, W), dict(common)))
    engine.conv_chain(ph, tile, name='test.chain')
    torch.cuda.synchronize()
    assert float(m2a.abs().max()) > 0
    assert torch.equal(_bits(m2a), _bits(m2b)), "conv2"
    if nphase > 1:
        assert float(xa.abs().max()) > 0
        assert torch.equal(_bits(xa), _bits(xb)), "conv3"
    if nphase > 2:
        assert float(m1a.abs().max()) > 0
        assert torch.equal(_bits(m1a), _bits(m1b)), "next conv1"


TILES = [(2, 4, 2, 1, 2), (2, 4, 2, 1, 1), (2, 4, 2, 2, 2), (2, 8, 2, 2, 2), (2, 8, 4, 2, 2), (4, 8, 3, 2, 2), (4, 8, 2, 4, 4)]


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("P", [64, 128, 256])
def test_bottleneck_chain_equals_separate_launches_bit_for_bit(dev, tile, P):
    """Every instantiated chain tile on every bottleneck width it can carry: ragged M (2 x 19 x 31 = 1178 rows: an image boundary
    inside a tile, a partial last tile), identity residual."""
    if 64 * tile[3] > P:
        pytest.skip("narrow tile wider than the bottleneck")
    _run_both(dev, P, 2, 19, 31, tile, 3, seed=P + tile[0] * 7 + tile[1])


@pytest.mark.parametrize("tile", [(2, 4, 2, 1, 2), (2, 8, 2, 2, 2), (4, 8, 3, 2, 2)])
@pytest.mark.parametrize("shortcut", [(64, 1), (256, 2)])
def test_chain_with_projection_shortcut_in_the_middle_phase(dev, tile, shortcut):
    """The first block of a layer: conv3 takes the block input as a second, K-concatenated operand (srcnn_conv_desc.x2), at the
    same or at twice the resolution (odd sizes)."""
    P = 64 if tile[3] == 1 else 128
    _run_both(dev, P, 2, 12, 21, tile, 3, seed=3, shortcut=shortcut)


@pytest.mark.parametrize("nphase", [1, 2])
def test_shorter_chains(dev, nphase):
    """The last block of a layer has no next conv1: two phases; one phase = a plain launch through the chain kernel."""
    _run_both(dev, 128, 1, 17, 23, (2, 8, 2, 2, 2), nphase, seed=nphase)


def test_chain_many_workgroups_in_several_rounds(dev):
    """A launch with more workgroups than the chip holds at once (2 x 75 x 249 = 37350 rows = 292 workgroups of 128 rows, two
    per CU at most): workgroups of different rounds are in different phases at the same time; the tensor the first phase
    reads (halo rows of the neighbours) is never written, the double-buffered m1 takes the last phase's output."""
    _run_both(dev, 64, 2, 75, 249, (2, 4, 2, 1, 2), 3, seed=9)
    _run_both(dev, 128, 2, 75, 249, (2, 8, 2, 2, 2), 3, seed=10)


def test_chain_refuses_what_it_cannot_run(dev):
    from stereo_rcnn_amd import engine, _lib
    assert _lib.lib().srcnn_conv2d_chain_supported(2, 8, 2, 2, 2) == 1
    assert _lib.lib().srcnn_conv2d_chain_supported(1, 4, 2, 1, 1) == 0
    P, B, H, W = 64, 1, 9, 11
    c2, c3, c1, m1, kw3 = _bottleneck(dev, P, B, H, W, 1)
    common = dict(precision='f16x3', x_fmt=1, y_fmt=1)
    e = lambda c: torch.zeros((B, H, W, c), device=dev)
    m2, x, _ = e(P), e(4 * P), e(P)
    # the last phase would overwrite the tensor the first phase reads
    with pytest.raises(RuntimeError, match="overwrite"):
        engine.conv_chain([((c2, m1, B, H, W, m2, H, W), dict(common)), ((c3, m2, B, H, W, x, H, W), dict(common, **kw3)),
                           ((c1, x, B, H, W, m1, H, W), dict(common))], (2, 4, 2, 1, 2))
    # a tile that is not instantiated is an error, never a silent fallback
    with pytest.raises(RuntimeError, match="tile configuration"):
        engine.conv_chain([((c2, m1, B, H, W, m2, H, W), dict(common))], (1, 4, 2, 1, 1))
    # phase 1 must read what phase 0 wrote
    with pytest.raises(RuntimeError, match="reads exactly"):
        engine.conv_chain([((c2, m1, B, H, W, m2, H, W), dict(common)), ((c3, e(P), B, H, W, x, H, W), dict(common, **kw3))], (2, 4, 2, 1, 2))


@pytest.mark.parametrize("tile", [(4, 4, 8, 2), (2, 2, 8, 2)])
def test_grouped_rpn_levels_equal_their_own_launches_bit_for_bit(dev, tile):
    """The stereo RPN conv + fused head over three pyramid levels with shared weights (stereo_rpn.py:73-95) as ONE launch: every
    level's partial planes are the bits its own launch writes."""
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(4)
    B = 2                                    # one stereo pair: left, right
    w = torch.randn(512, 256, 3, 3, generator=g) / 48.0
    cw = engine.prep_conv(w, torch.randn(512, generator=g) * 0.1, 1, 1, True, device=dev)
    pair = engine.ConvW(cw.weight, cw.bias, 3, 3, 1, 1, True, mode=2)
    hw = engine.prep_conv(torch.randn(24, 1024, 1, 1, generator=g) / 32.0, torch.randn(24, generator=g), 1, 0, False, device=dev)
    levels = [(19, 63), (10, 32), (5, 16)]
    parts = 8                                # planes are sized for the 128-column tile (2 eyes x 512 / 128); the 256-column tile fills 4
    xs = [engine.act_convert(torch.randn(B, h, w_, 256, generator=g).to(dev), 0, 1) for h, w_ in levels]
    one = [torch.zeros((parts, (B // 2) * h * w_, 24), device=dev) for h, w_ in levels]
    grp = [torch.zeros_like(t) for t in one]
    for (h, w_), x, y in zip(levels, xs, one):
        engine.conv2d(pair, x, B, h, w_, None, h, w_, precision='f16x3', x_fmt=1, head2=(hw, y, parts), plan=tile + (1,))
    engine.conv_group([((pair, x, B, h, w_, None, h, w_), dict(precision='f16x3', x_fmt=1, head2=(hw, y, parts)))
                       for (h, w_), x, y in zip(levels, xs, grp)], tile, name='test.group')
    torch.cuda.synchronize()
    for a, b in zip(one, grp):
        assert float(a.abs().max()) > 0
        assert torch.equal(_bits(a), _bits(b))


def test_grouped_plain_convolutions(dev):
    """Grouped launch without a head: three 3x3 convolutions of different sizes, SPLIT16 out."""
    from stereo_rcnn_amd import engine
    g = torch.Generator().manual_seed(6)
    cw = engine.prep_conv(torch.randn(256, 256, 3, 3, generator=g) / 48.0, torch.randn(256, generator=g) * 0.1, 1, 1, False, device=dev)
    levels = [(38, 125), (19, 63), (10, 32)]
    xs = [engine.act_convert(torch.randn(2, h, w_, 256, generator=g).to(dev), 0, 1) for h, w_ in levels]
    for tile in [(4, 2, 8, 3), (2, 2, 8, 2)]:
        one = [torch.zeros((2, h, w_, 256), device=dev) for h, w_ in levels]
        grp = [torch.zeros_like(t) for t in one]
        for (h, w_), x, y in zip(levels, xs, one):
            engine.conv2d(cw, x, 2, h, w_, y, h, w_, precision='f16x3', x_fmt=1, y_fmt=1, plan=tile + (1,))
        engine.conv_group([((cw, x, 2, h, w_, y, h, w_), dict(precision='f16x3', x_fmt=1, y_fmt=1)) for (h, w_), x, y in zip(levels, xs, grp)],
                          tile)
        torch.cuda.synchronize()
        for a, b in zip(one, grp):
            assert torch.equal(_bits(a), _bits(b)), tile


def test_forward_with_chained_bottlenecks_equals_the_unchained_forward(dev):
    """The whole network with layer1-3 on chained launches (SRCNN_BOTTLENECK_CHAIN=1) against the same forward unchained: the
    chained trunk runs other tiles than the tuned single launches (another summation order: the engine's plan-to-plan rounding),
    so proposals agree to 1e-3 px and the regressions to 2e-5 -- and the chained forward is bit-repeatable."""
    from stereo_rcnn_amd import engine, fixture
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    m = resnet(('__background__', 'Car'), 101, pretrained=False)
    m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3))
    m.cuda().eval()
    m.precision = 'f16x3'
    l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 120, 400, target_short=192)]
    saved = engine.BOTTLENECK_CHAIN, engine.CHAIN_MIN_WGS
    try:
        with torch.no_grad():
            engine.BOTTLENECK_CHAIN = '0'
            ref = [t.clone() for t in m(l, r, info)[:8]]
            engine.BOTTLENECK_CHAIN, engine.CHAIN_MIN_WGS = '1', 1
            engine.PLAN_EPOCH += 1
            engine.FlopCounter.enabled, engine.FlopCounter.rows = True, []
            a = [t.clone() for t in m(l, r, info)[:8]]
            rows, engine.FlopCounter.enabled, engine.FlopCounter.rows = engine.FlopCounter.rows, False, None
            b = [t.clone() for t in m(l, r, info)[:8]]
        torch.cuda.synchronize()
    finally:
        engine.BOTTLENECK_CHAIN, engine.CHAIN_MIN_WGS = saved
        engine.PLAN_EPOCH += 1
    chains = [x for x in rows if 'chain' in x]
    assert len(chains) == 3 + 4 + 23, len(chains)               # every block of layer1-3 is one launch
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    d = (ref[0][0][:, None, 1:] - a[0][0][None, :, 1:]).abs().amax(2)      # proposals: matched by position (a tie may swap two)
    best, idx = d.min(1)
    ok = best < 1e-3
    assert float(ok.float().mean()) > 0.98, float(ok.float().mean())
    for i in (2, 3, 4):                                         # cls_prob, bbox_pred, dim_orien_pred of the matched rois
        assert float((ref[i][0][ok] - a[i][0][idx[ok]]).abs().max()) < 2e-5
