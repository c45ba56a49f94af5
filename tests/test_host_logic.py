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
        # the keys of row-limited launches and of convs with a second operand, as engine.conv2d builds them: flat tuples
        cw = engine.ConvW(torch.zeros(256, 1, 1, 64 + 64), None, 1, 1, 1, 0, True, cin2=64, stride2=2)
        assert (cw.cin, cw.cin2, cw.alg_k) == (64, 64, 128)
        k2 = engine._shape_key(cw, 2, 38, 125, 38, 125, 64, 'f16x3', (1, 1, 0) + ('lim', 196) + ('x2', cw.cin2, cw.stride2, 75, 249))
        assert all(not isinstance(v, (tuple, list)) for v in k2)
        engine._TUNED[k2] = (2, 1, 4, 2, 1)
        path = str(tmp_path / 'plans.json')
        engine.save_plans(path)
        before = dict(engine._TUNED)
        engine._TUNED.clear()
        assert engine.load_plans(path) == 3
        assert engine._TUNED == before
    finally:
        engine._TUNED.clear()
        engine._TUNED.update(saved)


def test_prep_conv_shortcut_folds_and_concatenates():
    """engine.prep_conv_shortcut (host side of srcnn_conv_desc.x2): relu(bn3(conv3(t)) + bn_d(downsample(x))) equals ONE 1x1 GEMM
    over [channels of t | channels of x sampled with the shortcut's stride] with the concatenated folded weights and the summed
    folded biases (resnet.py:86-100) -- checked with torch on the CPU."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(4)
    cin, cin2, cout, s2 = 32, 64, 96, 2
    t, x = torch.randn(2, cin, 5, 7, generator=g), torch.randn(2, cin2, 9, 13, generator=g)
    w3, wd = torch.randn(cout, cin, 1, 1, generator=g), torch.randn(cout, cin2, 1, 1, generator=g)
    bn = lambda: {'weight': torch.rand(cout, generator=g) + 0.5, 'bias': torch.randn(cout, generator=g),
                  'running_mean': torch.randn(cout, generator=g) * 0.1, 'running_var': torch.rand(cout, generator=g) + 0.5}
    bn3, bnd = bn(), bn()
    fbn = lambda v, b: F.batch_norm(v, b['running_mean'], b['running_var'], b['weight'], b['bias'], False, 0.0, 1e-5)
    ref = F.relu(fbn(F.conv2d(t, w3), bn3) + fbn(F.conv2d(x, wd, None, s2), bnd))
    cw = engine.prep_conv_shortcut(w3, bn3, wd, bnd, s2, device='cpu')
    assert (cw.cin, cw.cin2, cw.stride2, cw.cout, cw.relu, cw.alg_k) == (cin, cin2, s2, cout, 1, cin + cin2)
    both = torch.cat((t, x[:, :, ::s2, ::s2]), 1)                          # the K-concatenated operand
    got = F.relu(F.conv2d(both, cw.weight.view(cout, cin + cin2, 1, 1), cw.bias))
    assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())


def test_layer_table_bounds_grouping_and_ranking():
    """stereo_rcnn_amd/layer_table.py (bench.py's roofline.layers): own bound = max(issued MFMA flops / peak, bytes / HBM rate),
    launches of one layer shape pooled, groups ranked by the time they lose -- on hand-made rows, no GPU."""
    from stereo_rcnn_amd import layer_table as lt
    rows = []
    for i in range(3):          # an MFMA-bound layer, three blocks: 10 GFLOP, 10 MB each, 100 us measured
        rows.append({'name': 'layer3.%d.conv2' % i, 'M': 9500, 'N': 256, 'K': 2304, 'flops': 10e9, 'bytes': 10e6, 'plan': (2, 2, 8, 4, 1),
                     'us': 100.0, 'wgs': 150})
    rows.append({'name': 'layer1.0.conv3', 'M': 149100, 'N': 256, 'K': 64, 'flops': 1e9, 'bytes': 630e6, 'plan': (2, 2, 8, 2, 1),
                 'us': 125.0, 'wgs': 2330})
    for r in rows:              # what measure() derives per row
        r['mfma_us'] = 3.0 * r['flops'] / 2.5e15 * 1e6
        r['hbm_us'] = r['bytes'] / 6.3e12 * 1e6
        r['bound_us'] = max(r['mfma_us'], r['hbm_us'])
        r['bound'] = 'mfma' if r['mfma_us'] >= r['hbm_us'] else 'hbm'
        r['frac_of_own_bound'] = r['bound_us'] / r['us']
        r['lost_us'] = r['us'] - r['bound_us']
        r['tflops'] = r['flops'] / r['us'] / 1e6
    assert abs(rows[0]['bound_us'] - 12.0) < 1e-9 and rows[0]['bound'] == 'mfma'
    assert abs(rows[3]['bound_us'] - 100.0) < 1e-9 and rows[3]['bound'] == 'hbm'
    g = lt.grouped(rows)
    assert [x['name'] for x in g] == ['layer3.*.conv2', 'layer1.*.conv3']          # 3 x 88 us lost ranks above 25 us lost
    assert g[0]['launches'] == 3 and abs(g[0]['us'] - 300.0) < 1e-9 and abs(g[0]['lost_us'] - 264.0) < 1e-9
    s = lt.summary(rows)
    assert s['launches'] == 4 and s['mfma_bound_launches'] == 3 and s['hbm_bound_launches'] == 1
    assert abs(s['frac_of_own_bounds'] - (36.0 + 100.0) / 425.0) < 1e-3
    top = lt.top_for_json(rows, 1)
    assert len(top) == 1 and top[0]['layer'] == 'layer3.*.conv2' and top[0]['workgroups'] == 150
    txt = lt.format_table(rows, 'title')
    assert 'layer3.*.conv2' in txt and 'every launch, in launch order' in txt


def test_rank_host_budget_and_numa_pinning(tmp_path, monkeypatch):
    """distributed.py: a rank's share of the host cores for its Newton-CG threads and the CPU list of its GPU's NUMA node read
    from sysfs (here: a fake tree)."""
    import os
    from stereo_rcnn_amd import distributed as sdist
    assert sdist._parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    ncpu = len(os.sched_getaffinity(0))
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert sdist.local_world_size() == 8 and sdist.host_solver_threads() == max(1, min(16, ncpu // 8))
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    assert sdist.host_solver_threads(4) == min(4, ncpu)
    # fake sysfs: GPU at 0000:c1:00.0 on NUMA node 1 whose cpulist is this process's own first CPU
    mine = sorted(os.sched_getaffinity(0))
    dev = tmp_path / 'bus' / 'pci' / 'devices' / '0000:c1:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    node = tmp_path / 'devices' / 'system' / 'node' / 'node1'
    node.mkdir(parents=True)
    (node / 'cpulist').write_text('%d\n' % mine[0])

    class Props(object):
        pci_domain_id, pci_bus_id, pci_device_id = 0, 0xc1, 0
    import torch
    monkeypatch.setattr(torch.cuda, 'get_device_properties', lambda i: Props())
    assert sdist.gpu_numa_cpus(0, sysfs=str(tmp_path)) == (1, {mine[0]})
    try:
        got = sdist.pin_to_gpu_numa(0, cpus={mine[0]})
        assert got == {'numa_node': None, 'cpus': 1} and os.sched_getaffinity(0) == {mine[0]}
        assert sdist.pin_to_gpu_numa(0, cpus={10 ** 6}) is None           # nothing of the mask left: unchanged
    finally:
        os.sched_setaffinity(0, mine)


def test_partition_masks_split_every_xcd_evenly():
    """streams.partition_masks: bit b of a queue's CU mask = CU b // 8 of XCD b % 8 (profiles/queue_mapping_r04.txt); every
    partition must hold the same number of CUs of EVERY XCD (a dispatch's blocks are dealt to all XCDs whatever the mask)."""
    from stereo_rcnn_amd import streams
    for parts in (1, 2, 3, 4, 8):
        masks = streams.partition_masks(parts, 256)
        assert len(masks) == parts and all(len(m) == 8 for m in masks)
        seen = 0
        for m in masks:
            bits = [b for b in range(256) if m[b >> 5] >> (b & 31) & 1]
            v = sum(1 << b for b in bits)
            assert seen & v == 0
            seen |= v
            per_xcd = [sum(1 for b in bits if b % 8 == x) for x in range(8)]
            assert len(set(per_xcd)) == 1 and per_xcd[0] >= 32 // parts - 1 and per_xcd[0] >= 1, (parts, per_xcd)
        assert seen == (1 << 256) - 1
    for parts in (2, 4):        # these use mask bits 3-4 only: an even split under an XCD-major enumeration as well
        for m in streams.partition_masks(parts, 256):
            assert len(set(m)) == 1


def test_branch_streams_only_while_one_forward_is_in_flight(monkeypatch):
    from stereo_rcnn_amd import streams
    prev = streams.pairs_in_flight()
    try:
        monkeypatch.setattr(streams, 'SIDE_KIND', 'auto')
        streams.set_pairs_in_flight(1)
        assert streams.branch_overlap()
        streams.set_pairs_in_flight(3)
        assert not streams.branch_overlap() and streams.pairs_in_flight() == 3
        monkeypatch.setattr(streams, 'SIDE_KIND', 'none')
        streams.set_pairs_in_flight(1)
        assert not streams.branch_overlap()
        monkeypatch.setattr(streams, 'SIDE_KIND', 'pool')
        streams.set_pairs_in_flight(4)
        assert streams.branch_overlap()
    finally:
        streams.set_pairs_in_flight(prev)


def test_ensure_hw_queues_sets_the_runtime_variable_before_hip_starts(monkeypatch):
    import torch
    from stereo_rcnn_amd import streams
    monkeypatch.setattr(torch.cuda, 'is_initialized', lambda: False)
    monkeypatch.setattr(streams, '_queues_at_hip_start', None)
    monkeypatch.delenv('GPU_MAX_HW_QUEUES', raising=False)
    monkeypatch.delenv('SRCNN_KEEP_HW_QUEUES', raising=False)
    assert streams.max_pairs_in_flight() == 3                    # HIP's default: 4 queues, one is the null stream's
    assert streams.ensure_hw_queues(8) and streams.max_pairs_in_flight() == 7
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '16')
    assert streams.ensure_hw_queues(8) and streams.max_pairs_in_flight() == 15      # a larger user setting is kept
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '2')
    monkeypatch.setenv('SRCNN_KEEP_HW_QUEUES', '1')
    assert not streams.ensure_hw_queues(8) and streams.max_pairs_in_flight() == 1   # a smaller one too, when marked deliberate
    monkeypatch.delenv('SRCNN_KEEP_HW_QUEUES')
    assert streams.ensure_hw_queues(8) and streams.max_pairs_in_flight() == 7       # ... otherwise it is raised
    monkeypatch.setattr(torch.cuda, 'is_initialized', lambda: True)
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    monkeypatch.setattr(streams, '_queues_at_hip_start', None)
    assert not streams.ensure_hw_queues(8) and streams.max_pairs_in_flight() == 3    # too late: HIP is up
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')                                     # changing the variable after HIP started changes
    assert not streams.ensure_hw_queues(8) and streams.max_pairs_in_flight() == 3    # nothing: the value HIP READ is what counts


def test_queue_supply_warning(monkeypatch, caplog):
    import logging
    from stereo_rcnn_amd import streams
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    monkeypatch.setattr(streams, '_warned', set())
    monkeypatch.setattr(streams, '_queues_at_hip_start', None)
    with caplog.at_level(logging.WARNING, logger='stereo_rcnn_amd'):
        assert streams.check_queue_supply(3, 'pool') and not caplog.records
        assert not streams.check_queue_supply(4, 'pool') and len(caplog.records) == 1       # 4 in flight + null stream > 4 queues
        assert not streams.check_queue_supply(4, 'pool')
        assert len(caplog.records) == 1                                                     # once per count
        assert streams.check_queue_supply(4, 'dedicated')
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')
    assert streams.check_queue_supply(7, 'pool')


def test_throughput_tuner_descends_on_the_measured_step(monkeypatch):
    """tune.tune_throughput with a model of the machine in place of the GPU: three shape keys, the step time is the sum of the
    chosen plans' costs IN THE MIX (which differ from the isolated times the in-situ tuner logged).  The descent must move each
    key whose mix-cost improves by more than the threshold, keep the others, and leave engine._TUNED at the result."""
    from stereo_rcnn_amd import engine, tune

    class Model(object):
        use_program = True

    keys = [('f16x3', 1, 38, 125, 38, 125, 256, 256 + i, 3, 3, 1, 1, 0, 256) for i in range(3)]      # engine._shape_key layout
    iso = {keys[0]: [((2, 2, 8, 2, 3), 0.050), ((2, 2, 8, 2, 1), 0.056), ((4, 4, 8, 2, 1), 0.070)],
           keys[1]: [((2, 1, 4, 2, 1), 0.030), ((2, 2, 8, 2, 1), 0.031)],
           keys[2]: [((1, 1, 4, 2, 1), 0.020), ((1, 2, 4, 2, 1), 0.021)]}
    mix = {keys[0]: {(2, 2, 8, 2, 3): 1.20, (2, 2, 8, 2, 1): 1.00, (4, 4, 8, 2, 1): 1.50},      # the split plan loses in the mix
           keys[1]: {(2, 1, 4, 2, 1): 0.70, (2, 2, 8, 2, 1): 0.699},                            # a gain below the threshold
           keys[2]: {(1, 1, 4, 2, 1): 0.40, (1, 2, 4, 2, 1): 0.45}}
    hits = {keys[0]: 23, keys[1]: 22, keys[2]: 1}
    monkeypatch.setattr(engine, '_TUNED', {k: v[0][0] for k, v in iso.items()})
    monkeypatch.setattr(engine, '_TUNE_LOG', {k: list(v) for k, v in iso.items()})

    class Runner(object):
        calls = 0

        def step(self, slot):
            for k, n in hits.items():
                engine.KEY_HITS[k] = n

        def measure(self, steps=24, repeats=3):
            Runner.calls += 1
            return 5.0 + sum(mix[k][engine._TUNED[k]] for k in keys)

    log = []
    base, final, changes = tune.tune_throughput(Model(), None, None, None, streams=4, min_gain=0.004, rounds=2, log=log.append,
                                                runner=Runner())
    assert abs(base - 7.30) < 1e-9 and abs(final - 7.10) < 1e-9
    assert [(c[0], c[1], c[2]) for c in changes] == [(keys[0], (2, 2, 8, 2, 3), (2, 2, 8, 2, 1))]
    assert engine._TUNED[keys[0]] == (2, 2, 8, 2, 1) and engine._TUNED[keys[1]] == (2, 1, 4, 2, 1) and engine._TUNED[keys[2]] == (1, 1, 4, 2, 1)
    assert any('round 2: 0 plans changed' in ln for ln in log)


def test_serving_regime_gates_the_shipped_plans(monkeypatch):
    """serving.enter: one forward at a time never adopts the throughput-tuned plan file; several in flight adopt it once, only on
    the GPU model it was tuned on, and never when switched off (the product runs what bench.py runs: VERDICT r4 item 4)."""
    from stereo_rcnn_amd import engine, serving, streams
    monkeypatch.setattr(serving, '_loaded', {})
    monkeypatch.setattr(engine, '_TUNED', {})
    prev = streams.pairs_in_flight()
    try:
        assert serving.enter(1)['shipped_plans'] == 0 and not engine._TUNED and streams.pairs_in_flight() == 1
        monkeypatch.setattr(serving, 'device_matches', lambda name='mi355x.json', device=None: False)
        assert serving.enter(4)['shipped_plans'] == 0 and not engine._TUNED            # another GPU model: in-situ tuner only
        serving.drop_shipped_plans()
        monkeypatch.setattr(serving, 'device_matches', lambda name='mi355x.json', device=None: True)
        info = serving.enter(4)
        assert info['shipped_plans'] == info['shipped_plans_adopted_now'] == len(engine._TUNED) > 50 and not info['branch_side_streams']
        assert all(k[1] in (1, 2, 300) for k in engine._TUNED)      # shapes of batch-1 pairs only: 2 images, 1 RPN map, 300 rois
        epoch = engine.PLAN_EPOCH
        assert serving.enter(3)['shipped_plans_adopted_now'] == 0 and engine.PLAN_EPOCH == epoch      # once per process
        serving.drop_shipped_plans()
        monkeypatch.setattr(serving, 'USE_SHIPPED_PLANS', False)
        monkeypatch.setattr(engine, '_TUNED', {})
        assert serving.enter(4)['shipped_plans'] == 0 and not engine._TUNED
    finally:
        streams.set_pairs_in_flight(prev)


def test_no_kernel_carries_a_device_scope_fence(tmp_path):
    """a `__threadfence()` (agent-scope release = `buffer_wbl2`: the XCD's whole L2 written back) in a kernel that runs
    beside other forwards' launches cost the four-in-flight mix 116 us per forward for a pass that took 25 us alone.  Grid-wide
    hand-offs in this library go through agent-scope atomics or a kernel boundary: no kernel source calls a fence, and the ISA of
    the radix-select passes (the one kernel with such a hand-off) has no L2 write-back / invalidate instruction."""
    import glob
    import os
    import re
    import shutil
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo_rcnn_amd', 'csrc')
    srcs = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')))
    assert len(srcs) >= 10
    fence = re.compile(r'__threadfence(_system|_block)?\s*\(|__atomic_thread_fence|__builtin_amdgcn_fence|atomic_thread_fence')
    for f in srcs:
        code = '\n'.join(l.split('//')[0] for l in open(f).read().split('\n'))          # comments may talk about fences
        assert not fence.search(code), 'device-scope fence in %s' % os.path.basename(f)
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not available: source check only')
    out = str(tmp_path / 'rpn.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only', '-o', out,
                           os.path.join(csrc, 'rpn_proposal.hip')], stderr=subprocess.DEVNULL)
    isa = open(out).read()
    assert 'tk_hist_kernel' in isa and 'global_atomic_add' in isa
    assert 'buffer_wbl2' not in isa and 'buffer_inv' not in isa
