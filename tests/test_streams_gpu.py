"""Stream placement (stereo_rcnn_amd/streams.py, include/srcnn_hip.h: srcnn_stream_create*, srcnn_probe_placement)."""
import collections

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _probe(stream, dev, blocks=2048):
    from stereo_rcnn_amd import _lib
    xcc = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    hw = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().srcnn_probe_placement(blocks, xcc.data_ptr(), hw.data_ptr(), stream.cuda_stream), 'srcnn_probe_placement')
    stream.synchronize()
    return xcc.cpu().tolist(), hw.cpu().tolist()


def test_dedicated_stream_runs_kernels_on_every_xcd(dev):
    from stereo_rcnn_amd import streams
    s = streams.new_stream('dedicated')
    xcc, hw = _probe(s, dev)
    per = collections.Counter(xcc)
    assert sorted(per) == list(range(8)) and set(per.values()) == {256}     # block b on XCD (b + r) % 8
    # (r = where the dispatcher's round-robin stood: 0 on most boxes / histories, not guaranteed -- the conv engine's tile map only
    #  needs blocks b and b + 8 to meet on one XCD, csrc/conv_f16s.hip)
    assert all(x == (b + xcc[0]) % 8 for b, x in enumerate(xcc))
    # ordinary torch work on the wrapped stream
    with torch.cuda.stream(s):
        a = torch.arange(1024, device=dev, dtype=torch.float32)
        b = (a * 2).sum()
    s.synchronize()
    assert float(b) == 1024 * 1023


def test_cu_mask_partition_is_one_shader_engine_of_every_xcd(dev):
    """partition k of 4: mask bits b with (b >> 3) & 3 == k -- bit b is CU b // 8 of XCD b % 8, CU i of an XCD sits in shader
    engine i % 4, so the partition is shader engine k of every XCD: 8 CUs per XCD, 64 in all, and every XCD still gets its
    share of the blocks (block b -> XCD b % 8 whatever the mask)."""
    from stereo_rcnn_amd import streams
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    masks = streams.partition_masks(4, n_cus)
    for k in (1, 2):
        xcc, hw = _probe(streams.masked_stream(masks[k]), dev)
        assert all(x == (b + xcc[0]) % 8 for b, x in enumerate(xcc))
        cus = collections.defaultdict(set)
        for x, h in zip(xcc, hw):
            cus[x].add(((h >> 13) & 7, (h >> 8) & 15))          # (shader engine, CU)
        assert sorted(cus) == list(range(8))
        for x in cus:
            assert len(cus[x]) == n_cus // 32 and {se for se, _ in cus[x]} == {k}, (x, sorted(cus[x]))


def test_forward_is_bit_identical_with_and_without_branch_streams(dev):
    """One forward in flight: the plan forks FPN laterals / small RPN levels / the box head onto side streams; several in flight:
    every launch stays on the forward's main stream.  Same launches either way -> bit-identical outputs; one recorded launch
    program per regime (the multi-in-flight one has no fork / join nodes)."""
    from stereo_rcnn_amd import _lib, fixture, streams
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    m = resnet(('__background__', 'Car'), 101, pretrained=False)
    m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3))
    m.cuda().eval()
    m.precision = 'f16x3'
    m.use_program = True
    l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 120, 400, target_short=192)]
    outs, sizes = {}, {}
    prev = streams.pairs_in_flight()
    try:
        for n in (1, 4, 1):
            streams.set_pairs_in_flight(n)
            with torch.no_grad():
                m(l, r, info)
                o = [t.clone() for t in m(l, r, info)[:8]]
            torch.cuda.synchronize()
            plan = m._get_plan(1, l.shape[2], l.shape[3])
            assert plan._par() == (n == 1)
            sizes[n] = _lib.lib().srcnn_program_size(plan.programs[plan.program_key('f16x3', True, par=(n == 1))][0])
            if n in outs:
                for a, b in zip(outs[n], o):
                    assert torch.equal(a, b)
            outs[n] = o
    finally:
        streams.set_pairs_in_flight(prev)
    for a, b in zip(outs[1], outs[4]):
        assert torch.equal(a, b)
    assert sizes[1] > sizes[4] > 150, sizes                     # the fork / join event nodes are gone


def test_four_pairs_in_flight_on_pooled_streams_equal_one_at_a_time(dev):
    from stereo_rcnn_amd import fixture, streams, tune
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    m = resnet(('__background__', 'Car'), 101, pretrained=False)
    m.create_architecture()
    m.load_state_dict(fixture.make_state_dict(3))
    m.cuda().eval()
    m.precision = 'f16x3'
    m.use_program = True
    l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 120, 400, target_short=192)]
    prev = streams.pairs_in_flight()
    try:
        with torch.no_grad():
            streams.set_pairs_in_flight(1)
            ref = [t.clone() for t in m(l, r, info)[:8]]
            torch.cuda.synchronize()
            ss = streams.main_streams(4)
            streams.set_pairs_in_flight(4)
            res = []
            for rep in range(3):
                for k, s in enumerate(ss):
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        res.append([t.clone() for t in m(l, r, info, slot=k)[:8]])
            torch.cuda.synchronize()
    finally:
        streams.set_pairs_in_flight(prev)
    for o in res:
        for a, b in zip(ref, o):
            assert torch.equal(a, b)
