"""The product runs what the benchmark runs: `serving.enter` -- shipped throughput-tuned conv plans
(stereo_rcnn_amd/plans/mi355x.json), several forwards in flight each on its own stream / hardware queue, launch programs -- and
in THAT regime the assembled forward still meets the reference-code goldens: full-size synthetic pair and the demo pair,
regressions within the north star's 1e-4, heads fed the reference's own rois within 1e-4 on every output."""
import json
import os

import numpy as np
import pytest
import torch

import tolerances as tol_

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _shipped_keys():
    from stereo_rcnn_amd import serving
    with open(serving.shipped_plans_path()) as f:
        return {tuple(k): tuple(v) for k, v in json.load(f)}


def _in_flight(m, inputs, n_slots, rounds):
    """`rounds` x n_slots forwards round-robin on the serving regime's own streams; returns every forward's 8 outputs."""
    from stereo_rcnn_amd import pipeline
    streams = pipeline._slot_streams(n_slots)           # enters the serving regime exactly as detect_3d_stream does
    outs = []
    with torch.no_grad():
        for k in range(rounds * n_slots):
            s = streams[k % n_slots]
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                o = inputs(m, k % n_slots)
                outs.append([t.clone() for t in o[:8]])
    torch.cuda.synchronize()
    return outs


def test_serving_regime_adopts_the_shipped_plans_on_this_gpu(dev):
    from stereo_rcnn_amd import engine, serving, streams
    prev = streams.pairs_in_flight()
    try:
        serving.drop_shipped_plans()
        info = serving.enter(4)
        assert serving.device_matches(), 'the GPU tests run on an MI355X (gfx950, 256 CUs)'
        assert info['pairs_in_flight'] == 4 and info['shipped_plans'] > 50 and not info['branch_side_streams']
        want = _shipped_keys()
        assert all(engine._TUNED.get(k) == v for k, v in want.items())
        assert serving.enter(4)['shipped_plans_adopted_now'] == 0           # once per process
        assert serving.enter(1)['branch_side_streams'] is True              # alone: latency regime, branches on side streams
    finally:
        streams.set_pairs_in_flight(prev)


def test_full_size_golden_four_in_flight_on_the_shipped_plans(dev):
    """BASELINE configs[1]'s frame (375x1242 -> 600x1987), default engine, launch programs, FOUR forwards in flight on the
    shipped plans -- the regime `value` is measured in -- against the reference code's own outputs."""
    from stereo_rcnn_amd import engine, fixture, serving, streams
    from test_model_gpu import HEAD_OUTS, _build_model, _check_end_to_end, _heads_on_reference_rois
    g = np.load(os.path.join(GOLD, 'reference_net_full_r101_seed3.npz'))
    seed, h, w, short = [int(v) for v in g['spec']]
    m, _ = _build_model(dev)
    m.precision, m.use_program = 'f16x3', True
    l, r, info = [t.to(dev) for t in fixture.make_inputs(seed, h, w, target_short=short)]
    prev = streams.pairs_in_flight()
    try:
        engine.KEY_HITS = {}
        outs = _in_flight(m, lambda mdl, slot: mdl(l, r, info, slot=slot), 4, 2)
        hits, engine.KEY_HITS = engine.KEY_HITS, None
        assert serving.plans_loaded() > 50
        used = set(hits) & set(_shipped_keys())
        assert len(used) >= 40, 'the forward must run on the shipped plans (%d of its %d shape keys do)' % (len(used), len(hits))
        for key in used:
            assert engine._TUNED[key] == _shipped_keys()[key]
        ref_out = {k: torch.from_numpy(g[k]) for k in HEAD_OUTS}
        rl, rr = torch.from_numpy(g['rois_left'])[0], torch.from_numpy(g['rois_right'])[0]
        worst = {}
        for k, out in enumerate(outs):
            frac, errs = _check_end_to_end(out, rl, rr, ref_out, 0.97)
            assert errs['bbox_pred'] < 1e-4 and errs['dim_orien_pred'] < 1e-4, (k, errs)
            for name, v in errs.items():
                assert v < tol_.HEAD_OUTPUT_E2E, (k, name, v)
                worst[name] = max(worst.get(name, 0.0), v)
            for a, b in zip(out, outs[k % 4]):                   # a slot's second forward repeats its first bit for bit
                assert torch.equal(a, b), k
        with torch.no_grad():
            iso = _heads_on_reference_rois(m, l, g, 'f16x3', dev)
    finally:
        engine.KEY_HITS = None
        streams.set_pairs_in_flight(prev)
    print('four in flight on the shipped plans (%d of them used) vs reference code: worst errs %s; heads fed the reference rois %s'
          % (len(used), worst, iso))
    assert all(v < 1e-4 for v in iso.values()), iso


def test_demo_pair_four_in_flight_on_the_shipped_plans(dev):
    """configs[0]'s natural image through the fused preprocessing, four in flight on the shipped plans."""
    from stereo_rcnn_amd import engine, fixture, streams
    from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet
    from test_demo_pair import _rows
    pair = np.load(os.path.join(GOLD, 'demo_pair_u8.npz'))
    gold = np.load(os.path.join(GOLD, 'reference_demo_pair_r101_seed3.npz'))
    m = resnet(('__background__', 'Car'), 101, pretrained=False)
    m.create_architecture()
    m.load_state_dict(fixture.demo_state_dict(3))
    m.cuda().eval()
    m.precision, m.use_program = 'f16x3', True
    lu, ru = torch.from_numpy(pair['left']).to(dev), torch.from_numpy(pair['right']).to(dev)
    prev = streams.pairs_in_flight()
    try:
        outs = _in_flight(m, lambda mdl, slot: mdl.forward_images(lu, ru, slot=slot)[0], 4, 2)
        ref_l, ref_r = _rows(gold['rois_left']), _rows(gold['rois_right'])
        for k, out in enumerate(outs):
            rl, rr = out[0][0].cpu(), out[1][0].cpu()
            d = (ref_l[:, None, 1:] - rl[None, :, 1:]).abs().amax(2)
            best, idx = d.min(1)
            ok = best < tol_.PROPOSAL_MATCH_PX
            assert float(ok.float().mean()) >= 0.97, k
            tol_.observe('proposal_match_px', best[ok].max())
            errs = {'rois_right': tol_.observe('proposal_match_px', (rr[idx[ok]] - ref_r[ok]).abs().max())}
            for name, t in (('cls_prob', out[2][0]), ('bbox_pred', out[3][0]), ('dim_orien_pred', out[4][0]), ('kpts_prob', out[5]),
                            ('left_border_prob', out[6]), ('right_border_prob', out[7])):
                errs[name] = tol_.observe('e2e_' + name, (t.cpu()[idx[ok]] - _rows(gold[name])[ok]).abs().max())
            assert errs['bbox_pred'] < 1e-4 and errs['dim_orien_pred'] < 1e-4, (k, errs)
            assert errs.pop('rois_right') < tol_.PROPOSAL_MATCH_PX and all(v < tol_.HEAD_OUTPUT_E2E for v in errs.values()), (k, errs)
        # heads fed the reference code's rois, on the plan of slot 0 as the in-flight forwards left it
        with torch.no_grad():
            plan = m._get_plan(1, 600, 1987, 0)
            plan.rois_left.copy_(torch.from_numpy(gold['rois_left']).to(dev))
            plan.rois_right.copy_(torch.from_numpy(gold['rois_right']).to(dev))
            saved, engine.PRECISION = engine.PRECISION, 'f16x3'
            try:
                plan.heads()
            finally:
                engine.PRECISION = saved
            torch.cuda.synchronize()
            o = plan.outputs()
        iso = {}
        for name in ('cls_prob', 'bbox_pred', 'dim_orien_pred', 'kpts_prob', 'left_border_prob', 'right_border_prob'):
            ref = torch.from_numpy(gold[name])
            iso[name] = float((o[name].cpu().reshape(ref.shape) - ref).abs().max())
    finally:
        streams.set_pairs_in_flight(prev)
    print('demo pair, four in flight on the shipped plans: last errs %s; heads fed the reference rois %s' % (errs, iso))
    assert all(v < 1e-4 for v in iso.values()), iso
