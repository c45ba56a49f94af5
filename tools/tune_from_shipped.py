"""Throughput tuner, incremental form (round 5): start from the SHIPPED plans (stereo_rcnn_amd/plans/mi355x.json) and try, for the
heaviest shapes of the forward, every legal unsplit tile of the SPLIT16 engine -- including the fat 256-row tiles on the small-M
layers and the lean 2-stage 8-wave tiles, which the in-situ tuner's latency ranking never offers -- against the measured
S-in-flight step (stereo_rcnn_amd.tune.tune_throughput does the descent and the confirmation).  A from-scratch run of the tuner
lands in a worse basin than the shipped file (profiles/tune_throughput_from_scratch_r05.txt: 7.26 ms/step against 6.68), so the file is refined,
not regenerated.      python tools/tune_from_shipped.py [--streams 4] [--out mi355x.json] [--top 32]
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import engine, fixture, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=4)
ap.add_argument('--out', default='mi355x.json')
ap.add_argument('--top', type=int, default=24)
ap.add_argument('--rounds', type=int, default=1)
ap.add_argument('--min-gain', type=float, default=0.004)
ap.add_argument('--steps', type=int, default=24)
args = ap.parse_args()
S = args.streams
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
regime = serving.enter(S)
print('starting from the shipped plans:', regime['shipped_plans'])
run = tune.StepRunner(m, l, r, info, S)
with torch.no_grad():
    for _ in range(2):
        run.run(S)
        torch.cuda.synchronize()
noise = [run.measure(args.steps) for _ in range(4)]
print('noise check, 4 x median-of-3 of %d steps: ' % args.steps + ' '.join('%.3f' % t for t in noise) + ' ms/step')

# which shape keys does a forward launch, how often
with torch.no_grad():
    prog, m.use_program = m.use_program, False
    engine.KEY_HITS = {}
    run.step(0)
    torch.cuda.synchronize()
    hits, engine.KEY_HITS = engine.KEY_HITS, None
    m.use_program = prog

TILES = [(4, 4, 8, 2), (4, 2, 8, 3), (2, 2, 8, 4), (2, 2, 8, 2), (2, 2, 4, 2), (2, 1, 4, 2), (1, 1, 4, 2),
         (4, 4, 4, 2), (4, 2, 4, 2), (4, 2, 4, 3), (2, 4, 4, 2), (2, 4, 4, 3)]       # round 6: the 4-wave forms with 128-wide wave tiles
for key, n in hits.items():
    if key[0] != 'f16x3' or key not in engine._TUNED or 'lim' in key:
        continue
    B, OH, OW, cin, cout, kh, kw, mode = key[1], key[4], key[5], key[6], key[7], key[8], key[9], key[12]
    M, N, K = B * OH * OW, cout, cin * kh * kw
    cur = tuple(engine._TUNED[key])
    head2, x2 = 'head2' in key, 'x2' in key
    cands = []
    for t in TILES:
        mr, nr = t[0], t[1]
        if head2 and t not in ((4, 4, 8, 2), (2, 2, 8, 2)):
            continue
        if nr == 4 and (N <= 128 or x2 or M < 1024):
            continue
        if nr == 2 and N <= 64:
            continue
        if mr >= 2 and M <= 64 * (mr // 2):
            continue
        cands.append(t + (1,))
    if cur[4] > 1 and not head2:                                   # the incumbent's split, and its tile unsplit, stay in the race
        cands += [cur, cur[:4] + (1,)]
    cands = list(dict.fromkeys([cur] + cands))
    est = 2.0 * M * N * K / 300e12 * 1e3                            # ms at 300 TF/s: only orders the shapes by weight
    engine._TUNE_LOG[key] = [(pl, est) for pl in cands]
base, final, changes = tune.tune_throughput(m, l, r, info, streams=S, top_shapes=args.top, cands_per_shape=16, min_gain=args.min_gain,
                                            rounds=args.rounds, steps=args.steps, log=lambda s: print(s, flush=True), runner=run)
tune.save_shipped(args.out, {'gpu': torch.cuda.get_device_name(0), 'streams': S, 'workload': 'BASELINE configs[1], network input 600x1987, batch 1',
                             'started_from': 'the shipped plans of round 4', 'ms_per_step_before': round(base, 3), 'ms_per_step_after': round(final, 3),
                             'changes': [[list(k), list(a), list(b), round(t, 3)] for k, a, b, t in changes]})
print('wrote', tune.shipped_plans_path(args.out))
