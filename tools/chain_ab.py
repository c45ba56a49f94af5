"""A/B of the chained bottleneck launches (csrc/conv_chain.hip) on the headline step: S batch-1 forwards in flight, forward + decode
+ class NMS, shipped plans.  One process, one box: the chain switched off / on, per-layer chain tiles, several depths.
    python tools/chain_ab.py [--streams 4,6] [--steps 40] [--configs off,default,...]
Configs: off | default | name=P:mr.waves.stages.na.nb[/P:...]   (P = bottleneck planes 64 / 128 / 256 / 512; 'x' = unchained)"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import engine, fixture, serving, tune
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', default='4')
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--repeats', type=int, default=3)
ap.add_argument('--configs', default='off,default')
args = ap.parse_args()
serving.before_hip()
dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101, pretrained=False)
m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3))
m.cuda().eval()
m.precision = 'f16x3'
m.use_program = True
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
DEFAULT = dict(engine.CHAIN_TILES)


def parse(cfg):
    if cfg == 'off':
        return 'off', None
    if cfg == 'default':
        return 'default', dict(DEFAULT)
    name, spec = cfg.split('=')
    tiles = dict(DEFAULT)
    for part in spec.split('/'):
        P, t = part.split(':')
        tiles[int(P)] = None if t == 'x' else tuple(int(v) for v in t.split('.'))
    return name, tiles


for S in [int(s) for s in args.streams.split(',')]:
    serving.enter(S, device=dev)
    run = tune.StepRunner(m, l, r, info, S)
    for cfg in args.configs.split(','):
        name, tiles = parse(cfg)
        engine.BOTTLENECK_CHAIN = '0' if tiles is None else '1'
        if tiles is not None:
            engine.CHAIN_TILES.clear()
            engine.CHAIN_TILES.update(tiles)
        engine.PLAN_EPOCH += 1                       # recorded launch programs are re-recorded
        with torch.no_grad():
            for _ in range(2):
                run.run(S)
                torch.cuda.synchronize()
        ms = run.measure(args.steps, args.repeats)
        print('S=%d %-28s %.3f ms/step = %.1f pairs/s   %s' % (S, name, ms, 1e3 / ms, '' if tiles is None else
              ' '.join('%d:%s' % (k, '.'.join(map(str, v)) if v else 'x') for k, v in sorted(tiles.items()))), flush=True)
