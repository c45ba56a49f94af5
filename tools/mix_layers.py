"""Per-layer evidence of the regime `value` is measured in: marginal in-mix cost of every conv group and the
workgroup residency table from the kernel's own stamps, S batch-1 forwards in flight (stereo_rcnn_amd/mix_table.py).

    python tools/mix_layers.py [--streams 4] [--steps 24] [--out profiles/mix_layers_r05.txt] [--no-residency] [--json out.json]
"""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import serving
serving.before_hip()
from stereo_rcnn_amd import engine, fixture, layer_table, mix_table, tune
from stereo_rcnn_amd import streams as sstreams
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=4)
ap.add_argument('--steps', type=int, default=24)
ap.add_argument('--out', default='')
ap.add_argument('--json', default='')
ap.add_argument('--no-residency', action='store_true')
ap.add_argument('--no-marginal', action='store_true')
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
run = tune.StepRunner(m, l, r, info, S)
lines = ['MI355X, BASELINE configs[1] frame (600x1987), default engine, %d forwards in flight, shipped plans: %d' % (S, regime['shipped_plans'])]
with torch.no_grad():
    for _ in range(2):
        run.run(S)
        torch.cuda.synchronize()
    base = run.measure(args.steps)
    lines.append('headline step (forward + decode + class NMS): %.3f ms = %.1f pairs/s' % (base, 1e3 / base))
    # every conv launch alone on the chip, SAME plans (the layer table of bench.py runs on the in-situ plans of one pair at a time)
    prog, m.use_program = m.use_program, False
    sstreams.set_pairs_in_flight(S)                   # branches on the main stream, as in the mix
    def serial():
        run.step(0)
    rows = layer_table.measure(serial, reps=3, precision='f16x3')
    m.use_program = prog
    alone_ms = sum(r['us'] for r in rows) / 1e3
    lines.append('the same %d conv launches each ALONE on the chip (these plans): %.3f ms in total' % (len(rows), alone_ms))
    marg = resid = None
    if not args.no_marginal:
        b, marg = mix_table.marginal(run, rows, steps=args.steps, log=lambda s: print(s, flush=True))
        lines += ['', mix_table.format_marginal(b, marg, S)]
    if not args.no_residency:
        ms, resid = mix_table.residency(run, S, steps=args.steps)
        lines += ['', mix_table.format_residency(ms, resid, S, base_ms=base)]
txt = '\n'.join(lines)
print(txt)
if args.out:
    with open(args.out, 'w') as f:
        f.write(txt + '\n')
if args.json:
    with open(args.json, 'w') as f:
        json.dump(mix_table.for_json(base, marg or [], resid), f)
