"""Throughput tuner of the conv plans: the measured multi-stream step is the objective.

engine._tune picks, per conv shape, the plan that is fastest ALONE on the chip.  The serving regime (bench.py's headline,
pipeline.detect_3d_stream) keeps several batch-1 forwards in flight on separate HIP streams, and there a plan is worth what
it costs the whole mix -- CU-time, LDS it blocks, reduction launches it adds -- which no per-launch timing predicts: timing
a launch beside copies of itself (engine.TUNE_MODE 'concurrent') picks plans that win that contest and LOSE the real one
(profiles/tune_objective_r04.txt: 132 vs 138 pairs/s).  So this tuner measures the real thing: coordinate descent over the
per-shape plans, heaviest shapes first, a few candidates per shape (the fastest of the in-situ tuner's own log), each
candidate judged by the wall time of the actual S-streams-in-flight step (forward + decode + class NMS), accepted only if it
beats the incumbent by more than the measurement's noise.  The result is a plan file (engine.save_plans) that the library
ships for the GPU model / frame size it was tuned on (stereo_rcnn_amd/plans/); other shapes keep the in-situ tuner.

Product code: nothing here touches the oracle; every launch is the library's.
"""
import json
import os
import time

import torch

from . import engine
from . import streams as _streams
from . import postprocess as hpost

PLANS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'plans')


def shipped_plans_path(name='mi355x.json'):
    return os.path.join(PLANS_DIR, name)


def load_shipped_plans(name='mi355x.json'):
    """Adopt the shipped throughput-tuned plans (shape keys carry batch and frame size, so only matching shapes use them).
    Returns the number of plans loaded (0 if the file is absent)."""
    from . import serving                  # the product's own loader (device-model gated); tools force a reload
    return serving.load_shipped_plans(name, force=True)


class StepRunner(object):
    """S batch-1 forwards in flight round-robin on S streams: the headline step of bench.py (forward + decode + class NMS)."""

    def __init__(self, model, im_l, im_r, im_info, streams=3, kpts=True):
        self.model, self.inputs, self.S, self.kpts = model, (im_l, im_r, im_info), int(streams), kpts
        self.streams = _streams.main_streams(self.S) if self.S > 1 else [None]

    def step(self, slot):
        """kpts=True: the headline step (keypoint branch for all 300 rois inside the forward).  kpts=False: the pipeline's default
        form of the same step -- forward without the branch, decode + class NMS, then the branch on the kept detections only."""
        im_l, im_r, im_info = self.inputs
        out = self.model(im_l, im_r, im_info, slot=slot, kpts=self.kpts, alias_outputs=True)
        det = hpost.decode_detections(out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], im_info[0:1])
        keep_idx, num = hpost.class_nms_device(det, 1, 0.05)
        if not self.kpts:
            plan = self.model._get_plan(int(im_l.shape[0]), int(im_l.shape[2]), int(im_l.shape[3]), slot)
            plan.kpts_for_kept(out[0][0].contiguous(), keep_idx, num, im_info[0:1].contiguous(), det['kpts'], self.model.precision)

    def run(self, n):
        _streams.set_pairs_in_flight(self.S)      # (the tuner itself starts from whatever plans are loaded: it does not call serving.enter)
        for k in range(n):
            if self.S == 1:
                self.step(0)
            else:
                with torch.cuda.stream(self.streams[k % self.S]):
                    self.step(k % self.S)

    def measure(self, steps=24, repeats=3):
        """median over `repeats` of the wall time per step (ms) of `steps` steps; one untimed round first (re-records the
        launch programs after a plan change)"""
        with torch.no_grad():
            self.run(self.S)
            torch.cuda.synchronize()
            ts = []
            for _ in range(repeats):
                t0 = time.perf_counter()
                self.run(steps)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3 / steps)
        return sorted(ts)[len(ts) // 2]


def tune_throughput(model, im_l, im_r, im_info, streams=3, top_shapes=40, cands_per_shape=5, min_gain=0.004, rounds=2,
                    steps=24, log=None, runner=None):
    """Coordinate descent described in the module docstring.  The model must be on the f16x3 engine with use_program set as
    in production.  Returns (ms_before, ms_after, [(key, old_plan, new_plan, ms)]).  `runner`: an object with StepRunner's
    step(slot) / measure(steps) (tests inject a model of the machine; default: the real multi-stream step)."""
    say = log or (lambda *a: None)
    run = runner if runner is not None else StepRunner(model, im_l, im_r, im_info, streams)
    with torch.no_grad():
        # one eager forward with the hit counter armed: which shape keys does a forward launch, how often (also makes sure the
        # in-situ tuner has seen every shape: its log supplies the candidates and their isolated times)
        prog, model.use_program = getattr(model, 'use_program', False), False
        engine.KEY_HITS = {}
        run.step(0)
        if runner is None:
            torch.cuda.synchronize()
        hits, engine.KEY_HITS = engine.KEY_HITS, None
        model.use_program = prog
    weight = {}
    for key, n in hits.items():
        logd = dict(engine._TUNE_LOG.get(key, []))
        cur = engine._TUNED.get(key)
        if not logd or cur is None:
            continue
        weight[key] = n * logd.get(cur, min(logd.values()))
    order = sorted(weight, key=weight.get, reverse=True)[:top_shapes]
    base = run.measure(steps)
    best = base
    say('throughput tuner: %d shape keys in a forward, tuning the %d heaviest; start %.3f ms/step (%d in flight)'
        % (len(hits), len(order), base, streams))
    changes = []
    for rnd in range(rounds):
        changed = 0
        for key in order:
            logd = dict(engine._TUNE_LOG[key])
            cur = engine._TUNED[key]
            by_time = [pl for pl in sorted(logd, key=logd.get) if pl != cur]
            # ... plus the FAT unsplit tiles (256 rows) whatever their isolated rank: few long-lived workgroups lose the latency
            # contest on the small-M layers and can still win the mix (profiles/fat_tiles_r05.txt: layer3 conv1 / conv2 on the
            # 256x128 tile + conv3 on 256x256: -2.5 % step time at four in flight)
            fat = [pl for pl in by_time if pl[0] >= 4 and pl[4] == 1]
            # ... and the LEAN 8-wave tiles (2-stage ring, 64 KB of LDS): they leave room on the CU for a workgroup of another
            # forward's launch, which the deeper rings that win alone do not
            lean = [pl for pl in by_time if pl[2] == 8 and pl[3] == 2 and pl[4] == 1 and pl[0] < 4]
            cands = by_time[:cands_per_shape]
            cands += [pl for pl in fat + lean if pl not in cands]
            for pl in cands:
                engine.set_plan(key, pl)
                t = run.measure(steps)
                if t < best * (1.0 - min_gain):
                    # confirm against a FRESH measurement of the incumbent (clocks and temperature drift by more than the
                    # gain looked for over a tuning run): incumbent, candidate again -- both must agree
                    engine.set_plan(key, cur)
                    ti = run.measure(steps)
                    engine.set_plan(key, pl)
                    t2 = run.measure(steps)
                    if t2 < ti * (1.0 - min_gain) and t < ti * (1.0 - min_gain):
                        say('  %s x%d: %s -> %s  %.3f -> %.3f ms/step' % (_fmt_key(key), hits[key], cur, pl, ti, max(t, t2)))
                        changes.append((key, cur, pl, max(t, t2)))
                        best, cur, changed = max(t, t2), pl, changed + 1
                        continue
                    best = ti                                      # follow the drift of the incumbent
                engine.set_plan(key, cur)
        say('round %d: %d plans changed, %.3f ms/step' % (rnd + 1, changed, best))
        if not changed:
            break
    final = run.measure(steps)
    say('throughput tuner: %.3f -> %.3f ms/step (%.1f -> %.1f pairs/s)' % (base, final, 1e3 / base, 1e3 / final))
    return base, final, changes


def _fmt_key(key):
    prec, B, H, W, OH, OW, cin, cout, kh, kw, s, p = key[:12]
    return 'B%d %dx%d %dx%d/%d %d->%d%s' % (B, H, W, kh, kw, s, cin, cout, (' ' + ' '.join(str(v) for v in key[14:])) if len(key) > 14 else '')


def save_shipped(name='mi355x.json', meta=None):
    os.makedirs(PLANS_DIR, exist_ok=True)
    engine.save_plans(shipped_plans_path(name))
    if meta is not None:
        with open(shipped_plans_path(name + '.meta'), 'w') as f:
            json.dump(meta, f, indent=1)
