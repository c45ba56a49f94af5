"""The serving regime: one place that sets up what `bench.py`'s headline is measured in, used by the benchmark AND by the
product's own entry points (`pipeline.detect_3d_stream`, `test_net.py`, `demo.py`, `tune.StepRunner`).

Three things make the regime (profiles/queue_mapping_r04.txt, profiles/tune_*_r04.txt):
  1. enough hardware queues that every forward in flight has one to itself (`GPU_MAX_HW_QUEUES`, read once when HIP starts:
     `before_hip()` must run before the first HIP call of the process);
  2. the plans know how many forwards are in flight (`streams.set_pairs_in_flight`): with several, every launch of a forward
     stays on its main stream; alone, the independent branches fork onto side streams;
  3. the conv plans tuned with the measured several-in-flight step as objective (`plans/mi355x.json`, written by
     tools/tune_headline.py): loaded once per process when several forwards are to be in flight -- and only on the GPU model
     they were tuned on; their shape keys carry batch and frame size, so other shapes keep the in-situ tuner whatever is
     loaded.  One forward at a time runs on the in-situ tuner's latency picks.

Nothing here touches the oracle; every launch is the library's.
"""
import logging
import os

import torch

from . import engine
from . import streams as _streams

_log = logging.getLogger('stereo_rcnn_amd')

DEFAULT_PAIRS_IN_FLIGHT = 4
PLANS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'plans')
# A/B switch (and the way to keep a process on the in-situ tuner): SRCNN_SHIPPED_PLANS=0
USE_SHIPPED_PLANS = os.environ.get('SRCNN_SHIPPED_PLANS', '1') != '0'

# the shipped plan files and the GPU they were tuned on: (gcnArchName prefix, compute units)
SHIPPED = {'mi355x.json': ('gfx950', 256)}
# A/B (dev): SRCNN_SHIPPED_PLANS_FILE=<name in plans/> is read in place of mi355x.json (same gating)
PLANS_FILE = os.environ.get('SRCNN_SHIPPED_PLANS_FILE', 'mi355x.json')

_loaded = {}          # plan file -> number of plans adopted (0: looked at, not applicable)


def before_hip(n_queues=None):
    """Call before the first HIP call of the process (entry points do): asks the runtime for one hardware queue per forward in
    flight plus the null stream's.  Returns False -- and changes nothing -- when HIP is already up with fewer queues."""
    return _streams.ensure_hw_queues(n_queues or _streams.HW_QUEUES)


def shipped_plans_path(name='mi355x.json'):
    return os.path.join(PLANS_DIR, name)


def device_matches(name='mi355x.json', device=None):
    """Is `device` (default: the current one) the GPU model the plan file was tuned on?  Plans are launch geometry, not
    arithmetic -- a foreign plan is still correct, but it was not measured there, so it is not adopted."""
    arch, cus = SHIPPED.get(name, (None, None))
    if arch is None or not torch.cuda.is_available():
        return False
    p = torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device)
    return str(getattr(p, 'gcnArchName', '')).startswith(arch) and int(p.multi_processor_count) == cus


def load_shipped_plans(name='mi355x.json', device=None, force=False):
    """Adopt the shipped throughput-tuned conv plans once per process (device-model gated; shape keys gate the rest).
    Returns the number of plans adopted by THIS call (0 when already loaded, absent, switched off or another GPU model)."""
    if name in _loaded and not force:
        return 0
    _loaded[name] = 0
    p = shipped_plans_path(PLANS_FILE if name == 'mi355x.json' else name)
    if not USE_SHIPPED_PLANS or not os.path.exists(p):
        return 0
    if not device_matches(name, device):
        _log.info('conv plans %s were tuned on another GPU model: not adopted, the in-situ tuner picks every plan', name)
        return 0
    _loaded[name] = engine.load_plans(p)
    return _loaded[name]


def plans_loaded(name='mi355x.json'):
    """Number of shipped plans this process runs on (0: none)."""
    return _loaded.get(name, 0)


def drop_shipped_plans(name='mi355x.json'):
    """Forget that the file was looked at (tests, A/B tools): the next enter() with several in flight loads it again."""
    _loaded.pop(name, None)


def enter(pairs_in_flight=DEFAULT_PAIRS_IN_FLIGHT, device=None, plans=True):
    """Put the process into the regime of `pairs_in_flight` forwards in flight: tells the plans (branch placement), checks the
    hardware-queue supply, and -- several in flight -- adopts the shipped throughput-tuned plans.  Idempotent and cheap; every
    streamed entry point calls it.  Returns a dict describing the regime (bench.py prints it)."""
    n = max(1, int(pairs_in_flight))
    _streams.set_pairs_in_flight(n)
    queues_ok = _streams.check_queue_supply(n) if n > 1 else True
    adopted = load_shipped_plans(device=device) if (plans and n > 1) else 0
    return {'pairs_in_flight': n, 'hw_queues_ok': bool(queues_ok), 'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
            'shipped_plans_adopted_now': adopted, 'shipped_plans': plans_loaded(),
            'branch_side_streams': _streams.branch_overlap()}
