"""Which hardware queue a HIP stream lands on, and when a plan may fork side streams.

HIP folds a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, least-used queue first.  Streams that share
a queue run in submission order, and an event wait issued on one of them (a barrier packet) holds up everything queued behind
it.  The serving regime keeps several forwards in flight; until round 4 each also forked two side streams (FPN laterals / small
RPN levels / box head), so 3 + 6 streams shared 4 queues: the side stream of forward A regularly sat in front of the main stream
of forward B, and B stood still until A reached the point its side stream waited for -- in a kernel trace of the
three-in-flight headline the main queues were busy 36-46 % of the time and ONE kernel had the chip to itself for 68 % of it
(profiles/queue_mapping_r04.txt).

Measured, pairs/s of the headline step, same box within a block (profiles/queue_mapping_r04.txt):
  pooled main + pooled side streams (rounds 1-3)          137.4-138.2     3 in flight
  pooled main, NO side streams                            141.6 -> 151.9-153.3 with plans tuned in this regime
  own-queue main ('dedicated'), no side streams           147.0-148.0 -> 151.7-152.1;  four in flight 152.4-152.8
  pooled main, four in flight, 4 queues                   142.5-143.3  (two forwards share a queue);  8 queues: 153.5
  own-queue main AND own-queue side streams               124.8  (nine busy queues: the command processor serves a few at a time)
  CU-masked partitions (each forward on 1/S of every XCD) 128 / 118 / 140 / 97 for S = 2 / 3 / 4 / 8: static partitioning
                                                          loses to the dispatcher's own sharing
Hence the policy:
  * every forward in flight gets a main stream with a hardware queue of its own: pooled streams with enough queues
    (`ensure_hw_queues`, called by the entry points before HIP starts), or `new_stream('dedicated')`, which owns a queue
    whatever the pool looks like (include/srcnn_hip.h: srcnn_stream_create) -- at the price that HIP gives such streams the
    legacy blocking relation to the NULL stream: work issued on the null stream while they exist pays for it (one pair at a
    time on the null stream: 8.3 -> 13 ms with 2-3 idle dedicated streams around), so they are not the default;
  * a plan forks its independent branches onto side streams only while ONE forward is in flight (latency mode, where they
    buy 2 %); with several in flight the other forwards fill the chip and the branches stay on the main stream
    (`set_pairs_in_flight` / `branch_overlap`; the launch programs are recorded per regime).

Environment switches (A/B experiments; the defaults are what the measurements picked):
  SRCNN_MAIN_STREAMS = pool | dedicated | partition    streams of the forwards in flight (tune.StepRunner, bench.py, pipeline)
  SRCNN_SIDE_STREAMS = auto | none | dedicated | pool | high     a plan's two side streams ('none': branches always run on the
                       main stream; 'auto': pooled side streams, used only while one forward is in flight)
"""
import ctypes
import os

import torch

from . import _lib

MAIN_KIND = os.environ.get('SRCNN_MAIN_STREAMS', 'pool')
SIDE_KIND = os.environ.get('SRCNN_SIDE_STREAMS', 'auto')
KINDS = ('pool', 'dedicated', 'high')


HW_QUEUES = 8


def ensure_hw_queues(n=HW_QUEUES):
    """Ask the HIP runtime for at least `n` hardware queues per device (GPU_MAX_HW_QUEUES, read once when HIP starts), so that
    the pooled streams of up to n - 1 forwards in flight and the null stream do not share queues.  Entry points that keep several pairs in flight
    (bench.py, test_net.py) call this before the first HIP call; returns False -- and changes nothing -- when HIP is already
    up with fewer queues (the caller may then use 'dedicated' main streams, or at most GPU_MAX_HW_QUEUES - 1 in flight).
    A smaller GPU_MAX_HW_QUEUES already in the environment is raised unless SRCNN_KEEP_HW_QUEUES=1 says it is deliberate."""
    cur = os.environ.get('GPU_MAX_HW_QUEUES')
    if torch.cuda.is_initialized() or _queues_at_hip_start is not None:
        return _hw_queues() >= n
    if cur is not None and cur.strip().isdigit():
        if int(cur) >= n:
            return True
        if os.environ.get('SRCNN_KEEP_HW_QUEUES', '0') not in ('', '0'):      # a smaller value set on purpose stays
            return False
    os.environ['GPU_MAX_HW_QUEUES'] = str(n)
    return True


_queues_at_hip_start = None       # GPU_MAX_HW_QUEUES as HIP read it


def _env_queues():
    cur = os.environ.get('GPU_MAX_HW_QUEUES', '4')
    return int(cur) if cur.strip().isdigit() else 4


# the value HIP starts with is the one in the environment when the runtime initialises -- and the runtime may be
# initialised by the LIBRARY (a HIP call behind _lib.lib()) before torch's lazy init says so.  The count is therefore pinned at
# the earliest of: this module's import if torch already runs HIP; the moment _lib loads the library (hip_may_start, called
# before its first call); the first query that finds torch initialised.  Until then the environment may still be changed.
if torch.cuda.is_initialized():
    _queues_at_hip_start = _env_queues()


def hip_may_start():
    """_lib calls this right before it loads libsrcnn_hip.so: from here on a library call may start HIP.  Last chance to ask for
    the serving queue count (as every entry point does through serving.before_hip()); what the environment says now is pinned."""
    global _queues_at_hip_start
    if _queues_at_hip_start is None and not torch.cuda.is_initialized():
        ensure_hw_queues()
        _queues_at_hip_start = _env_queues()


def _hw_queues():
    """The queue count HIP runs with: the variable is read once when HIP starts, so once HIP is up the value seen THEN counts,
    not whatever the environment says later."""
    global _queues_at_hip_start
    cur = os.environ.get('GPU_MAX_HW_QUEUES', '4')
    n = int(cur) if cur.strip().isdigit() else 4
    if _queues_at_hip_start is not None:
        return _queues_at_hip_start
    if torch.cuda.is_initialized():
        _queues_at_hip_start = n
    return n


def max_pairs_in_flight():
    """Forwards that can each have a pooled stream on a hardware queue of its own (one queue is the null stream's)."""
    return max(1, _hw_queues() - 1)


_handles = []      # native handles of the streams created here: they live as long as the process (the slots' streams are created
                   # once and cached; destroying a stream from a finalizer at interpreter exit races the HIP runtime's own
                   # teardown and crashes)


def destroy_all():
    """Destroys every native stream this module created, after dropping the wrappers the package itself caches (the in-flight
    streams of main_streams): for long-lived processes that rebuild their stream set.  Callers must have dropped THEIR wrappers
    (tune.StepRunner objects, streams handed out by main_streams) -- a wrapper used after this call is a dangling handle."""
    import sys
    _flight.clear()
    torch.cuda.synchronize()
    while _handles:
        _lib.lib().srcnn_stream_destroy(_handles.pop())


def new_stream(kind='pool', device=None):
    """A new stream on `device` (default: current).  kind: 'pool' (torch's pooled NON-BLOCKING streams on the shared hardware
    queues), 'high' (pooled, high priority: HIP keeps a separate queue set per priority), 'dedicated' (own hardware queue -- created
    through hipExtStreamCreateWithCUMask, which gives the stream HIP's default, NULL-stream-BLOCKING semantics: work on the null
    stream serialises with it, see the module docstring)."""
    assert kind in KINDS, kind
    if kind == 'pool':
        return torch.cuda.Stream(device=device)
    if kind == 'high':
        return torch.cuda.Stream(device=device, priority=-1)
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().srcnn_stream_create(1, ctypes.byref(h)), "srcnn_stream_create")
    _handles.append(h.value)
    return torch.cuda.ExternalStream(h.value, device=dev)


def partition_masks(parts, n_cus=256):
    """CU masks that split the device into `parts` equal partitions, each holding the same number of CUs of EVERY XCD.

    Bit b of a queue's CU mask: the driver deals the bits out round-robin over the XCDs (bit b -> XCD b % 8, that XCD's CU number
    b // 8; verified with srcnn_probe_placement, profiles/queue_mapping_r04.txt), so partition k takes the bits whose local CU
    number (b >> 3) falls into its share -- never a whole XCD (the workgroups of a dispatch are dealt to ALL XCDs, block b to
    XCD b % 8, whatever the mask), always `n_cus / 8 / parts` CUs in each.  parts in (2, 4) use bits 3-4 only, which is an even
    split under an XCD-major enumeration as well."""
    assert n_cus % 64 == 0 and parts >= 1
    per_xcd = n_cus // 8
    words = n_cus // 32
    masks = []
    for k in range(parts):
        m = [0] * words
        for b in range(n_cus):
            local = b >> 3
            if parts in (2, 4):
                mine = (local & (parts - 1)) == k
            else:
                mine = local * parts // per_xcd == k
            if mine:
                m[b >> 5] |= 1 << (b & 31)
        masks.append(m)
    return masks


def masked_stream(mask_words, device=None):
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    h = ctypes.c_void_p()
    arr = (ctypes.c_uint * len(mask_words))(*mask_words)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().srcnn_stream_create_cu_mask(len(mask_words), arr, ctypes.byref(h)), "srcnn_stream_create_cu_mask")
    _handles.append(h.value)
    s = torch.cuda.ExternalStream(h.value, device=dev)
    s._srcnn_cu_mask = list(mask_words)
    return s


_warned = set()


def check_queue_supply(n, kind=None):
    """Warn (once per count) when `n` forwards in flight on pooled streams cannot each have a hardware queue: two of them would
    share one and run in turns (four in flight on the default four queues: 142.5 instead of 152.6 pairs/s)."""
    kind = kind or MAIN_KIND
    if kind != 'pool' or n <= max_pairs_in_flight():
        return True
    if n not in _warned:
        _warned.add(n)
        import logging
        logging.getLogger('stereo_rcnn_amd').warning(
            '%d pairs in flight on pooled HIP streams but GPU_MAX_HW_QUEUES allows %d with a hardware queue each (plus the null '
            'stream): call stereo_rcnn_amd.streams.ensure_hw_queues() before the first HIP call, use fewer pairs in flight, or '
            'SRCNN_MAIN_STREAMS=dedicated', n, max_pairs_in_flight())
    return False


def main_streams(n, device=None, kind=None):
    """Streams for `n` forwards in flight.  kind 'partition': each on its own 1/n of every XCD's CUs."""
    kind = kind or MAIN_KIND
    check_queue_supply(n, kind)
    if kind == 'partition':
        n_cus = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
        return [masked_stream(m, device) for m in partition_masks(n, n_cus)]
    # ONE set of in-flight streams per (device, kind) and process (round 6): every caller that keeps n forwards in flight -- the
    # benchmark's headline loop, tune.StepRunner, pipeline.detect_3d_stream's slots -- gets the SAME first n streams.  Fresh streams
    # per caller looked harmless and were not: HIP binds a stream to a hardware queue when it is first used and never gives the
    # queue back, so bench.py's four headline streams + the pipeline's four slot streams + the null stream were nine streams on
    # eight queues, two of the slots' forwards shared one queue, and the 3-D flow inside bench.py ran at 7.1 ms per pair where the
    # same flow alone in a process took 5.9 (profiles/flow3d_queue_sharing_r06.txt).
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    have = _flight.setdefault((dev, kind), [])
    while len(have) < n:
        have.append(new_stream(kind, device))
    return have[:n]


_flight = {}      # (device index, kind) -> the process's in-flight streams of that kind


def side_streams(n, device=None, kind=None):
    """Side streams of one plan, or None when branches are to run on the main stream (kind 'none')."""
    kind = kind or SIDE_KIND
    if kind == 'none':
        return None
    return [new_stream('pool' if kind == 'auto' else kind, device) for _ in range(n)]


_pairs_in_flight = 1


def set_pairs_in_flight(n):
    """Tell the plans how many forwards the caller keeps in flight (1: one at a time).  Plans re-record their launch programs
    when the answer of branch_overlap() changes."""
    global _pairs_in_flight
    _pairs_in_flight = max(1, int(n))


def pairs_in_flight():
    return _pairs_in_flight


def branch_overlap():
    """Should a plan fork its independent branches onto side streams right now?"""
    if SIDE_KIND == 'none':
        return False
    if SIDE_KIND == 'auto':
        return _pairs_in_flight == 1
    return True
