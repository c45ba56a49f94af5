"""Host-side driver of the HIP convolution engine and NHWC helpers.

Thin Python over the C ABI (include/srcnn_hip.h): weight re-layout / frozen-BN folding
at load time, and one function per native op that fills the C descriptor and launches on
torch's current stream.  No arithmetic happens here.
"""
import ctypes

import torch

from . import _lib


class ConvW(object):
    """A convolution prepared for the engine: weight (Cout, KH, KW, Cin) K-contiguous, bias."""

    def __init__(self, weight, bias, kh, kw, stride, pad, relu, mode=0, cin2=0, stride2=1):
        self.weight = weight.contiguous()
        self.bias = None if bias is None else bias.contiguous()
        self.cout = int(weight.shape[0])
        # cin2 > 0 (1x1 only): the last cin2 columns of K belong to a SECOND input sampled with stride2 (prep_conv_shortcut)
        self.cin2, self.stride2 = int(cin2), int(stride2)
        self.cin = int(weight.numel() // (weight.shape[0] * kh * kw)) - self.cin2
        self.kh, self.kw, self.stride, self.pad, self.relu, self.mode = kh, kw, stride, pad, int(relu), mode
        self.alg_k = self.cin * kh * kw + self.cin2      # algorithmic K (the stem pads 147 -> 224; see prep_stem)
        self.w_hi = self.w_lo = None         # f16x3 engine operands (see split_f16x3)
        self.inv_scale = 1.0

    def split_f16x3(self):
        """Error-compensated split for the f16 MFMA engine (csrc/conv_f16x3.hip):
        w * 2^k = hi + lo with hi = f16(w 2^k), lo = f16(w 2^k - hi); k puts max|w| near 2^14 so
        that lo stays out of the f16 subnormal range.  Exact power-of-two scaling, undone in the epilogue."""
        if self.w_hi is None:
            import math
            m = float(self.weight.abs().max())
            k = math.floor(math.log2(16384.0 / m)) if m > 0 else 0
            ws = self.weight * (2.0 ** k)
            self.w_hi = ws.half()
            self.w_lo = (ws - self.w_hi.float()).half().contiguous()
            self.w_hi = self.w_hi.contiguous()
            self.inv_scale = 2.0 ** (-k)
        return self


def fold_bn(w, bn, eps=1e-5):
    """Frozen BN (resnet.py:300-309: always eval) folded into the preceding bias-free conv.
    Done in float64, stored float32:  w' = w * g/sqrt(v+eps),  b' = beta - mean * g/sqrt(v+eps)."""
    s = bn['weight'].double() / torch.sqrt(bn['running_var'].double() + eps)
    wf = (w.double() * s.view(-1, 1, 1, 1)).float()
    bf = (bn['bias'].double() - bn['running_mean'].double() * s).float()
    return wf, bf


def prep_conv(w, b, stride=1, pad=0, relu=False, bn=None, device='cuda'):
    """nn.Conv2d weight (Cout, Cin, KH, KW) -> engine layout (Cout, KH, KW, Cin)."""
    if bn is not None:
        w, b = fold_bn(w, bn)
    kh, kw = int(w.shape[2]), int(w.shape[3])
    wt = w.permute(0, 2, 3, 1).contiguous().to(device)
    return ConvW(wt, None if b is None else b.to(device), kh, kw, stride, pad, relu)


def prep_conv_shortcut(w3, bn3, wd, bnd, stride2, device='cuda'):
    """The last 1x1 conv of a bottleneck block and the block's projection shortcut (resnet.py:86-100:
    out = relu(bn3(conv3(t)) + bn_d(downsample(x)))) as ONE GEMM over K = [channels of t | channels of x]: both frozen BNs
    folded, weights concatenated along K, biases added.  The shortcut's result is then never written nor read back as a
    residual, and its launch is gone (srcnn_conv_desc.x2)."""
    w3f, b3 = fold_bn(w3, bn3)
    wdf, bd = fold_bn(wd, bnd)
    assert w3f.shape[2:] == (1, 1) and wdf.shape[2:] == (1, 1) and w3f.shape[0] == wdf.shape[0]
    wt = torch.cat((w3f[:, :, 0, 0], wdf[:, :, 0, 0]), 1).contiguous().view(w3f.shape[0], 1, 1, -1).to(device)
    return ConvW(wt, (b3.double() + bd.double()).float().to(device), 1, 1, 1, 0, True, cin2=int(wdf.shape[1]), stride2=stride2)


def prep_stem(w, bn, device='cuda'):
    """7x7/2 stem (resnet.py:109) as 7 row-taps of 32 floats = 8 pixels x NHWC4 (see srcnn_stem_pack)."""
    wf, bf = fold_bn(w, bn)
    cout = wf.shape[0]
    wt = torch.zeros(cout, 7, 8, 4, device=wf.device)
    wt[:, :, :7, :3] = wf.permute(0, 2, 3, 1)          # (co, kh, kw, c)
    cw = ConvW(wt.view(cout, 7, 1, 32).to(device), bf.to(device), 7, 1, 2, 0, True)
    cw.alg_k = 3 * 7 * 7
    return cw


def prep_deconv2x2(w, b, relu=True, device='cuda'):
    """nn.ConvTranspose2d(k=2, s=2) weight (Cin, Cout, 2, 2) -> GEMM rows ordered (i, j, co)."""
    cin, cout = int(w.shape[0]), int(w.shape[1])
    wt = w.permute(2, 3, 1, 0).contiguous().view(4 * cout, 1, 1, cin)
    return ConvW(wt.to(device), b.to(device), 1, 1, 1, 0, relu, mode=1)


def prep_linear_stack(ws, bs, device='cuda'):
    """Several nn.Linear(in, out_i) sharing the input -> one (sum out_i, 1, 1, in) conv."""
    w = torch.cat([x for x in ws], 0)
    b = torch.cat([x for x in bs], 0)
    return ConvW(w.view(w.shape[0], 1, 1, w.shape[1]).to(device), b.to(device), 1, 1, 1, 0, False)


def head_fragments(hcw):
    """The weights of a narrow 1x1 head (ConvW (N2, 1, 1, C)) for the MFMA-form fused head (srcnn_conv_desc.head_wf): split into
    hi / lo f16 like any weight of the f16x3 engine, rows zero-padded to a multiple of 8, and laid out in the order the kernel's
    lanes read them: [C / 16 steps][hi, lo][k group 0, 1][rows][8 halves], element (s, g, n, i) = W[n][16 s + 8 g + i].
    Returns (tensor, rows); cached on the ConvW."""
    if getattr(hcw, '_frag', None) is None:
        assert hcw.kh == 1 and hcw.kw == 1 and hcw.cin % 16 == 0 and hcw.cout <= 24
        hcw.split_f16x3()
        n2, C = hcw.cout, hcw.cin
        rows = -(-n2 // 8) * 8
        f = torch.zeros((2, rows, C), dtype=torch.float16, device=hcw.w_hi.device)
        f[0, :n2] = hcw.w_hi.view(n2, C)
        f[1, :n2] = hcw.w_lo.view(n2, C)
        f = f.view(2, rows, C // 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous()        # (step, hi/lo, k group, row, 8)
        hcw._frag = (f, rows)
    return hcw._frag


def conv_out_hw(h, w, k_h, k_w, stride, pad):
    return (h + 2 * pad - k_h) // stride + 1, (w + 2 * pad - k_w) // stride + 1


class FlopCounter(object):
    """Algorithmic conv FLOPs (2*M*N*K with the true K) of the launches issued while enabled, and their COMPULSORY bytes:
    every operand element the launch needs read once and every result written once, at the 4 bytes per element both activation
    formats and both weight forms (fp32, or hi + lo f16) occupy -- input pixels x Cin (only the sampled pixels of a strided
    1x1), weights Cout x K, residual and output M x Cout."""
    enabled = False
    flops = 0.0
    launches = 0
    bytes = 0.0
    rows = None          # list: one dict per launch (name, M, N, K, flops, bytes, plan) while enabled; see layer_table.py


# ---- SPLIT16 range guard (include/srcnn_hip.h: srcnn_range_flag_read): layers are tagged by name so that a tripped flag
# can be reported as the layer that produced the out-of-range activation
TAG_NAMES = {9001: 'upsample_add', 9002: 'input conversion to SPLIT16 (stem_pack / act_convert)'}
_TAG_IDS = {}


def layer_tag(name):
    if name is None:
        return 0
    t = _TAG_IDS.get(name)
    if t is None:
        t = _TAG_IDS[name] = len(_TAG_IDS) + 1
        TAG_NAMES[t + 1] = name                 # the flag holds tag + 1
    return t


class Split16RangeError(RuntimeError):
    """An activation left the range the SPLIT16 format can hold (|v| > 65504 or NaN): the f16x3 results of that forward are
    not valid; re-run it with precision 'f32'."""


def range_flag(reset=True):
    """(flag, layer name): flag 0 = every SPLIT16 tensor written since the last reset was in range.  Synchronises the device."""
    v = _lib.lib().srcnn_range_flag_read(1 if reset else 0)
    if v < 0:
        _lib.check(v, "srcnn_range_flag_read")
    return v, (TAG_NAMES.get(v, 'layer tag %d' % (v - 1)) if v else None)


PRECISION = 'f32'     # default engine: 'f32' (exact fp32 MFMA) or 'f16x3' (3-term split on the f16 MFMA)

# ---- per-shape launch-plan autotuning ("measure, don't guess"): every distinct conv shape is timed
# once on the device it runs on over the legal (tile, split-K) plans; the winner is cached.
AUTOTUNE = True
_TUNED = {}
_TUNE_LOG = {}       # key -> [(plan, ms)] of the last tuning run (dev tools print it)
# (tile_mr, tile_nr, waves, stages): workgroup tile (64*mr) x (64*nr), wavefronts, LDS ring depth
_CANDIDATES = [(2, 2, 4, 2), (2, 1, 4, 2), (1, 2, 4, 2), (1, 1, 4, 2)]
# extra variants of the SPLIT16 engine (csrc/conv_f16s.hip): 8-wave tiles and deeper DMA rings
# ((4, 4, 8, 2) = 256x256 on 8 waves of 64x128: 25 % fewer LDS fragment bytes per MFMA than the 64x64 per-wave tiles)
_CANDIDATES_F16S = [(4, 4, 8, 2), (4, 2, 8, 3), (2, 2, 8, 4), (2, 2, 8, 2), (2, 1, 4, 3), (1, 2, 4, 3), (1, 1, 4, 4)]


# Largest LDS footprint (KB per workgroup) a tuned plan may have.  Isolated-launch timing always favours the deepest ring /
# biggest tile (up to 144 KB: one workgroup per CU, nothing else fits beside it); with several pairs in flight a smaller
# footprint lets workgroups of OTHER launches share the CU and fill the matrix pipe's idle slots (profiles/lds_cap_r03.txt).
import os as _os
MAX_LDS_KB = int(_os.environ.get('SRCNN_MAX_LDS_KB', '160'))


ACT_SCALES = _os.environ.get('SRCNN_ACT_SCALES', '1') != '0'    # per-tensor power-of-two SPLIT16 activation scales (plan.calibrate)
LIMIT_TUNE_ROIS = 64          # row-limited launches (the lazy keypoint head) are tuned for this many units below the limit
RPN_PAIR_LAUNCH = _os.environ.get('SRCNN_RPN_PAIR', '1') != '0'     # A/B switch of the one-launch stereo RPN conv (conv mode 2)
# A/B switch: the projection shortcut of a layer's first block computed inside that block's conv3 (prep_conv_shortcut)
SHORTCUT_FUSION = _os.environ.get('SRCNN_SHORTCUT_FUSION', '1') != '0'
# A/B switches: the keypoint branch's 6-channel classifier computed inside the epilogue of the deconvolution -- '1' / 'valu' = round
# 4's fp32-FMA + DPP form (conv2d(head=...), srcnn_conv_desc.head_w), 'mfma' = as a second GEMM on the matrix pipe (conv2d(head2=...),
# srcnn_conv_desc.head_wf: the form the RPN head needs; for this 6-channel head on a K = 256 launch it is 10 % slower alone --
# 183 vs 166 us: the weight slice's load and four of eight waves computing -- and equal in the mix, profiles/kpts_head_form_r05.txt),
# '0' = two launches; and the stereo RPN's 24-channel head computed inside the RPN conv's epilogue as per-(eye, N tile) partial sums
KPTS_HEAD_FUSION = _os.environ.get('SRCNN_KPTS_HEAD_FUSION', '1')
KPTS_HEAD_FUSION = {'0': False, '1': 'valu'}.get(KPTS_HEAD_FUSION, KPTS_HEAD_FUSION)
RPN_HEAD_FUSION = _os.environ.get('SRCNN_RPN_HEAD_FUSION', '1') != '0'
# the five pyramid levels of the stereo RPN (shared RPN_Conv + head weights, stereo_rpn.py:73-95) in grouped launches
# (srcnn_conv2d_group): 'all' = one launch for P2..P6 behind the last smoothing conv, 'small' = P3..P6 in one launch (76 + 20 + 6 + 292
# tiles of 256x256: launches that cannot fill 256 CUs on their own) and P2 by itself, '0' = one launch per level.  Bit-identical.
RPN_GROUP = _os.environ.get('SRCNN_RPN_GROUP', 'small')
RPN_GROUP_TILE = tuple(int(c) for c in _os.environ.get('SRCNN_RPN_GROUP_TILE', '4482'))       # (tile_mr, tile_nr, waves, stages): 4482 or 2282
# A/B switch: the FPN top-down addition (_upsample_add) computed inside the lateral 1x1 conv's epilogue (conv2d(up=...),
# srcnn_conv_desc.up_top) whenever the lateral is launched inline behind its top map -- i.e. with several forwards in flight; a
# lone forward keeps the laterals on a side stream, early, and the separate srcnn_upsample_add.  Bit-identical either way.
# Built, tested, and NOT the default: three launches and 0.4 GB of traffic per pair fewer, and 0.2-0.4 % SLOWER four in flight
# (155.5 / 155.2 against 155.8 / 155.8 pairs/s, same box, profiles/upsample_fusion_ab_r05.txt) -- the mix is not short of HBM
# bandwidth, and the four bilinear taps lengthen the epilogue of a launch whose workgroups hold the matrix pipe's LDS.
UPSAMPLE_FUSION = _os.environ.get('SRCNN_UPSAMPLE_FUSION', '0') != '0'


# ---- what the tuner minimises.  'isolated': the latency of the launch alone on the chip (the right objective for one pair at a
# time).  'concurrent': the time per launch while TUNE_STREAMS copies of the launch run on as many HIP streams -- the regime of
# the headline benchmark and of pipeline.detect_3d_stream, where several batch-1 forwards are in flight and a plan is worth
# what it costs in CU-time, not in latency: a 150-workgroup plan that leaves 106 CUs to the neighbours beats a split-K plan
# that is 10 % faster alone but occupies the whole chip and needs a reduction launch (profiles/tune_objective_r04.txt).
# The two plan sets live side by side in _TUNED (the mode is part of the key).
TUNE_MODE = _os.environ.get('SRCNN_TUNE_MODE', 'isolated')
TUNE_STREAMS = int(_os.environ.get('SRCNN_TUNE_STREAMS', '3'))
_tune_side_streams = {}


def set_tune_mode(mode, streams=None):
    """'isolated' or 'concurrent' (see above).  Plans recorded into launch programs under the other mode are dropped by
    Plan.run (the mode is part of its epoch)."""
    global TUNE_MODE, TUNE_STREAMS
    assert mode in ('isolated', 'concurrent')
    TUNE_MODE = mode
    if streams is not None:
        TUNE_STREAMS = int(streams)


def tune_mode_key():
    return ('conc', TUNE_STREAMS) if TUNE_MODE == 'concurrent' and TUNE_STREAMS > 1 else ()


# Measurement hook (mix_table.py): [(compiled regex, n)] -- a conv launch whose layer name matches is issued n MORE times right
# behind itself (same arguments: the output is simply rewritten).  The step-time increase per extra launch is that layer's
# marginal cost INSIDE the several-forwards-in-flight mix, which no per-launch timing can give.  Empty in production.
REPEAT = []

# Measurement hook (tools/skip_probe.py): names of NON-conv stages of the forward that Plan leaves out while it records / runs
# ('maxpool', 'upsample_add', 'subsample', 'rpn_scores', 'proposals', 'roi_align', 'box_tail', 'kpts_tail') -- the buffers they would
# have written keep the previous frame's contents, so the step still runs on valid data; the step-time difference is what the
# stage costs INSIDE the several-forwards-in-flight mix.  Empty in production.
DEBUG_SKIP = frozenset()

# Bumped by whoever rewrites _TUNED under a model that already recorded launch programs (tune.tune_throughput, load_plans):
# every Plan compares it in run() and re-records.  KEY_HITS: while a dict, conv2d counts the launches per plan key.
PLAN_EPOCH = 0
KEY_HITS = None


def set_plan(key, plan):
    """Overwrite the tuned plan of one shape key; recorded launch programs are re-recorded on their next run."""
    global PLAN_EPOCH
    _TUNED[tuple(key)] = tuple(plan)
    PLAN_EPOCH += 1


def plan_lds_kb(mr, nr, waves, stages):
    return stages * 128 * 64 * (mr + nr) // 1024


def save_plans(path):
    """Write the tuned plans (shape key -> plan) as JSON: a later process on the same GPU model can skip the tuning launches."""
    import json
    with open(path, 'w') as f:
        json.dump([[list(k), list(v)] for k, v in _TUNED.items()], f)


def load_plans(path):
    """Adopt plans written by save_plans(); shapes not in the file are still tuned on first use.  Returns the count."""
    import json
    with open(path) as f:
        rows = json.load(f)
    global PLAN_EPOCH
    for k, v in rows:
        _TUNED[tuple(k)] = tuple(v)
    PLAN_EPOCH += 1
    return len(rows)


def _shape_key(cw, B, H, W, OH, OW, x_cstride, precision, fmts=(0, 0, 0)):
    return (precision, B, H, W, OH, OW, cw.cin, cw.cout, cw.kh, cw.kw, cw.stride, cw.pad, cw.mode, x_cstride) + tuple(fmts)


def _set_plan(d, plan):
    d.tile_mr, d.tile_nr, d.tile_waves, d.tile_stages, d.splits = plan


def _tune(d, key, device, only=None):
    """Times each candidate plan with HIP events on the current stream: three interleaved passes, then a play-off.
    only: the (mr, nr, waves, stages) tiles to choose from, unsplit (launches with an MFMA-form fused head)."""
    L = _lib.lib()
    M = d.B * d.OH * d.OW
    nkt = (d.KH * d.KW * d.Cin + d.Cin2) // 32
    cands = []
    tiles = list(_CANDIDATES)
    if d.precision == 1 and d.x_format == 1:
        tiles = _CANDIDATES_F16S + tiles
    if only is not None:
        tiles = [t for t in only if not (t[0] >= 4 and M < 256 * 4)]or list(only)[-1:]
    for mr, nr, waves, stages in tiles:
        if plan_lds_kb(mr, nr, waves, stages) > MAX_LDS_KB:
            continue
        if nr == 2 and d.Cout <= 64 and only is None:
            continue
        # the 256x256 tile: not with a second input (register budget); few fat workgroups lose the latency contest of this tuner on the
        # small-M layers but can win the several-in-flight step (tune.tune_throughput takes its candidates from this log)
        if nr == 4 and only is None and (d.Cout <= 128 or M < 256 * 8 or d.x2 or d.up_top):
            continue
        if mr >= 2 and M <= 64 * (mr // 2) and only is None:
            continue
        blocks = -(-M // (64 * mr)) * -(-d.Cout // (64 * nr))
        splits = [1]
        if d.mode != 1 and not d.x2 and not d.up_top and only is None:
            for s in (2, 3, 4, 6, 8, 12, 16):
                if blocks * s <= 4096 and nkt // s >= 4 and blocks < 1024:
                    splits.append(s)
        for s in splits:
            cands.append((mr, nr, waves, stages, s))
    log = _TUNE_LOG.setdefault(key, [])
    del log[:]
    st = _lib.stream()
    # the trial launches may run on whatever the buffers hold (a tensor last written in the other activation format reads as
    # NaNs): their range-guard reports go to a scratch word, not to the forward's
    global _tune_flag
    if _tune_flag is None or _tune_flag.device != device:
        _tune_flag = torch.zeros(1, dtype=torch.int32, device=device)
    bound = L.srcnn_range_flag_device_word()
    L.srcnn_range_flag_bind(_tune_flag.data_ptr())
    try:
        return _tune_candidates(d, key, device, cands, log, L, st)
    finally:
        L.srcnn_range_flag_bind(bound)


_tune_flag = None


def _tune_candidates(d, key, device, cands, log, L, st):
    conc = tune_mode_key()
    if conc:
        pool = _tune_side_streams.setdefault(str(device), [])
        while len(pool) < TUNE_STREAMS:
            pool.append(torch.cuda.Stream(device=device))
        side = pool[:TUNE_STREAMS]

    def timed_isolated(plan, launches):
        _set_plan(d, plan)
        ws = _lib.workspace(L.srcnn_conv2d_workspace_bytes(ctypes.byref(d)), device, "conv")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            _lib.check(L.srcnn_conv2d(ctypes.byref(d), ws.data_ptr(), ws.numel(), st), "srcnn_conv2d(tune)")
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / launches

    def timed_concurrent(plan, launches):
        """time per launch with the same launch running on TUNE_STREAMS streams at once (every stream its own split-K scratch;
        the copies read the same operands and write the same values to the same output)"""
        _set_plan(d, plan)
        need = L.srcnn_conv2d_workspace_bytes(ctypes.byref(d))
        wss = []
        for s in side:
            with torch.cuda.stream(s):
                wss.append(_lib.workspace(need, device, "conv"))
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for s in side:
            s.wait_event(e0)
        for _ in range(launches):
            for s, ws in zip(side, wss):
                _lib.check(L.srcnn_conv2d(ctypes.byref(d), ws.data_ptr(), ws.numel(), s.cuda_stream), "srcnn_conv2d(tune)")
        for s in side:
            cur.wait_stream(s)
        e1.record(cur)
        e1.synchronize()
        return e0.elapsed_time(e1) / (launches * len(side))

    timed = timed_concurrent if conc else timed_isolated

    # three passes over all candidates (a transient -- clock ramp, a neighbour's kernel -- then hits every plan once, not
    # one plan always), minimum per plan; the first launch of a plan is a warm-up
    best_of = {}
    for rnd in range(3):
        for plan in cands:
            if rnd == 0:
                timed(plan, 1)
            t = timed(plan, 2)
            if plan not in best_of or t < best_of[plan]:
                best_of[plan] = t
    # play-off between the three fastest
    finalists = sorted(best_of, key=best_of.get)[:3]
    for plan in finalists * 3:
        best_of[plan] = min(best_of[plan], timed(plan, 3))
    for plan in cands:
        log.append((plan, best_of[plan]))
    best = min(finalists, key=best_of.get)
    _TUNED[key] = best
    return best


def conv2d(cw, x, B, H, W, y, OH, OW, x_cstride=None, y_cstride=None, y_coffset=0, residual=None,
           res_cstride=None, x_offset_elems=0, relu=None, precision=None, x_fmt=0, y_fmt=0, res_fmt=0, plan=None, name=None,
           in_shift=0, out_shift=0, m_limit=None, m_limit_mul=0, x2=None, H2=0, W2=0, x2_cstride=None, head=None, head2=None, up=None,
           desc_only=False):
    """Launch the conv engine.  x / y / residual are device tensors (any shape; raw NHWC memory).
    *_fmt: _lib.FMT_F32 or _lib.FMT_SPLIT16 (f16x3 engine only; see include/srcnn_hip.h).
    in_shift / out_shift (f16x3 engine): the input tensor holds its values x 2^in_shift, the output (and the residual, which
    must carry the output's scale) is to be stored x 2^out_shift -- per-tensor power-of-two activation scales that keep
    SPLIT16 tensors in the middle of the f16 range (model/stereo_rcnn/plan.py: calibrate).  Exact: the factor goes into the
    epilogue's power-of-two rescale and a pre-scaled copy of the bias; ReLU commutes with it.
    m_limit (device int32 tensor) / m_limit_mul: only rows m < m_limit[0] * m_limit_mul are needed (srcnn_conv_desc.m_limit).
    x2 / H2 / W2 (weights from prep_conv_shortcut): the second input (B, H2, W2, cin2) SPLIT16, stored with the SAME scale as x.
    head = (cw_head, y_head) (f16x3 SPLIT16 engine): a 6-channel 1x1 conv applied to the activated output pixels inside this
    launch's epilogue (srcnn_conv_desc.head_w); y_head (pixels, 6) float32 receives it, y is not written (may be None).
    head2 = (cw_head, y_head, parts) (f16x3 SPLIT16 engine): the MFMA form of such a head (srcnn_conv_desc.head_wf), up to 24
    channels.  parts = 0: final (256-channel pixels; bias added; y_head (pixels, n) float32); parts > 0: y_head (parts, pixels, n)
    float32 receives one plane of partial sums per (eye, N tile) of the launch -- the caller adds them and the bias.
    up = (top, TH, TW, top_fmt) (f16x3 SPLIT16 engine): the FPN top-down addition inside this launch (srcnn_conv_desc.up_top):
    y = bilinear_align_corners(top (B, TH, TW, cout) -> (OH, OW)) + (conv + bias), top stored with the OUTPUT's scale; bit-identical
    to conv2d (float32 y) followed by upsample_add.
    desc_only (with an explicit plan): nothing is launched or counted; returns the filled descriptor (conv_chain / conv_group).
    Returns the plan the launch ran with: (tile_mr, tile_nr, waves, stages, splits)."""
    L = _lib.lib()
    d = _lib.ConvDesc()
    d.x = x.data_ptr() + 4 * x_offset_elems
    precision = PRECISION if precision is None else precision
    if precision == 'f16x3':
        cw.split_f16x3()
        d.w, d.w_lo, d.w_inv_scale, d.precision = cw.w_hi.data_ptr(), cw.w_lo.data_ptr(), cw.inv_scale * 2.0 ** (out_shift - in_shift), 1
    else:
        assert in_shift == 0 and out_shift == 0, "activation scales belong to the f16x3 engine's SPLIT16 tensors"
        d.w, d.w_lo, d.w_inv_scale, d.precision = cw.weight.data_ptr(), None, 1.0, 0
    bias = cw.bias
    if bias is not None and out_shift:
        cache = cw.__dict__.setdefault('_bias_shifted', {})
        bias = cache.get(out_shift)
        if bias is None:
            bias = cache[out_shift] = (cw.bias * 2.0 ** out_shift).contiguous()
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.y = y.data_ptr() if y is not None else None
    if head is not None:
        hcw, hy = head
        assert precision == 'f16x3' and x_fmt == _lib.FMT_SPLIT16 and hcw.cout == 6 and hcw.kh == 1 and hcw.kw == 1 and hcw.bias is not None
        assert hy.dtype == torch.float32 and hy.is_contiguous()
        d.head_w, d.head_bias, d.head_y = hcw.weight.data_ptr(), hcw.bias.data_ptr(), hy.data_ptr()
        d.head_cout, d.head_scale = 6, 2.0 ** -out_shift
        y_fmt = _lib.FMT_F32
    if head2 is not None:
        hcw, hy, parts = head2
        assert head is None and precision == 'f16x3' and x_fmt == _lib.FMT_SPLIT16 and hcw.kh == 1 and hcw.kw == 1
        assert hy.dtype == torch.float32 and hy.is_contiguous()
        frag, rows = head_fragments(hcw)
        d.head_wf, d.head_rows, d.head_cout, d.head_parts = frag.data_ptr(), rows, hcw.cout, int(parts)
        d.head_plane = hy.numel() // parts if parts else 0
        d.head_bias = hcw.bias.data_ptr() if (hcw.bias is not None and not parts) else None
        d.head_y, d.head_scale = hy.data_ptr(), hcw.inv_scale * 2.0 ** -out_shift
        y_fmt = _lib.FMT_F32
    d.B, d.H, d.W, d.Cin = B, H, W, cw.cin
    d.x_cstride = cw.cin if x_cstride is None else x_cstride
    d.OH, d.OW, d.Cout = OH, OW, cw.cout
    d.KH, d.KW, d.stride, d.pad = cw.kh, cw.kw, cw.stride, cw.pad
    cq = cw.cout // 4 if cw.mode == 1 else cw.cout
    d.y_cstride = cq if y_cstride is None else y_cstride
    d.y_coffset = y_coffset
    d.res_cstride = cw.cout if res_cstride is None else res_cstride
    d.relu = cw.relu if relu is None else int(relu)
    d.mode = cw.mode
    d.x_format, d.y_format, d.res_format = x_fmt, y_fmt, res_fmt
    d.layer_tag = layer_tag(name)
    assert (x2 is not None) == (cw.cin2 > 0), "weights from prep_conv_shortcut need the second input, and only they take one"
    if x2 is not None:
        assert precision == 'f16x3' and x_fmt == _lib.FMT_SPLIT16, "the second input belongs to the SPLIT16 engine"
        d.x2, d.Cin2, d.H2, d.W2, d.stride2 = x2.data_ptr(), cw.cin2, H2, W2, cw.stride2
        d.x2_cstride = cw.cin2 if x2_cstride is None else x2_cstride
    if up is not None:
        top, th, tw, top_fmt = up
        assert precision == 'f16x3' and x_fmt == _lib.FMT_SPLIT16 and residual is None and head is None and head2 is None and cw.mode == 0
        d.up_top, d.up_format, d.up_H, d.up_W = top.data_ptr(), int(top_fmt), int(th), int(tw)
        if plan is not None and tuple(plan) != (0, 0, 0, 0, 0):
            # the addition lives in the conv kernel's epilogue of every tile but 256x256, never in the split-K reduction: an explicit
            # plan is rewritten here (not silently replaced by the library's heuristic), so that the returned plan is the one that ran
            plan = ((4, 2, 8, 3) if tuple(plan[:2]) == (4, 4) else tuple(plan[:4])) + (1,)
    if desc_only:
        assert plan is not None and head is None and m_limit is None
        _set_plan(d, plan)
        return d
    if FlopCounter.enabled:
        FlopCounter.flops += 2.0 * B * OH * OW * cw.cout * cw.alg_k
        FlopCounter.launches += 1
        px_in = B * OH * OW if (cw.kh == 1 and cw.kw == 1) else B * H * W
        nbytes = 4.0 * (px_in * cw.cin + B * OH * OW * cw.cin2 + cw.cout * cw.alg_k + B * OH * OW * cw.cout * (2 if residual is not None else 1)
                        + (B * up[1] * up[2] * cw.cout if up is not None else 0))
        FlopCounter.bytes += nbytes
        if FlopCounter.rows is not None:
            FlopCounter.rows.append({'name': name or 'conv %dx%d %d->%d' % (cw.kh, cw.kw, cw.cin, cw.cout), 'M': B * OH * OW,
                                     'N': cw.cout, 'K': cw.alg_k, 'flops': 2.0 * B * OH * OW * cw.cout * cw.alg_k, 'bytes': nbytes})
    if head2 is not None and FlopCounter.enabled:
        # the head's own flops; its output (final: hn floats per pixel; partial: one plane per 256 conv channels) instead of y
        hn, taps = head2[0].cout, (4 if cw.mode == 1 else 1)
        fl = 2.0 * B * OH * OW * hn * cw.cout                       # (mode 1: 4 taps x Cout / 4 channels each)
        out_b = 4.0 * B * OH * OW * (taps * hn * (max(1, cw.cout // 256) if head2[2] else 1) - cw.cout)
        FlopCounter.flops += fl
        FlopCounter.bytes += out_b
        if FlopCounter.rows is not None:
            FlopCounter.rows[-1]['flops'] += fl
            FlopCounter.rows[-1]['bytes'] = nbytes + out_b
    if head2 is not None and plan is None and AUTOTUNE:
        # final form: the 256x256 tile owns the pixel's channels -- nothing to tune; partial form: that or the 128x128 8-wave tile
        if not head2[2]:
            plan = (4, 4, 8, 2, 1)
        else:
            key = _shape_key(cw, B, H, W, OH, OW, d.x_cstride, precision, (x_fmt, y_fmt, res_fmt) + ('head2', d.head_rows) + tune_mode_key())
            plan = _TUNED.get(key)
            if KEY_HITS is not None:
                KEY_HITS[key] = KEY_HITS.get(key, 0) + 1
            if plan is None:
                if torch.cuda.is_current_stream_capturing() or _lib.lib().srcnn_program_recording():
                    plan = (4, 4, 8, 2, 1) if B * OH * OW >= 2048 else (2, 2, 8, 2, 1)
                else:
                    plan = _tune(d, key, x.device, only=[(4, 4, 8, 2), (2, 2, 8, 2)])
    if head is not None:                   # the fused head lives in the 256x256 tile
        _set_plan(d, (4, 4, 8, 2, 1))
        if FlopCounter.enabled:            # the head's own (VALU) flops and its output instead of y
            FlopCounter.flops += 2.0 * B * OH * OW * (4 if cw.mode == 1 else 1) * 6 * (cw.cout // (4 if cw.mode == 1 else 1))
            FlopCounter.bytes += 4.0 * B * OH * OW * ((4 if cw.mode == 1 else 1) * 6 - cw.cout)
            if FlopCounter.rows is not None:
                FlopCounter.rows[-1]['flops'] = 2.0 * B * OH * OW * (cw.cout * cw.alg_k + 6 * cw.cout)
                FlopCounter.rows[-1]['bytes'] = nbytes + 4.0 * B * OH * OW * ((4 if cw.mode == 1 else 1) * 6 - cw.cout)
    elif plan is not None:                   # explicit (tile_mr, tile_nr, waves, stages, splits): tests and tools
        _set_plan(d, plan)
    elif AUTOTUNE:
        # a launch with a device-side row limit is tuned WITH a typical limit (LIMIT_TUNE_ROIS of its units): what is fastest for
        # the whole shape (the biggest tile, one round of CUs) is not what is fastest for a fifth of it
        key = _shape_key(cw, B, H, W, OH, OW, d.x_cstride, precision, (x_fmt, y_fmt, res_fmt) + (('lim', m_limit_mul) if m_limit is not None else ())
                         + (('x2', cw.cin2, cw.stride2, H2, W2) if x2 is not None else ()) + (('up',) if up is not None else ()) + tune_mode_key())
        plan = _TUNED.get(key)
        if KEY_HITS is not None:
            KEY_HITS[key] = KEY_HITS.get(key, 0) + 1
        if plan is None:
            if torch.cuda.is_current_stream_capturing() or _lib.lib().srcnn_program_recording():
                plan = (0, 0, 0, 0, 0)    # never time inside a graph capture / program recording; warm-up runs tune first
            else:
                if m_limit is not None:
                    typical = torch.tensor([LIMIT_TUNE_ROIS], dtype=torch.int32, device=x.device)
                    d.m_limit, d.m_limit_mul = typical.data_ptr(), int(m_limit_mul)
                plan = _tune(d, key, x.device)
                d.m_limit, d.m_limit_mul = None, 0
        _set_plan(d, plan)
    if FlopCounter.enabled and FlopCounter.rows is not None:
        FlopCounter.rows[-1]['plan'] = (d.tile_mr, d.tile_nr, d.tile_waves, d.tile_stages, d.splits)
    if m_limit is not None:
        assert m_limit.is_cuda and m_limit.dtype == torch.int32 and m_limit_mul > 0
        d.m_limit, d.m_limit_mul = m_limit.data_ptr(), int(m_limit_mul)
    need = L.srcnn_conv2d_workspace_bytes(ctypes.byref(d))
    ws = _lib.workspace(need, x.device, "conv")
    _lib.check(L.srcnn_conv2d(ctypes.byref(d), ws.data_ptr(), ws.numel(), _lib.stream()), "srcnn_conv2d")
    used = (d.tile_mr, d.tile_nr, d.tile_waves, d.tile_stages, d.splits)
    if REPEAT and name:
        for rx, n in REPEAT:
            if rx.match(name):
                for _ in range(n):
                    _lib.check(L.srcnn_conv2d(ctypes.byref(d), ws.data_ptr(), ws.numel(), _lib.stream()), "srcnn_conv2d(repeat)")
    return used


# ---- chained bottleneck launches (csrc/conv_chain.hip: srcnn_conv2d_chain): [conv2 -> conv3 -> conv1 of the next block] as ONE launch
# whose workgroups keep their rows through the three convolutions.  BOTTLENECK_CHAIN: '0' (default) = off, '1' = on, 'auto' = on while
# several forwards are in flight.  Built, bit-identical (tests/test_conv_chain_gpu.py), measured, and NOT the default: the headline
# step is the same with and without it (profiles/chain_ab_r06.txt: 6.35-6.39 ms off, 6.35-6.41 ms with layer1 / layer2 / layer3
# chained on their best tiles, at 3, 4, 6 and 8 forwards in flight) -- the intermediates come back through the fabric all the
# same (a layer3 block's weights alone are 4.4 MB, the XCD's whole L2), and the package power limit prices the step by its
# MFMA work, which a chain does not change.  It removes 46 launches per forward.
BOTTLENECK_CHAIN = _os.environ.get('SRCNN_BOTTLENECK_CHAIN', '0')
# planes of the bottleneck (64, 128, 256, 512) -> (tile_mr, waves, stages, narrow nr, wide nr) or None = that layer keeps its
# separate launches.  layer4 (M = 2394: 10 workgroups of 256 rows) stays unchained.
CHAIN_TILES = {64: (2, 4, 2, 1, 1), 128: (4, 8, 3, 2, 2), 256: (4, 8, 3, 2, 2), 512: None}
CHAIN_MIN_WGS = 32        # a chain launch with fewer workgroups than this is not worth its latency


def chain_enabled():
    if BOTTLENECK_CHAIN == 'auto':
        from . import streams
        return streams.pairs_in_flight() > 1
    return BOTTLENECK_CHAIN not in ('0', '', 'off')


def chain_tile(planes, M):
    """The chain tile of a bottleneck with `planes` channels and M output rows, or None."""
    t = CHAIN_TILES.get(planes)
    if t is None or -(-M // (64 * t[0])) < CHAIN_MIN_WGS:
        return None
    return t


def conv_chain(phases, tile, name=None):
    """ONE launch for up to three convolutions over the same rows (srcnn_conv2d_chain): phases = [(args, kwargs)] exactly as they
    would be passed to conv2d, in order; phase i > 0 must be a 1x1 / stride 1 convolution of phase i-1's output.  tile = (tile_mr,
    waves, stages, narrow nr, wide nr): a phase runs on the wide N tile when its Cout fills it (Cout >= 64 * wide nr), else on the
    narrow one.  Results are bit-identical to the separate launches with the same tiles."""
    L = _lib.lib()
    mr, waves, stages, na, nb = tile
    descs = (_lib.ConvDesc * len(phases))()
    flops = nbytes = 0.0
    rows = []
    for i, (args, kw) in enumerate(phases):
        cw = args[0]
        nr = nb if cw.cout >= 64 * nb else na
        kw = dict(kw)
        pname = kw.pop('name', None)
        d = conv2d(*args, plan=(mr, nr, waves, stages, 1), desc_only=True, name=pname, **kw)
        ctypes.memmove(ctypes.byref(descs[i]), ctypes.byref(d), ctypes.sizeof(d))
        B, OH, OW = args[2], args[6], args[7]
        M = B * OH * OW
        fl = 2.0 * M * cw.cout * cw.alg_k
        # compulsory bytes of the chain: the first phase's input, every weight, residuals / second inputs, every output once --
        # what a later phase reads is what the workgroup has just written (L2)
        px_in = M if (cw.kh == 1 and cw.kw == 1) else B * args[3] * args[4]
        by = 4.0 * ((px_in * cw.cin if i == 0 else 0) + M * cw.cin2 + cw.cout * cw.alg_k
                    + M * cw.cout * (2 if kw.get('residual') is not None else 1))
        flops += fl
        nbytes += by
        rows.append((pname, M, cw.cout, cw.alg_k))
    if FlopCounter.enabled:
        FlopCounter.flops += flops
        FlopCounter.bytes += nbytes
        FlopCounter.launches += 1
        if FlopCounter.rows is not None:
            FlopCounter.rows.append({'name': name or '+'.join(str(r[0]) for r in rows), 'M': rows[0][1], 'N': max(r[2] for r in rows),
                                     'K': sum(r[3] for r in rows), 'flops': flops, 'bytes': nbytes, 'plan': (mr, nb, waves, stages, 1),
                                     'chain': rows})
    _lib.check(L.srcnn_conv2d_chain(descs, len(phases), _lib.stream()), "srcnn_conv2d_chain")
    if REPEAT and name:
        for rx, n in REPEAT:
            if rx.match(name):
                for _ in range(n):
                    _lib.check(L.srcnn_conv2d_chain(descs, len(phases), _lib.stream()), "srcnn_conv2d_chain(repeat)")
    return tile


def conv_group(problems, tile, name=None):
    """ONE launch for up to five independent convolutions that share a tile (srcnn_conv2d_group): problems = [(args, kwargs)] as
    for conv2d; tile = (tile_mr, tile_nr, waves, stages).  Each result is bit-identical to its own launch with that tile."""
    L = _lib.lib()
    descs = (_lib.ConvDesc * len(problems))()
    flops = nbytes = 0.0
    rows = []
    for i, (args, kw) in enumerate(problems):
        cw = args[0]
        kw = dict(kw)
        pname = kw.pop('name', None)
        d = conv2d(*args, plan=tuple(tile) + (1,), desc_only=True, name=pname, **kw)
        ctypes.memmove(ctypes.byref(descs[i]), ctypes.byref(d), ctypes.sizeof(d))
        B, H, W, OH, OW = args[2], args[3], args[4], args[6], args[7]
        M = B * OH * OW
        fl = 2.0 * M * cw.cout * cw.alg_k
        px_in = M if (cw.kh == 1 and cw.kw == 1) else B * H * W
        by = 4.0 * (px_in * cw.cin + cw.cout * cw.alg_k + M * cw.cout * (2 if kw.get('residual') is not None else 1))
        h2 = kw.get('head2')
        if h2 is not None:                  # the head's own flops; its planes instead of y (as conv2d counts them)
            hn = h2[0].cout
            fl += 2.0 * M * hn * cw.cout
            by += 4.0 * M * (hn * (max(1, cw.cout // 256) if h2[2] else 1) - cw.cout)
        flops += fl
        nbytes += by
        rows.append((pname, M, cw.cout, cw.alg_k))
    if FlopCounter.enabled:
        FlopCounter.flops += flops
        FlopCounter.bytes += nbytes
        FlopCounter.launches += 1
        if FlopCounter.rows is not None:
            FlopCounter.rows.append({'name': name or '+'.join(str(r[0]) for r in rows), 'M': sum(r[1] for r in rows), 'N': rows[0][2],
                                     'K': rows[0][3], 'flops': flops, 'bytes': nbytes, 'plan': tuple(tile) + (1,), 'group': rows})
    _lib.check(L.srcnn_conv2d_group(descs, len(problems), _lib.stream()), "srcnn_conv2d_group")
    if REPEAT and name:
        for rx, n in REPEAT:
            if rx.match(name):
                for _ in range(n):
                    _lib.check(L.srcnn_conv2d_group(descs, len(problems), _lib.stream()), "srcnn_conv2d_group(repeat)")
    return tuple(tile) + (1,)


def preprocess_size(H, W, target_short=600):
    """(OH, OW, im_scale) of the reference's resize: im_scale = target / short side (demo.py:113-114), output size as
    cv2.resize derives it from fx/fy: cvRound(H*s), cvRound(W*s) -- Python's round() is the same ties-to-even."""
    scale = float(target_short) / float(min(H, W))
    return int(round(H * scale)), int(round(W * scale)), scale


def preprocess(img_rgb_u8, target_short=600, max_size=2484, packed=None, packed_fmt=0, planar=True):
    """A0 on the device: uint8 RGB (H, W, 3) device tensor -> (1, 3, OH, OW) float32 network input and im_scale
    (demo.py:103-129: RGB->BGR, -PIXEL_MEANS, cv2.resize INTER_LINEAR so that the short side is `target_short`;
    bit-equal to the OpenCV restatement oracle/preprocess.py).  `packed`: optional pre-allocated stem input buffer
    ((OH+6) x (OW+8) x 4 floats, see srcnn_stem_pack) written in the same pass in `packed_fmt` (the fused form);
    planar=False skips the float32 planes (detector-only callers that feed the stem through `packed`)."""
    assert img_rgb_u8.is_cuda and img_rgb_u8.dtype == torch.uint8 and img_rgb_u8.dim() == 3
    img = img_rgb_u8.contiguous()
    H, W = int(img.shape[0]), int(img.shape[1])
    OH, OW, scale = preprocess_size(H, W, target_short)
    out = torch.empty((1, 3, OH, OW), dtype=torch.float32, device=img.device) if planar else None
    _lib.check(_lib.lib().srcnn_preprocess(img.data_ptr(), H, W, scale, out.data_ptr() if planar else None, OH, OW,
                                           packed.data_ptr() if packed is not None else None, packed_fmt, _lib.stream()),
               "srcnn_preprocess")
    if planar and OW > max_size:                       # blob.py:59-61 (inert for KITTI)
        out = out[:, :, :, :max_size].contiguous()
    return out, scale


def stem_pack(im_nchw, out, batch_offset=0, out_fmt=0):
    """NCHW image -> zero-bordered NHWC4 for the stem conv; out_fmt FMT_SPLIT16 writes the same buffer in the layout the
    DMA conv engine reads (pass x_fmt=FMT_SPLIT16 to conv2d)."""
    B, C, H, W = im_nchw.shape
    assert C == 3
    L = _lib.lib()
    off = batch_offset * (H + 6) * (W + 8) * 4 * 4
    _lib.check(L.srcnn_stem_pack(_lib.ptr(im_nchw), B, H, W, out.data_ptr() + off, out_fmt, _lib.stream()),
               "srcnn_stem_pack")


def stem_pack_pair(left_nchw, right_nchw, out, out_fmt=0):
    """Both eyes of a stereo batch in one launch: out (2B, H+6, W+8, 4) = lefts, then rights."""
    B, C, H, W = left_nchw.shape
    assert C == 3 and tuple(right_nchw.shape) == tuple(left_nchw.shape)
    _lib.check(_lib.lib().srcnn_stem_pack_pair(_lib.ptr(left_nchw), _lib.ptr(right_nchw), B, H, W, out.data_ptr(), out_fmt, _lib.stream()),
               "srcnn_stem_pack_pair")


def maxpool3x3s2_ceil(x, B, H, W, C, y, OH, OW, y_fmt=0):
    _lib.check(_lib.lib().srcnn_maxpool3x3s2_ceil(x.data_ptr(), B, H, W, C, y.data_ptr(), OH, OW, y_fmt,
                                                  _lib.stream()), "srcnn_maxpool3x3s2_ceil")


def upsample_add(top, TH, TW, lateral, B, H, W, C, y, top_fmt=0, y_fmt=0):
    _lib.check(_lib.lib().srcnn_upsample_add(top.data_ptr(), TH, TW, lateral.data_ptr(), B, H, W, C, y.data_ptr(),
                                             top_fmt, y_fmt, _lib.stream()), "srcnn_upsample_add")


def act_convert(x, x_fmt, y_fmt):
    """NHWC activation tensor (.., C) F32 <-> SPLIT16 (same shape / byte size; the SPLIT16 tensor is an
    opaque float32-typed buffer)."""
    C = int(x.shape[-1])
    pixels = x.numel() // C
    y = torch.empty_like(x)
    _lib.check(_lib.lib().srcnn_act_convert(_lib.ptr(x), x_fmt, y.data_ptr(), y_fmt, pixels, C, _lib.stream()),
               "srcnn_act_convert")
    return y


def subsample2(x, B, H, W, C, y, OH, OW):
    _lib.check(_lib.lib().srcnn_subsample2(x.data_ptr(), B, H, W, C, y.data_ptr(), OH, OW, _lib.stream()),
               "srcnn_subsample2")


def nhwc_to_nchw(x_nhwc):
    """(B,H,W,C) contiguous -> new (B,C,H,W) contiguous tensor (API-edge helper)."""
    B, H, W, C = x_nhwc.shape
    y = torch.empty((B, C, H, W), dtype=x_nhwc.dtype, device=x_nhwc.device)
    _lib.check(_lib.lib().srcnn_nhwc_to_nchw(_lib.ptr(x_nhwc), B, H, W, C, y.data_ptr(), _lib.stream()),
               "srcnn_nhwc_to_nchw")
    return y


def nchw_to_nhwc(x_nchw):
    B, C, H, W = x_nchw.shape
    y = torch.empty((B, H, W, C), dtype=x_nchw.dtype, device=x_nchw.device)
    _lib.check(_lib.lib().srcnn_nchw_to_nhwc(_lib.ptr(x_nchw), B, C, H, W, y.data_ptr(), _lib.stream()),
               "srcnn_nchw_to_nhwc")
    return y
