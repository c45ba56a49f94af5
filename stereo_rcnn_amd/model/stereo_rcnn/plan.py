"""Static execution plan of the Stereo R-CNN inference forward on one MI355X.

MI355X-first structure (not the reference's eager module tree):
  * weights are re-laid-out once (frozen BN folded, K-contiguous GEMM rows) and stay resident;
  * left and right images run through the shared trunk/FPN as ONE batch of 2B images
    (the reference runs the Siamese halves sequentially, stereo_rcnn.py:155-185);
  * every activation lives in a pre-allocated NHWC buffer (288 GB of HBM: nothing is
    re-allocated per call), so the whole forward is a fixed list of asynchronous launches with
    no host synchronisation and can be captured in a hipGraph and replayed;
  * concatenations (RPN left|right, ROI left|right) are channel-offset writes, never copies.
Every arithmetic step is a call into libsrcnn_hip.so (see include/srcnn_hip.h).
"""
import ctypes
import logging

import torch

from ... import _lib, engine, streams

_log = logging.getLogger('stereo_rcnn_amd')
from ..utils.config import cfg


def _bn_dict(sd, prefix):
    return {k: sd[prefix + '.' + k] for k in ('weight', 'bias', 'running_mean', 'running_var')}


class Weights(object):
    """Engine-ready weights built from a reference-schema state_dict (CPU tensors)."""

    def __init__(self, sd, device):
        # re-layout / BN folding run on the device (load-time plumbing; float64 fold is cheap there)
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in sd.items()
              if not k.endswith('num_batches_tracked')}
        self.device = device
        # per-tensor power-of-two activation scales of the SPLIT16 engine: group name -> k (tensor stored x 2^k); empty until
        # Plan.calibrate() has seen one input (then fixed for the life of these weights, for every plan / slot)
        self.shifts = {}
        self.calibrated = False
        self.calibration_max = {}      # group -> largest |value| seen by the calibration forwards
        self.calib_epoch = 0           # bumped whenever the shifts change: plans drop their recorded programs / graphs
        # per layer: may the first block's conv3 take the block input as its second operand (both must carry one scale)
        self.fuse_shortcut = [True, True, True, True]
        self.stem = engine.prep_stem(sd['RCNN_layer0.0.weight'], _bn_dict(sd, 'RCNN_layer0.1'), device)
        self.layers = []
        for li in (1, 2, 3, 4):
            blocks = []
            b = 0
            while 'RCNN_layer%d.0.%d.conv1.weight' % (li, b) in sd:
                p = 'RCNN_layer%d.0.%d' % (li, b)
                stride = 2 if (b == 0 and li > 1) else 1          # stride on the first 1x1 (resnet.py:71)
                blk = {
                    'conv1': engine.prep_conv(sd[p + '.conv1.weight'], None, stride, 0, True, _bn_dict(sd, p + '.bn1'), device),
                    'conv2': engine.prep_conv(sd[p + '.conv2.weight'], None, 1, 1, True, _bn_dict(sd, p + '.bn2'), device),
                    'conv3': engine.prep_conv(sd[p + '.conv3.weight'], None, 1, 0, True, _bn_dict(sd, p + '.bn3'), device),
                    'down': None,
                }
                if p + '.downsample.0.weight' in sd:
                    blk['down'] = engine.prep_conv(sd[p + '.downsample.0.weight'], None, stride, 0, False,
                                                   _bn_dict(sd, p + '.downsample.1'), device)
                    # conv3 and the projection shortcut as one GEMM over [conv2's output | the block's input] (SPLIT16 engine)
                    blk['conv3_down'] = engine.prep_conv_shortcut(sd[p + '.conv3.weight'], _bn_dict(sd, p + '.bn3'),
                                                                  sd[p + '.downsample.0.weight'], _bn_dict(sd, p + '.downsample.1'),
                                                                  stride, device)
                blocks.append(blk)
                b += 1
            self.layers.append(blocks)

        def cb(name, stride=1, pad=0, relu=False):
            return engine.prep_conv(sd[name + '.weight'], sd[name + '.bias'], stride, pad, relu, None, device)

        self.toplayer = cb('RCNN_toplayer')
        self.smooth = [cb('RCNN_smooth%d' % i, 1, 1) for i in (1, 2, 3)]
        self.lateral = [cb('RCNN_latlayer%d' % i) for i in (1, 2, 3)]
        self.rpn_conv = cb('RCNN_rpn.RPN_Conv', 1, 1, True)
        # the same weights as ONE launch over [left images | right images] that writes [left 512 | right 512] per pixel
        self.rpn_conv_pair = engine.ConvW(self.rpn_conv.weight, self.rpn_conv.bias, 3, 3, 1, 1, True, mode=2)
        hw = torch.cat((sd['RCNN_rpn.RPN_cls_score.weight'], sd['RCNN_rpn.RPN_bbox_pred_left_right.weight']), 0)
        hb = torch.cat((sd['RCNN_rpn.RPN_cls_score.bias'], sd['RCNN_rpn.RPN_bbox_pred_left_right.bias']), 0)
        self.rpn_head = engine.prep_conv(hw, hb, 1, 0, False, None, device)
        # box head: 7x7/7 conv over the (7,7,512) NHWC roi tile == one GEMM row of 25088 (resnet.py:256-263)
        w0 = sd['RCNN_top.0.weight']                                  # (2048, 512, 7, 7)
        self.top0 = engine.ConvW(w0.permute(0, 2, 3, 1).contiguous().view(w0.shape[0], 1, 1, -1).to(device),
                                 sd['RCNN_top.0.bias'].to(device), 1, 1, 1, 0, True)
        self.top3 = cb('RCNN_top.3', 1, 0, True)
        self.fc = engine.prep_linear_stack(
            [sd['RCNN_bbox_pred.weight'], sd['RCNN_dim_orien_pred.weight'], sd['RCNN_cls_score.weight']],
            [sd['RCNN_bbox_pred.bias'], sd['RCNN_dim_orien_pred.bias'], sd['RCNN_cls_score.bias']], device)
        self.n_bbox = int(sd['RCNN_bbox_pred.weight'].shape[0])
        self.n_dim = int(sd['RCNN_dim_orien_pred.weight'].shape[0])
        self.n_cls = int(sd['RCNN_cls_score.weight'].shape[0])
        self.kpts = [cb('RCNN_kpts.%d' % i, 1, 1, True) for i in (0, 2, 4, 6, 8, 10)]
        self.kpts_up = engine.prep_deconv2x2(sd['RCNN_kpts.12.weight'], sd['RCNN_kpts.12.bias'], True, device)
        self.kpts_class = cb('kpts_class')


class Plan(object):
    """Buffers + launch list for a fixed (B, H, W) network input."""

    def __init__(self, weights, B, H, W):
        self.w = weights
        self.B, self.H, self.W = B, H, W
        dev = weights.device
        self.dev = dev
        N = 2 * B
        self.N = N
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self.im_left = e(B, 3, H, W)
        self.im_right = e(B, 3, H, W)
        self.im_info = e(B, 3)
        self.packed = e(N, H + 6, W + 8, 4)
        sh, sw = engine.conv_out_hw(H, W, 7, 7, 2, 3)
        self.stem_hw = (sh, sw)
        self.stem_out = e(N, sh, sw, 64)
        ph, pw = -(-(sh - 3) // 2) + 1, -(-(sw - 3) // 2) + 1           # ceil_mode (resnet.py:113)
        if (ph - 1) * 2 >= sh:
            ph -= 1
        if (pw - 1) * 2 >= sw:
            pw -= 1
        self.c1_hw = (ph, pw)
        self.c1 = e(N, ph, pw, 64)
        self.layer_hw, self.layer_bufs = [], []
        h, w_ = ph, pw
        for li, planes in enumerate((64, 128, 256, 512)):
            if li > 0:
                h, w_ = engine.conv_out_hw(h, w_, 1, 1, 2, 0)
            self.layer_hw.append((h, w_))
            self.layer_bufs.append({'m1': e(N, h, w_, planes), 'm1b': e(N, h, w_, planes), 'm2': e(N, h, w_, planes),
                                    'a': e(N, h, w_, 4 * planes), 'b': e(N, h, w_, 4 * planes)})
        self.c = [None] * 4
        # FPN
        (h2, w2), (h3, w3), (h4, w4), (h5, w5) = self.layer_hw
        self.p5 = e(N, h5, w5, 256)
        self.lat = [e(N, h4, w4, 256), e(N, h3, w3, 256), e(N, h2, w2, 256)]
        self.summed = [e(N, h4, w4, 256), e(N, h3, w3, 256), e(N, h2, w2, 256)]
        self.p4, self.p3, self.p2 = e(N, h4, w4, 256), e(N, h3, w3, 256), e(N, h2, w2, 256)
        h6, w6 = (h5 + 1) // 2, (w5 + 1) // 2
        self.p6 = e(N, h6, w6, 256)
        self.rpn_shapes = [(h2, w2), (h3, w3), (h4, w4), (h5, w5), (h6, w6)]
        self.A = sum(3 * a * b for a, b in self.rpn_shapes)
        self.rpn_cat = [e(B, a, b, 1024) for a, b in self.rpn_shapes]
        self.rpn_hd = [e(B, a, b, 24) for a, b in self.rpn_shapes]
        # fused RPN head (SPLIT16 engine): per level up to 8 planes of per-(eye, N tile) partial sums, (B * h * w, 24) floats each
        self.rpn_part = [torch.zeros((8, B * a * b, 24), dtype=torch.float32, device=dev) for a, b in self.rpn_shapes]   # zeroed: a plane count mismatch adds zeros, not garbage
        self.rpn_nparts = [0] * len(self.rpn_shapes)
        self._rpn_fused = False
        self.probs = e(B, self.A, 2)
        self.deltas = e(B, self.A, 6)
        self.post = cfg.TEST.RPN_POST_NMS_TOP_N
        R = B * self.post
        self.R = R
        self.rois_left = e(B, self.post, 5)
        self.rois_right = e(B, self.post, 5)
        self.num_valid = torch.empty((B,), dtype=torch.int32, device=dev)
        P = cfg.POOLING_SIZE
        self.sem = e(R, P, P, 512)
        self.h1 = e(R, 2048)
        self.h2 = e(R, 2048)
        self.fc = e(R, weights.fc.cout)
        self.cls_prob = e(R, weights.n_cls)
        self.bbox_pred = e(R, weights.n_bbox)          # contiguous copies of the fc output's regression columns (srcnn_box_head_tail)
        self.dim_orien = e(R, weights.n_dim)
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self.kp_in = e(R, 2 * P, 2 * P, 256)
        # zero-initialised: the lazy keypoint head (kpts_for_kept) computes the leading rows only, and the rows behind them
        # inside the last computed tile must hold finite numbers (stale results of earlier frames), never allocator garbage
        self.kp_a = z(R, 2 * P, 2 * P, 256)
        self.kp_b = z(R, 2 * P, 2 * P, 256)
        G = cfg.KPTS_GRID
        self.kp_up = z(R, G, G, 256)
        self.kp_logits = z(R, G, G, 6)
        self.kp_rois = z(self.post, 5)                    # the kept detections' rois, in keep order (lazy keypoint head)
        self.kp_prob = e(self.post, 4 * G)                # its outputs, kept order (the full head writes kpts_prob / left_prob / ...)
        self.kp_left = e(self.post, G)
        self.kp_right = e(self.post, G)
        self.kpts_prob = e(R, 4 * G)
        self.left_prob = e(R, G)
        self.right_prob = e(R, G)
        # this plan's own SPLIT16 range-flag word (srcnn_range_flag_bind): a forward in flight on another slot / stream never
        # sets or clears it, and it lives on the plan's device
        self.range_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._calib = None      # group -> max |value| while calibrate() runs
        self._epoch = (weights.calib_epoch, engine.tune_mode_key(), engine.PLAN_EPOCH)
        self._buf_shift = {}    # data_ptr -> shift of the tensor the buffer holds after the last run (as_f32 undoes it)
        self.graphs = {}
        self.programs = {}      # (precision, kpts, branches on side streams) -> (native launch program handle, buffers it keeps alive)
        self._rec = None        # program being recorded right now
        self.packed_fmt = -1    # format `packed` currently holds for the inputs of the NEXT run (-1: none, pack in trunk())
        self._src = (self.im_left, self.im_right)    # where trunk() packs the stem input from: this plan's own planes, or the caller's tensors (set_inputs)
        self.fmt = 0            # activation format of the internal buffers for the current/last run
        # independent branches of the forward (FPN laterals, small RPN levels, box head vs keypoint head) are
        # issued on side streams with event fork/join, so eager runs AND the captured hipGraph execute them
        # concurrently with the critical path instead of serialising many small launches
        # -- while ONE forward is in flight.  With several in flight the other forwards fill the chip, and side streams that
        # share hardware queues with other forwards' main streams stall them (streams.py): `overlap` None = follow
        # streams.branch_overlap(), True / False = forced (tools, the per-layer timing of bench.py)
        self.overlap = None
        self._side = None

    def _par(self):
        """branches on side streams in this run?"""
        want = streams.branch_overlap() if self.overlap is None else bool(self.overlap)
        return want and self.side[0] is not None           # SRCNN_SIDE_STREAMS=none: nothing to fork onto, whatever was forced

    @property
    def side(self):
        if self._side is None:
            self._side = streams.side_streams(2, self.dev) or [None, None]
        return self._side

    # ------------------------------------------------------------------ activation scales (SPLIT16 engine)
    def _k(self, group):
        """Shift of a scale group in the CURRENT run: tensors of the group are stored x 2^k (0 in F32 mode / before calibration)."""
        return self.w.shifts.get(group, 0) if (self.fmt and group is not None) else 0

    def _conv(self, cw, x, B, H, W, y, OH, OW, in_group, out_group, **kw):
        """engine.conv2d with the scale bookkeeping: the input tensor belongs to `in_group`, the output (and the residual) to
        `out_group` (None = an unscaled tensor: the image, the F32 results that leave the network)."""
        ko = self._k(out_group)
        used = engine.conv2d(cw, x, B, H, W, y, OH, OW, in_shift=self._k(in_group), out_shift=ko, **kw)
        if kw.get('head') is not None or kw.get('head2') is not None:   # fused head: y itself is not written (as_f32 / calibration must not read it)
            return used
        self._buf_shift[y.data_ptr()] = ko
        if self._calib is not None and out_group is not None:
            ycs = kw.get('y_cstride')
            if ycs is None:
                self._note(out_group, y)
            else:                          # a channel slice of a wider buffer: only what this launch wrote
                c0 = kw.get('y_coffset', 0)
                self._note(out_group, y.view(-1, ycs)[:, c0:c0 + cw.cout * (2 if cw.mode == 2 else 1)])
        return used

    def _conv_chain(self, phases, tile, name):
        """engine.conv_chain with _conv's scale bookkeeping per phase: phases = [((cw, x, B, H, W, y, OH, OW, in_group, out_group), kwargs)]."""
        ph = []
        for a, kw in phases:
            ko = self._k(a[9])
            ph.append((a[:8], dict(kw, in_shift=self._k(a[8]), out_shift=ko)))
            self._buf_shift[a[5].data_ptr()] = ko
        engine.conv_chain(ph, tile, name=name)
        if self._calib is not None:
            for a, kw in phases:
                if a[9] is not None:
                    self._note(a[9], a[5])

    def _note(self, group, t):
        self._calib[group] = max(self._calib.get(group, 0.0), float(t.abs().max()))

    def calibrate(self, merge=False):
        """Choose the power-of-two scale of every SPLIT16 tensor group from ONE forward of the inputs now in this plan (merge:
        together with what earlier calibration forwards saw -- _StereoRCNN.calibrate_activation_scales), run on
        the exact fp32 engine: k = round(log2(2^11 / max|v|)), so that the group's largest value sits 32x below the f16
        overflow threshold (the range guard still watches it) and its `lo` halves stay normal down to values 2^13 times
        smaller than the largest.  Power-of-two factors are exact: they travel in the weights' epilogue rescale and the
        bias; ReLU, max-pool, ROIAlign's averages and the nearest / bilinear up-sampling commute with them.  The residual
        stream of a layer, and the whole feature pyramid (mixed per roi by ROIAlign), share one scale each."""
        import math
        w = self.w
        saved = (self.fmt, engine.PRECISION, self.overlap)
        self.fmt, engine.PRECISION, self.overlap = _lib.FMT_F32, 'f32', False
        self._calib = {}
        try:
            self.launch_all()
            torch.cuda.synchronize()
        finally:
            calib, self._calib = self._calib, None
            self.fmt, engine.PRECISION, self.overlap = saved
        if merge:
            for g, mx in w.calibration_max.items():
                calib[g] = max(calib.get(g, 0.0), mx)
        if not (calib.get('stem', 0.0) > 0 and math.isfinite(calib.get('stem', 0.0))):
            # a blank (or non-finite) frame says nothing about the network's ranges: scales chosen from it would make every real
            # frame trip the range guard and fall back to the fp32 engine.  Stay uncalibrated; the next forward tries again.
            _log.warning('SPLIT16 activation scales NOT calibrated: the frame produced no finite non-zero stem output '
                         '(blank warm-up input?); the next forward calibrates again')
            self.packed_fmt = -1
            return
        shifts = {}
        for g, mx in calib.items():
            if mx > 0 and math.isfinite(mx):
                shifts[g] = int(max(-24, min(24, round(math.log2(2048.0 / mx)))))
        # conv3 of a layer's first block reads conv2's output AND the block's input in one GEMM (prep_conv_shortcut): one scale
        # for both.  conv2's output is private to the block, so it takes the input stream's scale -- when that costs at most 4
        # bits of its headroom below the f16 range (of 5) or 6 bits of its low end; otherwise that block keeps two launches.
        fuse = []
        for li in range(1, 5):
            gm, gx = 'L%d.0.m2' % li, ('stem' if li == 1 else 'L%d' % (li - 1))
            ok = gm in shifts and gx in shifts and -6 <= shifts[gx] - shifts[gm] <= 4
            if ok:
                shifts[gm] = shifts[gx]
            fuse.append(bool(ok))
        if shifts != w.shifts or fuse != w.fuse_shortcut:
            w.fuse_shortcut = fuse
            w.shifts = shifts
            w.calib_epoch += 1             # launch programs / graphs recorded with the old scales are stale (every plan checks)
        w.calibrated = True
        w.calibration_max = calib
        w.calibration_frames = (getattr(w, 'calibration_frames', 0) + 1) if merge else 1
        self.packed_fmt = -1
        _log.info('SPLIT16 activation scales calibrated on %d frame(s) (last: %dx%dx%d, max |stem| %.3g): shifts %s; projection '
                  'shortcuts fused: %s', w.calibration_frames, self.B, self.H, self.W, calib.get('stem', 0.0),
                  ' '.join('%s:%+d' % kv for kv in sorted(shifts.items())), fuse)

    # ------------------------------------------------------------------ stages
    def trunk(self):
        w, N = self.w, self.N
        H, W = self.H, self.W
        f = self.fmt                                  # SPLIT16 activations (f16x3 engine) or F32
        if self.packed_fmt != f:                      # set_images() / pack_inputs() already wrote the stem input in this format otherwise
            engine.stem_pack_pair(self._src[0], self._src[1], self.packed, out_fmt=f)
        self.packed_fmt = -1                          # consumed: the next forward packs again unless set_images() / pack_inputs() ran
        sh, sw = self.stem_hw
        self._conv(w.stem, self.packed, N, H + 6, W + 8, self.stem_out, sh, sw, None, 'stem', x_cstride=4, x_fmt=f, name='stem')   # F32 out
        ph, pw = self.c1_hw
        if 'maxpool' not in engine.DEBUG_SKIP:
            engine.maxpool3x3s2_ceil(self.stem_out, N, sh, sw, 64, self.c1, ph, pw, y_fmt=f)
        self._buf_shift[self.c1.data_ptr()] = self._k('stem')
        x, xh, xw, xg = self.c1, ph, pw, 'stem'
        for li, blocks in enumerate(w.layers):
            h, w_ = self.layer_hw[li]
            bufs = self.layer_bufs[li]
            cur, nxt = bufs['a'], bufs['b']
            lg = 'L%d' % (li + 1)                         # the residual stream of this layer: ONE scale for all its blocks
            # A block = conv1 -> conv2 (3x3) -> conv3 (+ residual or projection shortcut).  Launch order, either way: conv1 of the
            # layer's first block, then per block [conv2, conv3, conv1 of the NEXT block] -- as three launches, or (SPLIT16 engine,
            # several forwards in flight: engine.chain_enabled) as ONE chained launch whose workgroups keep their rows through the
            # three convolutions (csrc/conv_chain.hip).  conv1 writes the m1 buffers alternately: a chain's last phase must not
            # overwrite the tensor its first phase reads (other workgroups' halo rows).
            planes = blocks[0]['conv2'].cout
            tile = engine.chain_tile(planes, N * h * w_) if (f and engine.chain_enabled()) else None
            m1 = [bufs['m1'], bufs['m1b']]
            g1 = lg + '.0.m1'
            self._conv(blocks[0]['conv1'], x, N, xh, xw, m1[0], h, w_, xg, g1, x_fmt=f, y_fmt=f, name='layer%d.0.conv1' % (li + 1))
            for bi, blk in enumerate(blocks):
                nm = 'layer%d.%d.' % (li + 1, bi)
                g2 = lg + '.%d.m2' % bi
                phs = [((blk['conv2'], m1[bi & 1], N, h, w_, bufs['m2'], h, w_, g1, g2), dict(x_fmt=f, y_fmt=f, name=nm + 'conv2'))]
                if blk['down'] is not None and f and engine.SHORTCUT_FUSION and w.fuse_shortcut[li] and self._k(g2) == self._k(xg):
                    # the projection shortcut inside conv3: K = [conv2's output | the block's input], no residual round trip
                    phs.append(((blk['conv3_down'], bufs['m2'], N, h, w_, cur, h, w_, g2, lg),
                               dict(x2=x, H2=xh, W2=xw, x_fmt=f, y_fmt=f, name=nm + 'conv3+downsample')))
                else:
                    if blk['down'] is not None:
                        self._conv(blk['down'], x, N, xh, xw, nxt, h, w_, xg, lg, x_fmt=f, y_fmt=f, name=nm + 'downsample')
                        res = nxt
                    else:
                        res = x
                    phs.append(((blk['conv3'], bufs['m2'], N, h, w_, cur, h, w_, g2, lg),
                               dict(residual=res, x_fmt=f, y_fmt=f, res_fmt=f, name=nm + 'conv3')))
                if bi + 1 < len(blocks):
                    g1 = lg + '.%d.m1' % (bi + 1)
                    phs.append(((blocks[bi + 1]['conv1'], cur, N, h, w_, m1[(bi + 1) & 1], h, w_, lg, g1),
                               dict(x_fmt=f, y_fmt=f, name='layer%d.%d.conv1' % (li + 1, bi + 1))))
                if tile is not None:
                    self._conv_chain(phs, tile, nm + 'chain')
                else:
                    for a, kw in phs:
                        self._conv(*a, **kw)
                x, xh, xw, xg = cur, h, w_, lg
                cur, nxt = nxt, cur
            self.c[li] = x

    # ---- fork / join helpers: event record + stream wait, eagerly (torch; also what a hipGraph capture sees) or, while a
    # launch program records (run(use_program=True)), as nodes of that program
    def _signal(self, stream):
        """Marks the current end of `stream`; returns a token for _wait()."""
        if self._rec is not None:
            ev = _lib.lib().srcnn_program_record_event(self._rec, stream.cuda_stream)
            if ev < 0:
                _lib.check(ev, "srcnn_program_record_event")
            return ev
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def _wait(self, stream, token):
        if self._rec is not None:
            _lib.check(_lib.lib().srcnn_program_wait_event(self._rec, stream.cuda_stream, token), "srcnn_program_wait_event")
        else:
            stream.wait_event(token)

    def _fork(self, side):
        self._wait(side, self._signal(torch.cuda.current_stream()))

    def _join(self, side):
        self._wait(torch.cuda.current_stream(), self._signal(side))

    def _rpn_level(self, l):
        """RPN_Conv on left and right maps of level l into [left 512 | right 512], fused 1x1 heads, scoring."""
        w, B, f = self.w, self.B, self.fmt
        feats = [self.p2, self.p3, self.p4, self.p5, self.p6]
        h, w_ = self.rpn_shapes[l]
        cat, hd = self.rpn_cat[l], self.rpn_hd[l]
        if self._rpn_fused:
            # RPN_Conv on both eyes AND the 24-channel head (RPN_cls_score | RPN_bbox_pred_left_right over [left 512 | right 512],
            # stereo_rpn.py:77-91) in one launch: the head runs in the conv's epilogue as a second GEMM on each tile and leaves
            # partial sums per (eye, N tile); the (B, h, w, 1024) tensor is neither written nor read, five head launches are gone
            used = self._conv(w.rpn_conv_pair, feats[l], 2 * B, h, w_, None, h, w_, 'P', 'rpn', x_fmt=f, name='rpn_conv+head.P%d' % (l + 2),
                              head2=(w.rpn_head, self.rpn_part[l], 8))
            # planes the launch wrote = (eye, N tile) pairs of the tile that RAN: the library runs a head2 launch on the 128x128
            # 8-wave tile when asked for exactly that (or, without a request, for small M), on the 256x256 tile otherwise
            # (conv_mfma.hip: plan_for) -- whatever else the descriptor asked for
            ran_small = (used[0] == 2 and used[1] == 2) or (used[0] <= 0 and 2 * B * h * w_ < 256 * 8)
            self.rpn_nparts[l] = 2 * (w.rpn_conv_pair.cout // (128 if ran_small else 256))
            return
        if f and engine.RPN_PAIR_LAUNCH:       # SPLIT16 engine: both eyes in one launch (conv mode 2: the right half lands 512 channels further)
            self._conv(w.rpn_conv_pair, feats[l], 2 * B, h, w_, cat, h, w_, 'P', 'rpn', y_cstride=1024, y_coffset=0, x_fmt=f, y_fmt=f,
                       name='rpn_conv.P%d' % (l + 2))
        else:
            self._conv(w.rpn_conv, feats[l], B, h, w_, cat, h, w_, 'P', 'rpn', y_cstride=1024, y_coffset=0, x_fmt=f, y_fmt=f,
                       name='rpn_conv.P%d' % (l + 2))
            self._conv(w.rpn_conv, feats[l], B, h, w_, cat, h, w_, 'P', 'rpn', y_cstride=1024, y_coffset=512,
                       x_offset_elems=B * h * w_ * 256, x_fmt=f, y_fmt=f, name='rpn_conv.P%d' % (l + 2))
        self._conv(w.rpn_head, cat, B, h, w_, hd, h, w_, 'rpn', None, x_fmt=f, name='rpn_head.P%d' % (l + 2))

    def _rpn_group(self, levels):
        """RPN_Conv + fused head of several pyramid levels in ONE grouped launch (shared weights; srcnn_conv2d_group): every level's
        partial planes are the bits its own launch would write."""
        w, B, f = self.w, self.B, self.fmt
        feats = [self.p2, self.p3, self.p4, self.p5, self.p6]
        tile = engine.RPN_GROUP_TILE
        ki, ko = self._k('P'), self._k('rpn')
        probs = []
        for l in levels:
            h, w_ = self.rpn_shapes[l]
            probs.append(((w.rpn_conv_pair, feats[l], 2 * B, h, w_, None, h, w_),
                          dict(x_fmt=f, head2=(w.rpn_head, self.rpn_part[l], 8), in_shift=ki, out_shift=ko, precision=engine.PRECISION,
                               name='rpn_conv+head.P%d' % (l + 2))))
            self.rpn_nparts[l] = 2 * (w.rpn_conv_pair.cout // (64 * tile[1]))
        engine.conv_group(probs, tile, name='rpn_conv+head.P%s' % '+'.join(str(l + 2) for l in levels))

    def _rpn_scores(self):
        """Pair softmax + NHWC flatten of every level's head output into probs / deltas (stereo_rpn.py:81-91): one launch for all
        five levels, after the last of them (on whatever stream it ran) has been joined."""
        nl = len(self.rpn_shapes)
        hw = (ctypes.c_int * nl)(*[a * b for a, b in self.rpn_shapes])
        if 'rpn_scores' in engine.DEBUG_SKIP:
            return
        if self._rpn_fused:
            parts = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.rpn_part])
            npl = (ctypes.c_int * nl)(*self.rpn_nparts)
            planes = (ctypes.c_longlong * nl)(*[t.numel() // 8 for t in self.rpn_part])
            _lib.check(_lib.lib().srcnn_rpn_score_parts(parts, npl, planes, hw, nl, self.B, self.w.rpn_head.bias.data_ptr(), self.probs.data_ptr(),
                                                        self.deltas.data_ptr(), self.A, _lib.stream()), "srcnn_rpn_score_parts")
            return
        heads = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.rpn_hd])
        _lib.check(_lib.lib().srcnn_rpn_score_levels(heads, hw, nl, self.B, 24, self.probs.data_ptr(), self.deltas.data_ptr(), self.A,
                                                     _lib.stream()), "srcnn_rpn_score_levels")

    def fpn_rpn(self):
        """FPN top-down path (stereo_rcnn.py:161-168) with the stereo RPN head (stereo_rpn.py:73-95) of every
        level started as soon as that level exists: laterals on side stream 0, RPN levels 6..3 on side stream 1,
        the big P2 level on the main stream."""
        w, N, f = self.w, self.N, self.fmt
        (h2, w2), (h3, w3), (h4, w4), (h5, w5) = self.layer_hw
        c2, c3, c4, c5 = self.c
        self._rpn_fused = bool(f and engine.RPN_HEAD_FUSION and engine.RPN_PAIR_LAUNCH)
        par = self._par()
        s_lat, s_rpn = self.side if par else (None, None)
        lat_done = []
        if par:
            self._fork(s_lat)
            with torch.cuda.stream(s_lat):
                for i, (cin, (h, w_)) in enumerate(((c4, (h4, w4)), (c3, (h3, w3)), (c2, (h2, w2)))):
                    self._conv(w.lateral[i], cin, N, h, w_, self.lat[i], h, w_, 'L%d' % (3 - i), 'P', x_fmt=f,
                               name='fpn.lateral%d' % (i + 1))     # lateral stays F32 (at the pyramid's scale: it is added to `top`)
                    lat_done.append(self._signal(s_lat))
        self._conv(w.toplayer, c5, N, h5, w5, self.p5, h5, w5, 'L4', 'P', x_fmt=f, y_fmt=f, name='fpn.toplayer')
        h6, w6 = self.rpn_shapes[4]
        if 'subsample' not in engine.DEBUG_SKIP:
            engine.subsample2(self.p5, N, h5, w5, 256, self.p6, h6, w6)                          # stereo_rcnn.py:168
        self._buf_shift[self.p6.data_ptr()] = self._k('P')
        # grouped RPN launches (engine.RPN_GROUP): the levels wait for the last one of their group
        grp = engine.RPN_GROUP if (self._rpn_fused and engine.RPN_GROUP in ('all', 'small')) else None
        if grp:
            pass
        elif par:
            self._fork(s_rpn)
            with torch.cuda.stream(s_rpn):
                self._rpn_level(4)
                self._rpn_level(3)
        else:
            self._rpn_level(4)
            self._rpn_level(3)
        tops = [(self.p5, h5, w5), None, None]
        for i, (cin, (h, w_), out) in enumerate(((c4, (h4, w4), self.p4), (c3, (h3, w3), self.p3),
                                                 (c2, (h2, w2), self.p2))):
            top, th, tw = tops[i]
            if par:
                self._wait(torch.cuda.current_stream(), lat_done[i])
            elif f and engine.UPSAMPLE_FUSION:
                # lateral conv + top-down addition in one launch (srcnn_conv_desc.up_top): the float32 lateral map is neither written
                # nor read back, three launches and 0.4 GB per pair go; bit-identical to the two-launch form below
                self._conv(w.lateral[i], cin, N, h, w_, self.summed[i], h, w_, 'L%d' % (3 - i), 'P', x_fmt=f, y_fmt=f,
                           up=(top, th, tw, f), name='fpn.lateral%d' % (i + 1))
            else:
                self._conv(w.lateral[i], cin, N, h, w_, self.lat[i], h, w_, 'L%d' % (3 - i), 'P', x_fmt=f, name='fpn.lateral%d' % (i + 1))
            if (par or not (f and engine.UPSAMPLE_FUSION)) and 'upsample_add' not in engine.DEBUG_SKIP:
                engine.upsample_add(top, th, tw, self.lat[i], N, h, w_, 256, self.summed[i], top_fmt=f, y_fmt=f)   # stereo_rcnn.py:91-108
            if self._calib is not None:
                self._note('P', self.summed[i])
            self._conv(w.smooth[i], self.summed[i], N, h, w_, out, h, w_, 'P', 'P', x_fmt=f, y_fmt=f, name='fpn.smooth%d' % (i + 1))
            if i + 1 < 3:
                tops[i + 1] = (out, h, w_)
            level = 2 - i                        # p4 -> RPN level 2, p3 -> 1, p2 -> 0
            if grp:
                if grp == 'small' and level == 1:               # P3 exists: P3..P6 in one launch (on the side stream when forked)
                    if par:
                        self._fork(s_rpn)
                        with torch.cuda.stream(s_rpn):
                            self._rpn_group([1, 2, 3, 4])
                    else:
                        self._rpn_group([1, 2, 3, 4])
                elif level == 0:
                    if grp == 'all':
                        self._rpn_group([0, 1, 2, 3, 4])
                    else:
                        self._rpn_level(0)
            elif par and level > 0:
                self._fork(s_rpn)                # side stream also waits for this level's smooth conv
                with torch.cuda.stream(s_rpn):
                    self._rpn_level(level)
            else:
                self._rpn_level(level)
        if par:
            self._join(s_lat)
            self._join(s_rpn)
        self._rpn_scores()

    def fpn(self):
        prev, self.overlap = self.overlap, False
        try:
            self.fpn_rpn()
        finally:
            self.overlap = prev

    def rpn(self):
        """(kept for stage timing tools) the RPN head alone, serial."""
        self._rpn_fused = bool(self.fmt and engine.RPN_HEAD_FUSION and engine.RPN_PAIR_LAUNCH)
        if self._rpn_fused and engine.RPN_GROUP == 'all':
            self._rpn_group([0, 1, 2, 3, 4])
        elif self._rpn_fused and engine.RPN_GROUP == 'small':
            self._rpn_group([1, 2, 3, 4])
            self._rpn_level(0)
        else:
            for l in range(5):
                self._rpn_level(l)
        self._rpn_scores()

    def proposals(self):
        if 'proposals' in engine.DEBUG_SKIP:
            return
        L = _lib.lib()
        nl = len(self.rpn_shapes)
        hw = (ctypes.c_int * (2 * nl))(*[int(v) for s in self.rpn_shapes for v in s])
        pre = cfg.TEST.RPN_PRE_NMS_TOP_N
        ws = _lib.workspace(L.srcnn_proposal_workspace_bytes(self.B, self.A, pre, self.post), self.dev, "proposal")
        _lib.check(L.srcnn_proposal_layer(self.probs.data_ptr(), self.deltas.data_ptr(), self.B, self.A, hw, nl,
                                          self.im_info.data_ptr(), pre, self.post, float(cfg.TEST.RPN_NMS_THRESH),
                                          self.rois_left.data_ptr(), self.rois_right.data_ptr(),
                                          self.num_valid.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()),
                   "srcnn_proposal_layer")

    def _pyramid(self, right, rois, A, out, cstride, coffset, n_rois=None, limit=None):
        if 'roi_align' in engine.DEBUG_SKIP or ('roi_align%d' % A) in engine.DEBUG_SKIP:
            return
        maps = [self.p2, self.p3, self.p4, self.p5]
        hw = self.rpn_shapes[:4]
        ptrs = (ctypes.c_void_p * 4)()
        for l in range(4):
            h, w_ = hw[l]
            ptrs[l] = maps[l].data_ptr() + (4 * self.B * h * w_ * 256 if right else 0)
        mh = (ctypes.c_int * 4)(*[h for h, _ in hw])
        mw = (ctypes.c_int * 4)(*[w_ for _, w_ in hw])
        _lib.check(_lib.lib().srcnn_pyramid_roi_align(ptrs, mh, mw, 256, float(self.H), rois.data_ptr(), self.R if n_rois is None else n_rois, A,
                                                      out.data_ptr(), cstride, coffset, self.fmt, self.fmt,
                                                      None if limit is None else limit.data_ptr(), _lib.stream()), "srcnn_pyramid_roi_align")

    def box_head(self):
        w, R, f = self.w, self.R, self.fmt
        P = cfg.POOLING_SIZE
        self._pyramid(False, self.rois_left, P, self.sem, 512, 0)        # stereo_rcnn.py:248-249
        self._pyramid(True, self.rois_right, P, self.sem, 512, 256)
        self._conv(w.top0, self.sem, R, 1, 1, self.h1, 1, 1, 'P', 'h1', x_fmt=f, y_fmt=f, name='box.top0')   # 7x7/7 conv == GEMM (resnet.py:257)
        self._conv(w.top3, self.h1, R, 1, 1, self.h2, 1, 1, 'h1', 'h2', x_fmt=f, y_fmt=f, name='box.top3')
        self._conv(w.fc, self.h2, R, 1, 1, self.fc, 1, 1, 'h2', None, x_fmt=f, name='box.fc')
        if 'box_tail' not in engine.DEBUG_SKIP:
            _lib.check(_lib.lib().srcnn_box_head_tail(self.fc.data_ptr(), R, w.n_bbox, w.n_dim, w.n_cls, w.fc.cout, self.bbox_pred.data_ptr(),
                                                  self.dim_orien.data_ptr(), self.cls_prob.data_ptr(), _lib.stream()), "srcnn_box_head_tail")

    def kpts_head(self, rois=None, n_rois=None, limit=None, outs=None):
        """Keypoint branch (stereo_rcnn.py:260-271).  Default: all R rois of the forward.  rois / n_rois / limit: the lazy form
        -- `rois` holds n_rois rois of which only the first limit[0] (device int32) matter: every conv of the tower gets the
        device-side row limit, so the launch list is fixed and no host read-back is needed."""
        w, f = self.w, self.fmt
        R = self.R if n_rois is None else n_rois
        P = cfg.POOLING_SIZE
        lim = lambda mul: {} if limit is None else {'m_limit': limit, 'm_limit_mul': mul}
        self._pyramid(False, self.rois_left if rois is None else rois, 2 * P, self.kp_in, 256, 0, n_rois=R, limit=limit)   # stereo_rcnn.py:260
        x = self.kp_in
        s = 2 * P
        g = 'P'                                         # ROIAlign averages pyramid values: the pooled map keeps the pyramid's scale
        for i, cw in enumerate(w.kpts):
            y = self.kp_a if i % 2 == 0 else self.kp_b
            self._conv(cw, x, R, s, s, y, s, s, g, 'k%d' % i, x_fmt=f, y_fmt=f, name='kpts.%d' % (2 * i), **lim(s * s))
            x, g = y, 'k%d' % i
        G = cfg.KPTS_GRID
        if f and engine.KPTS_HEAD_FUSION == 'mfma' and w.kpts_class.cout <= 24 and w.kpts_up.cout == 4 * 256:
            # ... the classifier as a second GEMM on the deconvolution's 256x256 tile (MFMA form, srcnn_conv_desc.head_wf)
            self._conv(w.kpts_up, x, R, s, s, None, s, s, g, 'kup', x_fmt=f, name='kpts.deconv+class',
                       head2=(w.kpts_class, self.kp_logits, 0), **lim(s * s))
        elif f and engine.KPTS_HEAD_FUSION and w.kpts_class.cout == 6 and w.kpts_up.cout == 4 * 256:
            # SPLIT16 engine: ConvTranspose2d + ReLU + the 6-channel classifier (resnet.py:258-262) in ONE launch -- the classifier
            # runs in the deconvolution's epilogue on the pixels a workgroup has just activated; the (R, 28, 28, 256) upsampled
            # tensor is neither written nor read back (srcnn_conv_desc.head_w)
            self._conv(w.kpts_up, x, R, s, s, self.kp_up, s, s, g, 'kup', x_fmt=f, name='kpts.deconv+class',
                       head=(w.kpts_class, self.kp_logits), **lim(s * s))
        else:
            self._conv(w.kpts_up, x, R, s, s, self.kp_up, s, s, g, 'kup', x_fmt=f, y_fmt=f, name='kpts.deconv', **lim(s * s))
            self._conv(w.kpts_class, self.kp_up, R, G, G, self.kp_logits, G, G, 'kup', None, x_fmt=f, name='kpts.class', **lim(G * G))
        kp, lp, rp = (self.kpts_prob, self.left_prob, self.right_prob) if outs is None else outs
        if 'kpts_tail' not in engine.DEBUG_SKIP:
            _lib.check(_lib.lib().srcnn_kpts_tail(self.kp_logits.data_ptr(), R, G, kp.data_ptr(), lp.data_ptr(), rp.data_ptr(),
                                                  None if limit is None else limit.data_ptr(), _lib.stream()), "srcnn_kpts_tail")

    def kpts_for_kept(self, rois_left_b, keep_idx, num, im_info_b, det_kpts, precision):
        """The keypoint head for the detections of ONE image that survived class NMS (postprocess.class_nms_device: keep_idx
        (n) int32, -1 padded; num (1) int32 -- both stay on the device): gathers their rois in keep order, runs the tower with
        the device-side row limit `num` and writes their decoded keypoints (demo.py:196-209) into their own rows of
        `det_kpts` (n, 5).  Every roi's keypoint computation is independent of the other rois, so the kept detections get the
        values the full head gives them up to the engine's own plan-to-plan rounding (the row-limited launches are tuned to
        other tile / split-K plans, which add the K products in another order: probabilities within ~1e-5 relative, like any
        two tunings of one layer); rows of the 300 that the reference's scripts never read are skipped."""
        L = _lib.lib()
        n = int(keep_idx.shape[0])
        assert n == self.post and rois_left_b.is_contiguous()
        # the decode kernel dereferences im_info: a host tensor (which forward(), decode_detections and set_inputs all accept)
        # must be brought to the device here, not handed over as a host pointer
        im_info_b = im_info_b.reshape(-1, 3)[0:1].to(device=self.dev, dtype=torch.float32).contiguous()
        for name, tns in (('rois_left_b', rois_left_b), ('keep_idx', keep_idx), ('num', num), ('det_kpts', det_kpts)):
            assert tns.is_cuda and tns.is_contiguous(), "kpts_for_kept: %s must be a contiguous device tensor" % name
        prev, engine.PRECISION = engine.PRECISION, precision
        self.fmt = _lib.FMT_SPLIT16 if precision == 'f16x3' else _lib.FMT_F32
        # the tower's range-guard reports belong to this plan's forward (the record packed after this call carries the word)
        _lib.check(L.srcnn_range_flag_bind(self.range_flag.data_ptr()), "srcnn_range_flag_bind")
        try:
            _lib.check(L.srcnn_gather_rows(rois_left_b.data_ptr(), keep_idx.data_ptr(), n, 5, self.kp_rois.data_ptr(), _lib.stream()),
                       "srcnn_gather_rows")
            self.kpts_head(rois=self.kp_rois, n_rois=n, limit=num if self.fmt else None, outs=(self.kp_prob, self.kp_left, self.kp_right))
            G = cfg.KPTS_GRID
            _lib.check(L.srcnn_decode_kept_kpts(rois_left_b.data_ptr(), self.kp_prob.data_ptr(), self.kp_left.data_ptr(),
                                                self.kp_right.data_ptr(), keep_idx.data_ptr(), num.data_ptr(),
                                                im_info_b.data_ptr(), n, G, det_kpts.data_ptr(), _lib.stream()),
                       "srcnn_decode_kept_kpts")
        finally:
            engine.PRECISION = prev

    def __del__(self):
        # run() / kpts_for_kept() leave this plan's range word bound on the thread (the decode / pack calls that follow a forward
        # report to it); once the plan is evicted or freed that binding would dangle -- hand the thread back to the library's own word
        try:
            L = _lib.lib()
            if L.srcnn_range_flag_device_word() == self.range_flag.data_ptr():
                L.srcnn_range_flag_bind(None)
        except Exception:          # interpreter shutdown: nothing left to protect
            pass

    def heads(self, kpts=True):
        """Box head (M=300 GEMMs, poor chip fill on their own) runs beside the keypoint tower (kpts=False: box head only --
        the keypoints then come from kpts_for_kept after class NMS)."""
        if not kpts:
            self.box_head()
        elif self._par():
            side = self.side[0]
            self._fork(side)
            with torch.cuda.stream(side):
                self.box_head()
            self.kpts_head()
            self._join(side)
        else:
            self.box_head()
            self.kpts_head()

    def launch_all(self, kpts=True):
        self.trunk()
        self.fpn_rpn()
        self.proposals()
        self.heads(kpts)

    # ------------------------------------------------------------------ driver
    def set_inputs(self, im_left, im_right, im_info, copy=False):
        """The network inputs of the next run.  Float32 contiguous device tensors are NOT copied: the stem's pack kernel reads them
        where they are (the plan keeps a reference until the next call; two 14 MB copy launches per forward saved) -- unless
        `copy` (a captured hipGraph reads fixed addresses) or the tensors need a conversion anyway."""
        ok = lambda t: (t.is_cuda and t.device == self.dev and t.dtype == torch.float32 and t.is_contiguous()
                        and tuple(t.shape) == tuple(self.im_left.shape))
        if copy or not (ok(im_left) and ok(im_right)):
            self.im_left.copy_(im_left, non_blocking=True)
            self.im_right.copy_(im_right, non_blocking=True)
            self._src = (self.im_left, self.im_right)
        else:
            self._src = (im_left, im_right)
        self.packed_fmt = -1
        self.im_info.copy_(im_info.view(self.B, 3), non_blocking=True)

    def pack_inputs(self, fmt):
        """Issue the stem-input pack of the current sources NOW (outside a recorded launch program: the program's own launch list
        then starts at the stem conv and never holds a pointer to a caller's tensor)."""
        engine.stem_pack_pair(self._src[0], self._src[1], self.packed, out_fmt=fmt)
        self.packed_fmt = fmt

    def set_images(self, img_left_u8, img_right_u8, precision='f32', target_short=600):
        """Fused A0 (B = 1): uint8 RGB device images -> this plan's network-input planes (kept: dense alignment reads
        them) AND its packed stem input, one pass per eye (srcnn_preprocess); trunk() then skips the stem_pack launches.
        Returns im_scale."""
        assert self.B == 1, "set_images feeds one stereo pair"
        fmt = _lib.FMT_SPLIT16 if precision == 'f16x3' else _lib.FMT_F32
        L = _lib.lib()
        scale = None
        per_image = (self.H + 6) * (self.W + 8) * 4 * 4
        for i, (img, planar) in enumerate(((img_left_u8, self.im_left), (img_right_u8, self.im_right))):
            # device images, or PAGE-LOCKED host images the kernel reads over the bus itself (zero-copy: no hipMemcpy in the stream)
            assert (img.is_cuda or img.is_pinned()) and img.dtype == torch.uint8 and img.dim() == 3
            if img.is_cuda:
                img = img.contiguous()
            assert img.is_contiguous(), 'a page-locked host image must be contiguous (it is read where it is)'
            H0, W0 = int(img.shape[0]), int(img.shape[1])
            OH, OW, scale = engine.preprocess_size(H0, W0, target_short)
            assert (OH, OW) == (self.H, self.W), "plan was built for another input size"
            _lib.check(L.srcnn_preprocess(img.data_ptr(), H0, W0, scale, planar.data_ptr(), OH, OW,
                                          self.packed.data_ptr() + i * per_image, fmt, _lib.stream()), "srcnn_preprocess")
        self.im_info.copy_(torch.tensor([[self.H, self.W, scale]], dtype=torch.float32), non_blocking=True)
        self.packed_fmt = fmt
        self._src = (self.im_left, self.im_right)      # the planes the preprocessing just wrote
        return scale

    def _record_program(self, precision, kpts=True):
        """One pass through launch_all() with the library in record mode: nothing is launched, every kernel launch / memset /
        stream dependency lands in a native list that srcnn_program_run re-issues (include/srcnn_hip.h)."""
        L = _lib.lib()
        self.launch_all(kpts)                 # warm-up: tunes the plans, sizes every workspace, splits the weights
        torch.cuda.synchronize()
        self.packed_fmt = self.fmt            # `packed` holds this input in the run's format: the recorded list starts behind the pack
        prog = L.srcnn_program_create()
        refs = []
        _lib._recording_refs = refs
        _lib.check(L.srcnn_program_begin(prog, torch.cuda.current_stream().cuda_stream), "srcnn_program_begin")
        self._rec = prog
        try:
            self.launch_all(kpts)
        finally:
            self._rec = None
            _lib._recording_refs = None
            _lib.check(L.srcnn_program_end(prog), "srcnn_program_end")
        self.programs[self.program_key(precision, kpts)] = (prog, refs)
        return prog

    def program_key(self, precision, kpts, par=None):
        """Key of a recorded launch program in self.programs: one list per (engine, keypoint branch, branches on side streams,
        chained bottlenecks)."""
        return (precision, kpts, self._par() if par is None else bool(par), engine.chain_enabled())

    def run(self, use_graph=False, precision='f32', use_program=False, kpts=True):
        """precision: 'f32' (exact fp32 MFMA engine) or 'f16x3' (3-term split on the f16 MFMA).
        use_program: replay the forward from the native launch list (recorded on first use) instead of walking the Python
        launch code -- same launches, same streams, same results."""
        prev = engine.PRECISION
        engine.PRECISION = precision
        # every launch of this forward -- and the decode / pack calls the caller makes right after it on this thread -- report
        # range violations to this plan's word (it stays bound until the next forward binds its own)
        _lib.check(_lib.lib().srcnn_range_flag_bind(self.range_flag.data_ptr()), "srcnn_range_flag_bind")
        # the f16x3 engine keeps activations in the SPLIT16 format between convolutions so that
        # both GEMM operands are DMA'd into LDS (csrc/conv_f16s.hip); the fp32 engine uses F32
        self.fmt = _lib.FMT_SPLIT16 if precision == 'f16x3' else _lib.FMT_F32
        if self.fmt and engine.ACT_SCALES and not self.w.calibrated:
            self.calibrate()               # once per weights: the scales of every SPLIT16 tensor group, from this first input
        epoch = (self.w.calib_epoch, engine.tune_mode_key(), engine.PLAN_EPOCH)
        if self._epoch != epoch:      # the scales, or the tuner's objective (= the plan set), changed since this plan recorded its launch lists
            for prog, _ in self.programs.values():
                _lib.lib().srcnn_program_destroy(prog)
            self.programs, self.graphs, self._epoch = {}, {}, epoch
        try:
            if use_program and not use_graph:
                # the stem input is packed here, eagerly, from wherever the inputs are (set_inputs keeps references instead of
                # copying; set_images wrote `packed` itself); the recorded list starts at the stem conv
                ent = self.programs.get(self.program_key(precision, kpts))      # one list per (engine, branch, stream regime)
                if ent is None:
                    prog = self._record_program(precision, kpts)             # (its warm-up run packs and consumes the input)
                    self.packed_fmt = -1
                else:
                    prog = ent[0]
                if self.packed_fmt != self.fmt:
                    self.pack_inputs(self.fmt)
                self.packed_fmt = -1
                _lib.check(_lib.lib().srcnn_program_run(prog, torch.cuda.current_stream().cuda_stream), "srcnn_program_run")
                return
            if not use_graph or not kpts:
                self.launch_all(kpts)
                return
            self.packed_fmt = -1                  # a captured graph always contains the stem_pack launches
            if precision not in self.graphs:
                self.launch_all()                 # warm-up: sizes every workspace, splits weights, before capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.launch_all()
                self.graphs[precision] = g
            self.graphs[precision].replay()
        finally:
            engine.PRECISION = prev

    def as_f32(self, buf):
        """fp32 NHWC copy of an internal activation buffer (whatever format the last run used)."""
        if not self.fmt:
            return buf
        y = engine.act_convert(buf, self.fmt, _lib.FMT_F32)
        k = self._buf_shift.get(buf.data_ptr(), 0)
        return y * (2.0 ** -k) if k else y

    def outputs(self, kpts=True, alias=False):
        """The forward's results.  Default: tensors the caller OWNS -- after an eager run the result buffers themselves are handed
        over and replaced by fresh allocations (no copy kernels); a recorded launch program / captured graph writes to fixed
        addresses, so once one exists the results are cloned instead (8 small copy launches).
        alias=True (the streamed entry points: pipeline flows, bench.py's step): VIEWS of this plan's own buffers, valid until the
        next forward on this plan (slot) -- for callers that consume them in stream order right away (decode + class NMS write
        fresh tensors), which is what every serving flow does: no copy launches at all."""
        w, B = self.w, self.B
        names = ('rois_left', 'rois_right', 'cls_prob', 'bbox_pred', 'dim_orien', 'kpts_prob', 'left_prob', 'right_prob')
        if alias:
            t = {n: getattr(self, n) for n in names}
        elif self.graphs or self.programs:
            t = {n: (getattr(self, n).clone() if kpts or n not in ('kpts_prob', 'left_prob', 'right_prob') else None) for n in names}
        else:
            t = {n: getattr(self, n) for n in names}
            for n in names:
                if kpts or n not in ('kpts_prob', 'left_prob', 'right_prob'):
                    setattr(self, n, torch.empty_like(getattr(self, n)))
        res = {'rois_left': t['rois_left'], 'rois_right': t['rois_right'], 'cls_prob': t['cls_prob'].view(B, self.post, -1),
               'bbox_pred': t['bbox_pred'].view(B, self.post, -1), 'dim_orien_pred': t['dim_orien'].view(B, self.post, -1),
               'kpts_prob': t['kpts_prob'], 'left_border_prob': t['left_prob'], 'right_border_prob': t['right_prob']}
        if not kpts:        # lazy keypoint head: the three keypoint outputs do not exist for this forward
            res['kpts_prob'] = res['left_border_prob'] = res['right_border_prob'] = None
        return res
