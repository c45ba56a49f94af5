"""`_StereoRCNN` - reference lib/model/stereo_rcnn/stereo_rcnn.py:24-324 (inference branch).

The class keeps the reference's surface: construction through a subclass that defines
`_init_modules`, `create_architecture()`, `load_state_dict()` with the reference's key
schema, `PyramidRoI_Feat`, and `forward(9 args) -> 15-tuple`.  The nn.Module parameters are
containers only; `forward` executes a static launch plan (plan.py) in the HIP library.
"""
import ctypes

import torch
import torch.nn as nn

from ... import _lib, engine
from ..roi_align.modules.roi_align import RoIAlignAvg
from ..rpn.stereo_rpn import _Stereo_RPN
from ..utils.config import cfg
from .plan import Plan, Weights


class _StereoRCNN(nn.Module):
    """ FPN-based Stereo R-CNN, eval mode only (the training branch of the reference,
    stereo_rcnn.py:198-230,273-311, is out of scope). """

    def __init__(self, classes):
        super(_StereoRCNN, self).__init__()
        self.classes = classes
        self.n_classes = len(classes)
        self.RCNN_loss_cls = 0
        self.RCNN_loss_bbox_left_right = 0
        self.RCNN_loss_bbox = 0
        self.RCNN_loss_dis = 0
        self.RCNN_loss_dim = 0
        self.RCNN_loss_dim_orien = 0
        self.RCNN_loss_kpts = 0
        self.RCNN_rpn = _Stereo_RPN(self.dout_base_model)
        self.RCNN_roi_align = RoIAlignAvg(cfg.POOLING_SIZE, cfg.POOLING_SIZE, 1.0 / 16.0)
        self.RCNN_roi_kpts_align = RoIAlignAvg(cfg.POOLING_SIZE * 2, cfg.POOLING_SIZE * 2, 1.0 / 16.0)
        self.use_graph = False            # replay the forward as one hipGraph (see plan.py)
        self.use_program = False          # replay the forward from the native launch list (srcnn_program_run; see plan.py)
        # conv engine: 'f16x3' (default: error-compensated 3-term split on the f16 MFMA, fp32-class results, same parity
        # tolerances) or 'f32' (exact fp32 MFMA, ~2.4x slower)
        self.precision = 'f16x3'
        # forward() returns tensors the caller owns (copies, once a launch program / graph writes to fixed addresses).  True -- or
        # forward(alias_outputs=True), what the streamed entry points pass -- returns VIEWS of the slot's own result buffers instead,
        # valid until the next forward on that slot: for callers that consume them right away in stream order (plan.Plan.outputs)
        self.alias_outputs = False
        self._weights = None
        self._plans = {}

    # ------------------------------------------------------------------ construction
    def _init_weights(self):
        """stereo_rcnn.py:47-85: N(0, std) weights / zero bias for the new layers."""
        def normal_(m, std):
            m.weight.data.normal_(0.0, std)
            m.bias.data.zero_()
        for m in (self.RCNN_toplayer, self.RCNN_smooth1, self.RCNN_smooth2, self.RCNN_smooth3,
                  self.RCNN_latlayer1, self.RCNN_latlayer2, self.RCNN_latlayer3,
                  self.RCNN_rpn.RPN_Conv, self.RCNN_rpn.RPN_cls_score, self.RCNN_rpn.RPN_bbox_pred_left_right,
                  self.RCNN_cls_score):
            normal_(m, 0.01)
        normal_(self.RCNN_bbox_pred, 0.001)
        normal_(self.RCNN_dim_orien_pred, 0.001)
        normal_(self.kpts_class, 0.1)
        for seq in (self.RCNN_top, self.RCNN_kpts):
            for m in seq.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    normal_(m, 0.02)

    def create_architecture(self):
        self._init_modules()
        self._init_weights()
        self.invalidate()

    MAX_PLANS = 16          # LRU bound of the per-(B, H, W, slot) plan cache (see _get_plan)

    def invalidate(self):
        """Drop the engine-side copies of the weights (call after editing parameters in place)."""
        self._weights = None
        self._plans = {}

    def load_state_dict(self, state_dict, strict=True):
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith('num_batches_tracked')}
        own = self.state_dict()
        missing = [k for k in own if k not in state_dict and not k.endswith('num_batches_tracked')]
        extra = [k for k in state_dict if k not in own]
        if strict and (missing or extra):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing[:5], extra[:5]))
        with torch.no_grad():
            for k, v in state_dict.items():
                if k in own:
                    own[k].copy_(v)
        self.invalidate()

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("this build is the inference path; training is out of scope")
        return nn.Module.train(self, False)

    # ------------------------------------------------------------------ execution
    def _device(self):
        return self.RCNN_toplayer.weight.device

    def _get_plan(self, B, H, W, slot=0):
        dev = self._device()
        if dev.type != 'cuda':
            raise RuntimeError("_StereoRCNN.forward needs the model on a GPU (call .cuda()); there is no CPU path")
        if self._weights is None or self._weights.device != dev:
            self._weights = Weights(self.state_dict(), dev)
            self._plans = {}
        key = (B, H, W, slot)
        plan = self._plans.pop(key, None)
        if plan is None:
            # a plan pre-allocates every activation of its (B, H, W) -- ~2 GB at B = 1, 600 x 1987 -- plus side streams.  KITTI has
            # four frame sizes (x `slot`s in flight); the cache keeps the MAX_PLANS most recently used and drops the oldest, so
            # arbitrary-size inputs cannot grow device memory without bound.
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            plan = Plan(self._weights, B, H, W)
        self._plans[key] = plan                      # (re)insert as the most recently used
        return plan

    def PyramidRoI_Feat(self, feat_maps, rois, im_info, kpts=False, single_level=None):
        """stereo_rcnn.py:110-139 with the reference's NCHW in / NCHW out contract.
        feat_maps: 4 NCHW maps (P2..P5); rois (n,5).  One fused native call (level routing,
        lattice sampling, 2x2 average) instead of the per-level Python loop."""
        maps = [engine.nchw_to_nhwc(m.contiguous().float()) for m in feat_maps]
        C = int(feat_maps[0].shape[1])
        A = cfg.POOLING_SIZE * 2 if kpts else cfg.POOLING_SIZE
        n = int(rois.shape[0])
        rois = rois.contiguous().float()
        out = torch.empty((n, A, A, C), device=rois.device)
        ptrs = (ctypes.c_void_p * 4)(*[m.data_ptr() for m in maps])
        mh = (ctypes.c_int * 4)(*[int(m.shape[1]) for m in maps])
        mw = (ctypes.c_int * 4)(*[int(m.shape[2]) for m in maps])
        _lib.check(_lib.lib().srcnn_pyramid_roi_align(ptrs, mh, mw, C, float(im_info[0][0]), rois.data_ptr(), n, A,
                                                      out.data_ptr(), C, 0, _lib.FMT_F32, _lib.FMT_F32, None,
                                                      _lib.stream()), "srcnn_pyramid_roi_align")
        return engine.nhwc_to_nchw(out)

    def forward(self, im_left_data, im_right_data, im_info, gt_boxes_left=None, gt_boxes_right=None,
                gt_boxes_merge=None, gt_dim_orien=None, gt_kpts=None, num_boxes=None, slot=0, kpts=True, alias_outputs=None):
        """Reference signature and 15-tuple return (stereo_rcnn.py:141-142,322-324).
        The gt_* / num_boxes arguments are accepted and ignored exactly as in eval mode.
        `slot` (extension): independent buffer set, so that several pairs can be in flight on different HIP
        streams (each stream uses its own slot).
        `kpts=False` (extension, stereo_rcnn_amd.pipeline): leave the keypoint branch out of the forward -- the three keypoint
        outputs are None -- because the caller will run it on the detections that survive class NMS only
        (plan.Plan.kpts_for_kept; the reference's scripts read no other row: demo.py:196-257).
        INPUT OWNERSHIP: the forward reads im_left_data / im_right_data / im_info WHERE THEY ARE (no copy; a captured
        hipGraph is the exception) -- asynchronously, on the calling stream, and possibly AGAIN later: a pair whose SPLIT16 range
        guard trips is re-run on the fp32 engine from the same tensors (pipeline), and a re-calibration reads them once more.
        The caller must leave them unmodified until it has consumed this forward's results (for the streamed entry points: until
        the slot's pair has been collected).  Callers that refill one input buffer per frame in place should pass a clone."""
        if self.training:
            raise NotImplementedError("training forward is out of scope; call .eval()")
        B, _, H, W = im_left_data.shape
        plan = self._get_plan(int(B), int(H), int(W), slot)
        plan.set_inputs(im_left_data, im_right_data, im_info, copy=bool(self.use_graph))
        return self._run(plan, kpts, alias_outputs)

    def forward_images(self, img_left_u8, img_right_u8, target_short=None, slot=0, kpts=True, alias_outputs=None):
        """Extension: the reference's preprocessing (demo.py:103-129) fused in front of the forward.  uint8 RGB
        (H, W, 3) DEVICE images -> (the forward's 15-tuple, im_left_data, im_right_data, im_info); the network-input planes
        and the stem's packed input are produced in one pass per eye, the float32 planes are returned because the dense
        alignment takes them (demo.py:306-308) -- they alias the plan's buffers until the next call on this slot."""
        if self.training:
            raise NotImplementedError("training forward is out of scope; call .eval()")
        short = cfg.TEST.SCALES[0] if target_short is None else target_short
        H0, W0 = int(img_left_u8.shape[0]), int(img_left_u8.shape[1])
        OH, OW, _ = engine.preprocess_size(H0, W0, short)
        plan = self._get_plan(1, OH, OW, slot)
        plan.set_images(img_left_u8, img_right_u8, self.precision, short)
        return self._run(plan, kpts, alias_outputs), plan.im_left, plan.im_right, plan.im_info

    def calibrate_activation_scales(self, frames, slot=0):
        """Extension: choose the SPLIT16 engine's per-tensor power-of-two activation scales (plan.Plan.calibrate) from SEVERAL
        representative inputs instead of from the first forward alone.  frames: iterable of (im_left_data, im_right_data,
        im_info) as `forward` takes them; one forward each on the exact fp32 engine; every tensor group's scale covers the
        largest value any of them produced (x32 headroom on top; the range guard watches the rest).  Returns the shifts."""
        first = True
        for im_left_data, im_right_data, im_info in frames:
            B, _, H, W = im_left_data.shape
            plan = self._get_plan(int(B), int(H), int(W), slot)
            plan.set_inputs(im_left_data, im_right_data, im_info)
            if first and self._weights is not None:
                self._weights.calibration_max = {}
            plan.calibrate(merge=not first)
            first = False
        return dict(self._weights.shifts) if self._weights is not None else {}

    def check_range(self, reset=True):
        """SPLIT16 range guard of the f16x3 engine (engine.range_flag): raises engine.Split16RangeError naming the layer if an
        activation of a forward since the last reset left the f16 range (the results are then invalid: re-run with precision
        'f32').  Synchronises the device.  The 3-D pipeline checks the same flag through the detection record instead."""
        flag, name = engine.range_flag(reset)
        if flag:
            raise engine.Split16RangeError('SPLIT16 range exceeded in %s' % name)

    def _run(self, plan, kpts=True, alias=None):
        plan.run(self.use_graph, self.precision, getattr(self, 'use_program', False), kpts=kpts)
        o = plan.outputs(kpts, alias=self.alias_outputs if alias is None else bool(alias))
        self.RCNN_loss_cls = 0
        self.RCNN_loss_bbox = 0
        rpn_loss_cls, rpn_loss_bbox_left_right = 0, 0
        rois_label = None
        return o['rois_left'], o['rois_right'], o['cls_prob'], o['bbox_pred'], o['dim_orien_pred'], \
            o['kpts_prob'], o['left_border_prob'], o['right_border_prob'], rpn_loss_cls, rpn_loss_bbox_left_right, \
            self.RCNN_loss_cls, self.RCNN_loss_bbox, self.RCNN_loss_dim_orien, self.RCNN_loss_kpts, rois_label
