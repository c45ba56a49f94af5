"""Inference configuration (`cfg`) with the reference's names (lib/model/utils/config.py).

Only the values the inference hot path reads are kept; line numbers refer to the
reference file.  `cfg` is attribute- and key-addressable like the reference's easydict
(`cfg.TEST.NMS`, `cfg['TEST'].RPN_NMS_THRESH`).
"""
import numpy as np


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


cfg = _Cfg()
cfg.TRAIN = _Cfg()
cfg.TEST = _Cfg()
cfg.RESNET = _Cfg()

cfg.TRAIN.TRUNCATED = False                       # :43
cfg.TRAIN.SCALES = (600,)                         # :49
cfg.TRAIN.MAX_SIZE = 2484                         # :52
cfg.TRAIN.BBOX_NORMALIZE_MEANS = (0.0, 0.0, 0.0, 0.0)      # :77
cfg.TRAIN.BBOX_NORMALIZE_STDS = (0.1, 0.1, 0.2, 0.2)       # :78
cfg.TRAIN.DIM_NORMALIZE_MEANS = (1.6, 1.5, 4.0, 0.0, 0.0)  # :81
cfg.TRAIN.DIM_NORMALIZE_STDS = (0.5, 0.5, 0.5, 0.5, 0.5)   # :82
cfg.TEST.SCALES = (600,)                          # :117
cfg.TEST.MAX_SIZE = 2484                          # :120
cfg.TEST.NMS = 0.3                                # :124
cfg.TEST.RPN_NMS_THRESH = 0.7                     # :127
cfg.TEST.RPN_PRE_NMS_TOP_N = 6000                 # :129
cfg.TEST.RPN_POST_NMS_TOP_N = 300                 # :132
cfg.TEST.RPN_MIN_SIZE = 16                        # :135 (filter is commented out, proposal_layer.py:90)
cfg.PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])   # :170
cfg.KPTS_GRID = 28                                # :173
cfg.RNG_SEED = 3                                  # :178
cfg.USE_GPU_NMS = True                            # :196
cfg.POOLING_SIZE = 7                              # :204
cfg.ANCHOR_RATIOS = [0.5, 1, 2]                   # :210
cfg.FEAT_STRIDE = [16, ]                          # :213
cfg.FPN_ANCHOR_SCALES = [32, 64, 128, 256, 512]   # :216
cfg.FPN_FEAT_STRIDES = [4, 8, 16, 32, 64]         # :219
cfg.FPN_ANCHOR_STRIDE = 1                         # :222
cfg.RESNET.FIXED_BLOCKS = 1
