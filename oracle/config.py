"""Inference constants of the reference (TEST INFRASTRUCTURE: oracle side).

Values restated from /root/reference/lib/model/utils/config.py (line numbers in
comments) and the demo/test scripts.  The product package keeps its own copy in
stereo_rcnn_amd/model/utils/config.py; tests assert the two agree.
"""
import numpy as np

RPN_PRE_NMS_TOP_N = 6000        # config.py:129  TEST.RPN_PRE_NMS_TOP_N
RPN_POST_NMS_TOP_N = 300        # config.py:132
RPN_NMS_THRESH = 0.7            # config.py:127
TEST_NMS = 0.3                  # config.py:124
SCALES = (600,)                 # config.py:49 (demo.py:115 uses TRAIN.SCALES)
MAX_SIZE = 2484                 # config.py:52,120
PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])  # config.py:170 (BGR)
KPTS_GRID = 28                  # config.py:173
RNG_SEED = 3                    # config.py:178
POOLING_SIZE = 7                # config.py:204
ANCHOR_RATIOS = [0.5, 1, 2]     # config.py:210
FPN_ANCHOR_SCALES = [32, 64, 128, 256, 512]   # config.py:216
FPN_FEAT_STRIDES = [4, 8, 16, 32, 64]         # config.py:219
FPN_ANCHOR_STRIDE = 1           # config.py:222
BBOX_NORMALIZE_MEANS = (0.0, 0.0, 0.0, 0.0)   # config.py:77
BBOX_NORMALIZE_STDS = (0.1, 0.1, 0.2, 0.2)    # config.py:78
DIM_NORMALIZE_MEANS = (1.6, 1.5, 4.0, 0.0, 0.0)  # config.py:81
DIM_NORMALIZE_STDS = (0.5, 0.5, 0.5, 0.5, 0.5)   # config.py:82
EVAL_THRESH = 0.05              # demo.py:94
VIS_THRESH = 0.7                # demo.py:95
CLASSES = ('__background__', 'Car')  # demo.py:74
