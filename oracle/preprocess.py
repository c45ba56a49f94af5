"""A0 preprocessing oracle -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the reference's image preparation (demo.py:103-129 == lib/model/utils/blob.py:39-64):

    img = imread(path)[:, :, ::-1].astype(np.float32)       # RGB -> BGR
    img -= cfg.PIXEL_MEANS                                   # float32 array -= float64 (1,1,3) array
    img = cv2.resize(img, None, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR)

The arithmetic of the last line lives in a third-party dependency that is absent from
/root/reference and from this image (`opencv-python`, unpinned in requirements.txt:3), so it is
restated here from OpenCV's published algorithm -- modules/imgproc/src/resize.cpp, identical in
every 3.x / 4.x release for this code path:

  * cv::resize with an empty dsize:  dsize = (saturate_cast<int>(cols*fx), saturate_cast<int>(rows*fy)),
    saturate_cast<int>(double) = cvRound = round-half-to-even; the sampling step is 1/fx (NOT cols/dsize.width);
  * resizeGeneric_ set-up for INTER_LINEAR, float source:
        fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = cvFloor(fx);  fx -= sx;          (double math, then float)
        sx < 0          ->  sx = 0, fx = 0            sx >= cols-1  ->  sx = cols-1, fx = 0   (xmax: plain copy)
        alpha = {1.f - fx, fx};      rows: the same with fy / sy, but fy is NOT zeroed at the border --
        the two source rows are clipped to [0, rows-1] instead (resizeGeneric_Invoker: clip(sy + k, 0, rows));
  * HResizeLinear<float,float,float>:  D[dx] = S[sx]*a0 + S[sx + cn]*a1      for dx <  xmax
                                       D[dx] = S[sx]*1.f                     for dx >= xmax
  * VResizeLinear<float,float,float>:  dst[x] = S0[x]*b0 + S1[x]*b1
    every product and sum rounded to float32 separately: the opencv-python x86-64 wheels compile this file for the SSE3
    baseline (no FMA; v_muladd = mul + add there), which is what "the reference CPU path" runs.

Parity status of THIS function: pinned to the published algorithm above and to hand-computed known answers
(tests/test_preprocess_cpu.py), not to outputs of cv2 itself (cv2 cannot be imported here).
"""
import numpy as np

PIXEL_MEANS_BGR = (102.9801, 115.9465, 122.7717)          # lib/model/utils/config.py:170


def cv_round(v):
    """cvRound(double): nearest integer, ties to even (lrint in the default rounding mode)."""
    return int(np.rint(v))


def resize_dsize(rows, cols, fx, fy):
    """cv::resize's output size for an empty dsize (resize.cpp: saturate_cast<int>(ssize.width * inv_scale_x))."""
    return cv_round(rows * float(fy)), cv_round(cols * float(fx))


def _taps(n_dst, n_src, scale, zero_frac_at_border):
    """Per destination index: (s0, s1, w0, w1) float32 weights of the two source taps."""
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * np.float64(scale) - 0.5).astype(np.float32)         # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                                     # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)                    # fx -= sx  (float)
    if zero_frac_at_border:                                              # horizontal pass
        lo = s < 0
        s = np.where(lo, 0, s)
        f = np.where(lo, np.float32(0), f)
        hi = s >= n_src - 1                                              # dx >= xmax: D = S[cols-1] * 1.f
        s = np.where(hi, n_src - 1, s)
        f = np.where(hi, np.float32(0), f)
        s1 = np.minimum(s + 1, n_src - 1)                                # weight 0 there; never read out of bounds
        copy = hi
    else:                                                                # vertical pass: clip rows, keep fy
        s1 = np.clip(s + 1, 0, n_src - 1)
        s = np.clip(s, 0, n_src - 1)
        copy = np.zeros(n_dst, dtype=bool)
    w1 = f.astype(np.float32)
    w0 = (np.float32(1.0) - w1).astype(np.float32)
    return s, s1, w0, w1, copy


def cv2_resize_linear_f32(src, fx, fy):
    """cv2.resize(src, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for a float32 (H, W, C) array."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    rows, cols = src.shape[0], src.shape[1]
    orows, ocols = resize_dsize(rows, cols, fx, fy)
    scale_x, scale_y = 1.0 / float(fx), 1.0 / float(fy)                 # double scale_x = 1./inv_scale_x
    xs0, xs1, a0, a1, xcopy = _taps(ocols, cols, scale_x, True)
    ys0, ys1, b0, b1, _ = _taps(orows, rows, scale_y, False)
    # horizontal pass of every source row (float32 products and sum rounded one by one)
    s0 = src[:, xs0, :]
    s1 = src[:, xs1, :]
    h = (s0 * a0[None, :, None]).astype(np.float32) + (s1 * a1[None, :, None]).astype(np.float32)
    h = np.where(xcopy[None, :, None], s0, h).astype(np.float32)         # dx >= xmax: S[sx] * ONE
    # vertical pass
    out = (h[ys0] * b0[:, None, None]).astype(np.float32) + (h[ys1] * b1[:, None, None]).astype(np.float32)
    return out.astype(np.float32)


def prepare_image(img_rgb_u8, target_short=600, max_size=2484):
    """uint8 RGB (H, W, 3) -> (float32 (1, 3, OH, OW) BGR mean-subtracted resized array, im_scale)  (demo.py:107-124;
    the `max_size` crop is blob.py:59-61 and is inert for KITTI)."""
    im = np.asarray(img_rgb_u8)[:, :, ::-1].astype(np.float32)
    # `img -= cfg.PIXEL_MEANS`: float32 operand, float64 operand -> computed in float64, stored as float32
    im = (im.astype(np.float64) - np.asarray(PIXEL_MEANS_BGR, np.float64).reshape(1, 1, 3)).astype(np.float32)
    im_scale = float(target_short) / float(min(im.shape[0], im.shape[1]))
    im = cv2_resize_linear_f32(im, im_scale, im_scale)
    if im.shape[1] > max_size:
        im = im[:, :max_size, :]
    return np.ascontiguousarray(im.transpose(2, 0, 1)[None]), im_scale
