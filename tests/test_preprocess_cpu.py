"""A0: the OpenCV-resize restatement (oracle/preprocess.py) against hand-computed known answers, an independent
scalar transcription of the same published algorithm, and cv::resize's output-size rule on every KITTI frame size."""
import math

import numpy as np
import pytest

from oracle import preprocess as opre


def _scalar_resize(src, fx, fy):
    """Second, loop-level transcription of resize.cpp's INTER_LINEAR float path (independent of the vectorised one)."""
    f32 = np.float32
    rows, cols, cn = src.shape
    orows, ocols = int(np.rint(rows * fy)), int(np.rint(cols * fx))
    sx_, sy_ = 1.0 / fx, 1.0 / fy
    xofs, alpha, xmax = [], [], ocols
    for dx in range(ocols):
        f = f32((dx + 0.5) * sx_ - 0.5)
        s = int(math.floor(f))
        f = f32(f - f32(s))
        if s < 0:
            s, f = 0, f32(0)
        if s + 1 >= cols:
            xmax = min(xmax, dx)
            if s >= cols - 1:
                s, f = cols - 1, f32(0)
        xofs.append(s)
        alpha.append((f32(1) - f, f))
    hbuf = np.zeros((rows, ocols, cn), f32)
    for y in range(rows):
        for dx in range(ocols):
            for k in range(cn):
                if dx < xmax:
                    hbuf[y, dx, k] = f32(src[y, xofs[dx], k] * alpha[dx][0]) + f32(src[y, xofs[dx] + 1, k] * alpha[dx][1])
                else:
                    hbuf[y, dx, k] = src[y, xofs[dx], k] * f32(1)
    out = np.zeros((orows, ocols, cn), f32)
    for dy in range(orows):
        f = f32((dy + 0.5) * sy_ - 0.5)
        s = int(math.floor(f))
        f = f32(f - f32(s))
        r0, r1 = min(max(s, 0), rows - 1), min(max(s + 1, 0), rows - 1)
        b0, b1 = f32(1) - f, f
        out[dy] = (hbuf[r0] * b0).astype(f32) + (hbuf[r1] * b1).astype(f32)
    return out


def test_known_answer_2x2_times_2():
    # the textbook case: cv2.resize([[0,10],[20,30]], fx=fy=2, INTER_LINEAR)
    src = np.array([[0, 10], [20, 30]], np.float32)[:, :, None]
    got = opre.cv2_resize_linear_f32(src, 2.0, 2.0)[:, :, 0]
    want = np.array([[0, 2.5, 7.5, 10], [5, 7.5, 12.5, 15], [15, 17.5, 22.5, 25], [20, 22.5, 27.5, 30]], np.float32)
    assert np.array_equal(got, want)


def test_identity_scale_is_a_copy():
    rng = np.random.default_rng(0)
    src = rng.normal(size=(7, 9, 3)).astype(np.float32)
    # scale 1: every tap has fraction 0 -> S*1 + S'*0
    assert np.array_equal(opre.cv2_resize_linear_f32(src, 1.0, 1.0), src)


def test_downscale_by_two_averages_pairs():
    src = np.arange(4 * 6, dtype=np.float32).reshape(4, 6, 1)
    got = opre.cv2_resize_linear_f32(src, 0.5, 0.5)[:, :, 0]
    # taps at (2d + 0.5): mean of the 2x2 block
    want = src[:, :, 0].reshape(2, 2, 3, 2).mean(axis=(1, 3))
    assert np.array_equal(got, want.astype(np.float32))


@pytest.mark.parametrize("hw,want", [((370, 1224), (600, 1985)), ((374, 1238), (600, 1986)), ((375, 1242), (600, 1987)),
                                     ((376, 1241), (600, 1980))])
def test_dsize_of_every_kitti_frame_size(hw, want):
    """cv::resize: dsize = cvRound(src * fx) with fx = 600 / short side (floor gives 1984 for 370x1224)."""
    s = 600.0 / min(hw)
    assert opre.resize_dsize(hw[0], hw[1], s, s) == want
    from stereo_rcnn_amd import engine
    assert engine.preprocess_size(hw[0], hw[1], 600)[:2] == want


def test_cv_round_ties_to_even():
    assert [opre.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


@pytest.mark.parametrize("shape,scale", [((5, 7, 3), 1.6), ((9, 4, 1), 2.37), ((11, 13, 3), 0.61), ((3, 3, 2), 4.0)])
def test_vectorised_equals_scalar_transcription(shape, scale):
    rng = np.random.default_rng(sum(shape))
    src = (rng.normal(size=shape) * 50).astype(np.float32)
    a = opre.cv2_resize_linear_f32(src, scale, scale)
    b = _scalar_resize(src, scale, scale)
    assert a.shape == b.shape and np.array_equal(a, b)


def test_borders_horizontal_exact_vertical_blend():
    """Left/right borders replicate exactly (fraction zeroed); top/bottom blend the SAME row with (1-fy, fy), which is
    what resize.cpp does and may differ from the row by an ulp."""
    rng = np.random.default_rng(5)
    src = (rng.normal(size=(6, 8, 1)) * 100).astype(np.float32)
    out = opre.cv2_resize_linear_f32(src, 1.6, 1.6)
    hres_first_col = out[:, 0, 0]
    # column 0 only ever sees source column 0; bottom row only the last source row
    col0 = opre.cv2_resize_linear_f32(src[:, :1].repeat(2, 1), 1.0, 1.6)[:, 0, 0]
    assert np.array_equal(hres_first_col, col0)
    f32 = np.float32
    fy = f32(f32((0 + 0.5) / 1.6 - 0.5) - f32(-1))
    assert out[0, 0, 0] == f32(src[0, 0, 0] * (f32(1) - fy)) + f32(src[0, 0, 0] * fy)


def test_prepare_image_pipeline():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(12, 20, 3), dtype=np.uint8)
    t, s = opre.prepare_image(img, target_short=24)
    assert s == 2.0 and t.shape == (1, 3, 24, 40) and t.dtype == np.float32
    # channel 0 is B - 102.9801: check one interior sample against the formula by hand
    b = (img[:, :, 2].astype(np.float64) - 102.9801).astype(np.float32)
    f32 = np.float32
    want = f32(f32(f32(b[0, 0] * f32(.75)) + f32(b[0, 1] * f32(.25))) * f32(.75)) + \
        f32(f32(f32(b[1, 0] * f32(.75)) + f32(b[1, 1] * f32(.25))) * f32(.25))
    assert t[0, 0, 1, 1] == want
