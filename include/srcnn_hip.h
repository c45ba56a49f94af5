/*
 * srcnn_hip.h -- flat C ABI of libsrcnn_hip.so, the MI355X (gfx950) native
 * replacement for the Stereo R-CNN inference hot path.
 *
 * This is the drop-in boundary.  The reference crosses it with cffi
 * (torch.utils.ffi) into two CUDA extensions; every entry point below names the
 * reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless
 *     the name ends in _host; no torch types;
 *   - the caller allocates every output and every workspace (query the size
 *     with the matching *_workspace_bytes); the library never allocates memory
 *     the caller can see and never frees/syncs behind the caller's back;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream) and is hipGraph-capturable;
 *   - return value: 0 = SRCNN_OK, negative = error (srcnn_last_error() gives
 *     text).  The two legacy-named wrappers keep the reference's "1 = ok,
 *     0 = bad roi shape" convention (roi_align_cuda.c:19-22,39; nms_cuda.c:18).
 *   - activations are NHWC float32 inside the library; NCHW only at the edge.
 */
#ifndef SRCNN_HIP_H
#define SRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRCNN_OK 0
#define SRCNN_ERR_ARG (-1)
#define SRCNN_ERR_HIP (-2)
#define SRCNN_ERR_WORKSPACE (-3)

#define SRCNN_FMT_F32 0
#define SRCNN_FMT_SPLIT16 1

typedef void *srcnn_stream_t; /* hipStream_t */

#define SRCNN_API __attribute__((visibility("default")))

SRCNN_API int srcnn_version(void);   /* 230 = round 5, second half: srcnn_conv_desc.up_top / up_format / up_H / up_W appended, srcnn_stem_pack_pair, srcnn_pool2x2_s1; 220 = round 5: srcnn_conv_desc.head_wf / head_rows / head_parts / head_plane appended, srcnn_rpn_score_levels / _parts, srcnn_box_head_tail, srcnn_proposal_workspace_layout; 210 = round 4: stream creation, placement probe, srcnn_conv_desc.head_* appended (older callers that zero the struct are unaffected) */
SRCNN_API const char *srcnn_last_error(void);

/* ------------------------------------------------------------------ NMS (A6)
 * Replaces  int nms_cuda(THCudaIntTensor *keep_out, THCudaTensor *boxes_host,
 *                        THCudaIntTensor *num_out, float nms_overlap_thresh)
 *           lib/model/nms/src/nms_cuda.h:4-5, nms_cuda.c:8-19, and the kernel +
 *           host greedy loop nms_cuda_kernel.cu:31-161.
 * dets (n, dim>=4) float32 [x1,y1,x2,y2,(score)], ALREADY score-sorted.
 * keep_out (n) int32, num_out (1) int32 -- both on the device, as in the
 * reference (nms_gpu.py:8-9).  The greedy reduction also runs on the device:
 * nothing is copied to the host.  Keep lists are bit-identical to the reference's own
 * kernel (built for gfx950 and run on the MI355X by tests/test_ref_kernels_gpu.py).
 * LIMIT: n <= 16384 boxes per problem (the reference's RPN uses 6000, cfg.TEST.RPN_PRE_NMS_TOP_N); a larger n
 * returns SRCNN_ERR_ARG (tests/test_ops_gpu.py covers 16384 and 16385).
 */
SRCNN_API size_t srcnn_nms_workspace_bytes(int n);
SRCNN_API int srcnn_nms(int *keep_out, const float *dets, int *num_out, int n, int dim, float thresh,
              void *workspace, size_t workspace_bytes, srcnn_stream_t stream);
/* nb independent problems of n boxes each (RPN: {left,right} x batch). n_valid (nb) may be NULL. */
SRCNN_API size_t srcnn_nms_batched_workspace_bytes(int nb, int n);
SRCNN_API int srcnn_nms_batched(int *keep_out, const float *dets, int *num_out, const int *n_valid,
                      int nb, int n, int dim, float thresh,
                      void *workspace, size_t workspace_bytes, srcnn_stream_t stream);
/* legacy name/return convention: 1 = ok. Scratch comes from a library-owned pool sized at first use. */
SRCNN_API int nms_cuda(int *keep_out, const float *boxes, int *num_out, int boxes_num, int boxes_dim,
             float nms_overlap_thresh, srcnn_stream_t stream);

/* ------------------------------------------------------------- ROIAlign (A8)
 * Replaces  int roi_align_forward_cuda(int aligned_height, int aligned_width,
 *                 float spatial_scale, THCudaTensor *features,
 *                 THCudaTensor *rois, THCudaTensor *output)
 *           lib/model/roi_align/src/roi_align_cuda.h:1-2, roi_align_cuda.c:7-40,
 *           kernel roi_align_kernel.cu:15-91.
 * features (B,C,H,W) NCHW, rois (n, roi_cols) [batch,x1,y1,x2,y2], output (n,C,ah,aw).
 * Returns 1 on success, 0 when roi_cols != 5 (output untouched) -- reference convention.
 */
SRCNN_API int roi_align_forward_cuda(int aligned_height, int aligned_width, float spatial_scale,
                           const float *features, int batch, int channels, int height, int width,
                           const float *rois, int num_rois, int roi_cols, float *output,
                           srcnn_stream_t stream);
/* The reduction behind RoIAlignAvg / RoIAlignMax (modules/roi_align.py:26-29, 41-44: avg_pool2d / max_pool2d(kernel 2,
 * stride 1) of the (A+1) x (A+1) lattice roi_align_forward_cuda returns): x (planes, h, w) -> y (planes, h-1, w-1).
 * take_max 0: ((a + b) + c) + d over rows then columns, x 0.25 (ATen's order); 1: the maximum (NaN propagates). */
SRCNN_API int srcnn_pool2x2_s1(const float *x, long long planes, int h, int w, float *y, int take_max, srcnn_stream_t stream);
/* Fused PyramidRoI_Feat (stereo_rcnn.py:110-139) = level routing (natural log, round half away)
 * + RoIAlignAvg (modules/roi_align.py:26-29: (A+1)^2 lattice, then 2x2/s1 avg-pool), NHWC maps.
 * maps[l] is the level-(l+2) map (B, mh[l], mw[l], C) NHWC.  out is (n, A, A, out_cstride) NHWC and
 * channels [out_coffset, out_coffset+C) are written (lets left|right be concatenated in place).
 * roi_limit: NULL, or a device int -- only rois [0, *roi_limit) are pooled (the keypoint head on the kept detections). */
SRCNN_API int srcnn_pyramid_roi_align(const float *const *maps_host, const int *mh_host, const int *mw_host,
                            int channels, float im_height, const float *rois, int num_rois, int A,
                            float *out, int out_cstride, int out_coffset, int maps_format, int out_format,
                            const int *roi_limit, srcnn_stream_t stream);
/* format conversion of an NHWC activation tensor (pixels x C): F32 <-> SPLIT16 (API edge / tests) */
SRCNN_API int srcnn_act_convert(const void *x, int x_format, void *y, int y_format, long long pixels, int C,
                      srcnn_stream_t stream);

/* ------------------------------------------------- convolution engine (A1-A3, A9, A10)
 * Replaces the cuDNN calls behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear in
 * stereo_rcnn/resnet.py:66-146,243-286, rpn/stereo_rpn.py:32-40.  Implicit GEMM on the
 * fp32 MFMA (v_mfma_f32_32x32x2_f32) or, with desc.precision = 1, on the f16 MFMA with an
 * error-compensated 3-term split: y = act(conv(x, w) + bias + residual).
 * x: (B,H,W,*) NHWC with pixel stride x_cstride floats, Cin must be a multiple of 32.
 * w: (Cout, KH, KW, Cin) float32 (K-contiguous rows).  Frozen BN is folded by the caller.
 */
typedef struct srcnn_conv_desc {
    const float *x;
    const float *w;
    const float *bias;     /* (Cout) or NULL */
    const float *residual; /* NHWC with pixel stride res_cstride, or NULL */
    float *y;              /* NHWC, pixel stride y_cstride, channels [y_coffset, y_coffset+Cout) */
    int B, H, W, Cin, x_cstride;
    int OH, OW, Cout;
    int KH, KW, stride, pad;
    int y_cstride, y_coffset, res_cstride;
    int relu;
    int mode;              /* 0 = conv; 1 = ConvTranspose2d(k=2,s=2): Cout = 4*Cq ordered (i,j,co); 2 = conv over a batch of
                            * B = 2*B' images whose second half is written beside the first: image b >= B' goes to pixel
                            * (b - B', oh, ow), channels y_coffset + Cout + co -- the stereo RPN's [left | right] concatenation
                            * (stereo_rpn.py:77-78) in one launch (SPLIT16 f16x3 engine only, no residual) */
    /* precision 0: fp32 MFMA, `w` is float32.
     * precision 1: error-compensated 3xf16 MFMA (fp32-class result): `w` = hi halves and `w_lo` = lo
     *   halves of (weight * 2^k), both (Cout, KH, KW, Cin) _Float16; w_inv_scale = 2^-k. */
    int precision;
    const void *w_lo;
    float w_inv_scale;
    /* launch plan override (0 = built-in heuristic): workgroup tile = (64*tile_mr) x (64*tile_nr), tile_mr/tile_nr
     * in {1,2}; splits = number of K slices (deterministic workspace reduction).  The SPLIT16 f16x3 engine also
     * takes tile_waves (4 or 8 wavefronts per workgroup; 0 = 4) and tile_stages (LDS ring depth 2..4 = K tiles of
     * DMA in flight + 1; 0 = 2), and with 8 waves tile_mr = 4 (256x128, 3 stages) and tile_mr = tile_nr = 4 (256x256,
     * 2 stages; layers whose channel counts / offsets are multiples of 8).  An override the engine does not
     * implement falls back to the heuristic.  Lets the host autotune each layer shape on the device it runs on. */
    int tile_mr, tile_nr, splits;
    /* activation formats (SRCNN_FMT_*).  SPLIT16: per pixel, each group of 8 channels is stored as
     * [8 x f16 hi][8 x f16 lo] (hi = f16(v), lo = f16(v - hi); same bytes as float32, channel strides
     * are still given in float32-equivalents).  With precision 1 and x_format SPLIT16 both GEMM
     * operands are DMA'd straight into LDS (buffer_load ... lds, 32-bit offsets relative to each tile's first input row).  fp32 engine: all formats must be F32. */
    int x_format, y_format, res_format;
    int tile_waves, tile_stages;
    int layer_tag;         /* caller's id of this layer (> 0) for srcnn_range_flag_read; 0 = untagged */
    /* Device-side row limit (SPLIT16 f16x3 engine; NULL = none): only output rows m < (*m_limit) * m_limit_mul are needed.
     * Workgroups whose whole tile lies beyond exit at once; rows beyond the limit inside a computed tile are still written.
     * Lets a fixed-shape launch list serve a data-dependent row count without a host read-back: the keypoint head runs on
     * the detections that survived class NMS only (*m_limit = the device-side keep count, m_limit_mul = rows per roi). */
    const int *m_limit;
    int m_limit_mul;
    /* Second input (SPLIT16 f16x3 engine, KH = KW = 1, pad 0, mode 0; NULL = none): y = act(W[:, :Cin] . x + W[:, Cin:] . x2' + bias)
     * with x2' = the (B, H2, W2, x2_cstride) SPLIT16 tensor x2 sampled at (oh * stride2, ow * stride2), Cin2 channels -- the
     * ResNet projection shortcut (resnet.py:86-100: out = bn3(conv3(.)) + downsample(x)) computed inside the block's last 1x1
     * conv as one GEMM over K = Cin + Cin2 instead of a launch of its own whose result is written and read back as the residual.
     * w / w_lo are (Cout, Cin + Cin2); both inputs must carry the same activation scale; no split-K. */
    const void *x2;
    int Cin2, H2, W2, x2_cstride, stride2;
    /* Fused narrow 1x1 head (SPLIT16 f16x3 engine; NULL = none): instead of storing y, the epilogue applies a second, 1x1
     * convolution with head_cout (= 6) output channels to the activated output pixel and stores only that:
     *   head_y[pixel, k] = head_scale * sum_c act(conv(x))[pixel, c] * head_w[k, c] + head_bias[k]      (float32, pixel stride 6)
     * head_w is float32 (head_cout, Cq) with Cq the channels of an output pixel (Cout, or Cout / 4 in mode 1), which must be 256
     * = the N extent of the 256x256 tile this form always runs on (one workgroup then owns every channel of its pixels);
     * head_scale undoes the activation scale 2^k the caller folded into w_inv_scale / bias.  The products are float32 FMAs in
     * a fixed order (8 channels per lane, then a DPP scan over the 32 lanes of a pixel): deterministic.  The keypoint branch
     * uses it for ConvTranspose2d + ReLU + the 6-channel classifier (resnet.py:258-262 of the reference: RCNN_kpts[12:14],
     * kpts_class): the (300, 28, 28, 256) upsampled tensor is neither written nor read back, one launch goes.  y may be NULL. */
    const void *head_w;
    const void *head_bias;
    void *head_y;
    int head_cout;
    float head_scale;
    /* MFMA form of the fused narrow head (NULL = none; not together with head_w): the same second 1x1 convolution, computed as a
     * small GEMM on the matrix pipe from the tile the epilogue holds in LDS -- 3-term f16 split like the engine itself, so the
     * result equals what a separate launch of the head on the stored SPLIT16 activations gives (up to the summation order).
     * head_wf: the head's weights (head_cout, C_h) split into hi / lo f16 (x 2^k like any weight) and zero-padded to head_rows
     *   (a multiple of 8, <= 24) rows, in FRAGMENT ORDER: [C_h / 16 steps][hi, lo][k group 0, 1][head_rows][8 halves], element
     *   (step s, g, n, i) = W[n][16 s + 8 g + i]   (stereo_rcnn_amd/engine.py: head_fragments builds it).
     * head_parts = 0, FINAL form (Cq = 256: the 256x256 tile owns every channel of its pixels; modes 0 and 1):
     *   head_y[pixel, k] = head_scale * sum_c act(conv(x))[pixel, c] W[k, c] + head_bias[k]     (head_cout floats per pixel)
     * head_parts > 0, PARTIAL form (modes 0 and 2; Cout a multiple of 256): C_h = Cout (mode 0) or 2 Cout (mode 2: the head reads
     *   [left Cout | right Cout], rows of the second half of the batch are the right eye) and every (eye, N tile) of the launch
     *   stores its share   head_y[(eye * ntiles + nt) * head_plane + pixel * head_cout + k] = head_scale * partial sum;
     *   the caller adds the planes in a fixed order and the bias (srcnn_rpn_score_parts).  ntiles = Cout / 256 (256x256 tile) or
     *   Cout / 128 (tile_mr = tile_nr = 2: the 128x128 8-wave tile): size head_y for head_parts >= eyes * Cout / 128 planes.
     * In both forms y is not written; the SPLIT16 range guard watches the activations that enter the head. */
    const void *head_wf;
    int head_rows;
    int head_parts;
    long long head_plane;
    /* Fused _upsample_add (SPLIT16 f16x3 engine, mode 0, no residual / head / split-K, channel counts multiples of 8; NULL = none):
     *   y = bilinear_align_corners(up_top -> (OH, OW)) + act(conv(x) + bias)
     * with up_top the coarser pyramid level, (B, up_H, up_W, Cout) NHWC in up_format -- the FPN lateral 1x1 conv and the
     * top-down addition behind it (stereo_rcnn.py:91-108, 161-167) in ONE launch: the float32 lateral map is neither written nor
     * read back.  The interpolation is srcnn_upsample_add's arithmetic operation by operation, and the conv's value is rounded to
     * float32 exactly as the two-launch form stores it, so the result is bit-identical to srcnn_conv2d (y float32) followed by
     * srcnn_upsample_add.  Not available on the 256x256 tile (an override asking for it falls back to the heuristic plan). */
    const void *up_top;
    int up_format, up_H, up_W;
} srcnn_conv_desc;
SRCNN_API size_t srcnn_conv2d_workspace_bytes(const srcnn_conv_desc *d);
SRCNN_API int srcnn_conv2d(const srcnn_conv_desc *d, void *workspace, size_t workspace_bytes, srcnn_stream_t stream);

/* CHAIN: up to three convolutions over the SAME output rows in ONE launch (SPLIT16 f16x3 engine, SPLIT16 in and out, mode 0,
 * unsplit).  descs[0] is any convolution; descs[i > 0] must be a 1x1 / stride 1 / pad 0 convolution whose input is exactly the
 * tensor descs[i-1] writes (same pointer, channel stride = y_cstride, y_coffset 0).  A workgroup owns one M tile and computes
 * every N tile of descs[0], then of descs[1], then of descs[2]: what a phase reads is what the SAME workgroup has just
 * written (read back from the L2), so nothing but a workgroup barrier orders the phases and the results are bit-identical to
 * the same convolutions launched one after the other with the same tiles.  This is the ResNet bottleneck
 * (/root/reference/lib/model/stereo_rcnn/resnet.py:82-102) SHIFTED BY ONE convolution -- [conv2 3x3 -> conv3 (+ residual or
 * projection shortcut) -> conv1 of the next block] -- so that the only convolution with a spatial footprint comes first and
 * reads a tensor the previous launch completed.  No phase may write the tensor descs[0] reads (other workgroups still need its
 * halo rows): the caller double-buffers it.  Residuals / second inputs are read at the workgroup's own rows and must be
 * tensors no phase writes.
 * Tile: descs[i].tile_mr / tile_waves / tile_stages must agree; tile_nr may take two values (the narrow and the wide
 * convolutions of a bottleneck); srcnn_conv2d_chain_supported says whether (tile_mr, waves, stages, narrow nr, wide nr) is
 * instantiated: (2,4,2,1,2) (2,4,2,1,1) (2,4,2,2,2) (2,8,2,2,2) (2,8,4,2,2) (4,8,3,2,2) (4,8,2,4,4).  Returns SRCNN_ERR_ARG otherwise
 * -- an explicit request is never silently replaced. */
SRCNN_API int srcnn_conv2d_chain_supported(int tile_mr, int tile_waves, int tile_stages, int nr_narrow, int nr_wide);
SRCNN_API int srcnn_conv2d_chain(const srcnn_conv_desc *descs, int n, srcnn_stream_t stream);
/* GROUP: up to five independent convolutions with ONE tile configuration in one launch (SPLIT16 f16x3 engine, unsplit; all with
 * the same output format and, if any, the same MFMA-form head shape): the grid runs over the tiles of all of them.  The stereo
 * RPN applies the shared RPN_Conv + heads to five pyramid levels (/root/reference/lib/model/rpn/stereo_rpn.py:73-95); the
 * coarse levels are launches of 6 to 76 tiles.  Tiles: (2,2,8,2) (4,2,8,3) (4,4,8,2) (2,1,4,2) as (tile_mr, tile_nr, waves, stages).
 * Each convolution's result is bit-identical to its own srcnn_conv2d launch with that tile. */
SRCNN_API int srcnn_conv2d_group(const srcnn_conv_desc *descs, int n, srcnn_stream_t stream);

/* SPLIT16 range guard.  The format stores hi = f16(v) unscaled: an activation beyond +-65504 (or a NaN) becomes inf and
 * poisons what it touches, where the fp32 engine would carry on.  Every kernel that writes SPLIT16 from fresh arithmetic
 * records it: a library-owned device word keeps max(layer_tag + 1) over the launches that produced such a value since
 * the last reset (0 = every SPLIT16 tensor was in range; 9001 = srcnn_upsample_add, 9002 = an input converted by
 * srcnn_stem_pack / srcnn_act_convert; a NaN is looked for BEFORE a ReLU, which would turn it into 0).  srcnn_range_flag_read copies the word
 * to the host (it SYNCHRONISES the device) and optionally clears it; srcnn_pack_detections also drops it into
 * rec[0][1], so that the 3-D flow sees it without an extra copy.  A caller that finds it set re-runs the pair with
 * desc.precision = 0 (exact fp32 engine, F32 activations): stereo_rcnn_amd/pipeline.py does. */
SRCNN_API int srcnn_range_flag_read(int reset);
SRCNN_API const void *srcnn_range_flag_device_word(void);
/* Gives the calling THREAD its own flag word (4 zero-initialised bytes of device memory owned by the caller, on the device
 * the launches go to) for every library call it makes from now on; NULL returns to the library's process-wide word.
 * With several forwards in flight on different streams each one binds its own word before it is enqueued, so that a
 * forward is judged by its own range flag only.  srcnn_pack_detections copies the bound word into the record (row 0,
 * column 1) and clears it on its stream; srcnn_range_flag_read reads (and resets) the bound word. */
SRCNN_API int srcnn_range_flag_bind(void *device_word);

/* A0 preprocessing (demo.py:103-129, blob.py:39-64): uint8 RGB (H,W,3) on the device -> BGR, PIXEL_MEANS subtracted
 * (in double, stored float32, as numpy's float32 -= float64), then cv2.resize(img, None, None, fx=scale, fy=scale,
 * INTER_LINEAR) restated operation by operation from OpenCV's float path (oracle/preprocess.py cites it): bit-equal
 * to that restatement.  OH/OW MUST be what cv::resize derives: cvRound(H*scale), cvRound(W*scale) (ties to even).
 * Outputs (either may be NULL): out_nchw = float32 (3,OH,OW) planes, the network input forward()/dense alignment take;
 * packed = the stem's zero-bordered NHWC4 input (1, OH+6, OW+8, 4) exactly as srcnn_stem_pack would write it from
 * out_nchw, in packed_format F32 or SPLIT16 -- the fused form: no float32 intermediate is re-read to feed the stem. */
SRCNN_API int srcnn_preprocess(const unsigned char *img_rgb, int H, int W, double scale, float *out_nchw, int OH, int OW,
                     float *packed, int packed_format, srcnn_stream_t stream);
/* stem input repack: NCHW (B,3,H,W) -> zero-bordered NHWC4 (B, H+6, W+8, 4) so that the 7x7/2
 * stem (resnet.py:109) becomes 7 taps of 32 contiguous floats for the conv engine (srcnn_conv2d with Cin = 32,
 * x_cstride = 4, KH = 7, KW = 1, stride 2, pad 0).  out_format SRCNN_FMT_SPLIT16: the same bytes hold, per padded row,
 * one [8 x f16 hi][8 x f16 lo] group per two pixels (groups aligned to the row start; an odd last pixel is not
 * written -- the stem never reads it), which is what the DMA form of the f16x3 engine takes as x_format. */
SRCNN_API int srcnn_stem_pack(const float *im_nchw, int B, int H, int W, float *out, int out_format,
                              srcnn_stream_t stream);
/* ... of a stereo batch in one launch: out (2B, H+6, W+8, 4) = the B left images, then the B right ones -- the batch order
 * the shared trunk runs on (stereo_rcnn.py:155-158 feeds both eyes through the same RCNN_layer0..4). */
SRCNN_API int srcnn_stem_pack_pair(const float *left_nchw, const float *right_nchw, int B, int H, int W, float *out,
                                   int out_format, srcnn_stream_t stream);
/* MaxPool2d(3, stride 2, pad 0, ceil_mode) NHWC (resnet.py:113). */
SRCNN_API int srcnn_maxpool3x3s2_ceil(const float *x, int B, int H, int W, int C, float *y, int OH, int OW,
                            int y_format, srcnn_stream_t stream);
/* _upsample_add (stereo_rcnn.py:91-108): y = bilinear_align_corners(top -> (H,W)) + lateral, NHWC. */
SRCNN_API int srcnn_upsample_add(const float *top, int TH, int TW, const float *lateral, int B, int H, int W, int C,
                       float *y, int top_format, int y_format /* lateral is always F32 */, srcnn_stream_t stream);
/* MaxPool2d(1, stride 2) (stereo_rcnn.py:39,168): y[b,i,j,:] = x[b,2i,2j,:]. */
SRCNN_API int srcnn_subsample2(const float *x, int B, int H, int W, int C, float *y, int OH, int OW, srcnn_stream_t stream);
/* layout edge helpers */
SRCNN_API int srcnn_nhwc_to_nchw(const float *x, int B, int H, int W, int C, float *y, srcnn_stream_t stream);
SRCNN_API int srcnn_nchw_to_nhwc(const float *x, int B, int C, int H, int W, float *y, srcnn_stream_t stream);

/* ---------------------------------------------------- stereo RPN scoring + proposals (A3-A5)
 * head: (B, hw, head_cstride>=24) NHWC rows of one level's fused 1x1 heads, channels [0,6) =
 * RPN_cls_score logits, [6,24) = RPN_bbox_pred_left_right.  Writes that level's slice of probs
 * (B, A, 2) and deltas (B, A, 6) (anchor index = level_offset + loc*3 + a) in the reference's
 * flattened order, reproducing the (c, c+3) softmax pairing quirk (stereo_rpn.py:81-83,89-91). */
SRCNN_API int srcnn_rpn_score(const float *head, int B, int hw, int head_cstride, float *probs, float *deltas,
                    int level_offset, int num_anchors_total, srcnn_stream_t stream);
/* The same for ALL pyramid levels in one launch: heads[l] (B, level_hw[l], head_cstride) of level l (host arrays of nlevels <= 5
 * entries); level l's anchors start at 3 x the locations of the levels before it; num_anchors_total = 3 x all locations. */
SRCNN_API int srcnn_rpn_score_levels(const float *const *heads, const int *level_hw, int nlevels, int B, int head_cstride,
                                     float *probs, float *deltas, int num_anchors_total, srcnn_stream_t stream);
/* ... and from the per-(eye, N tile) PARTIAL sums the RPN conv's fused head leaves (srcnn_conv_desc.head_wf, partial form): level l
 * has nparts[l] planes of plane_floats[l] >= B * level_hw[l] * 24 floats each, plane q at parts[l] + q * plane_floats[l], rows
 * (b * hw + loc) x 24 channels; the planes are added in index order, then bias24 (the head's 24 biases, device), then as above. */
SRCNN_API int srcnn_rpn_score_parts(const float *const *parts, const int *nparts, const long long *plane_floats, const int *level_hw,
                                    int nlevels, int B, const float *bias24, float *probs, float *deltas, int num_anchors_total,
                                    srcnn_stream_t stream);
/* Whole _ProposalLayer.forward (proposal_layer.py:42-145): anchors (generate_anchors.py:112-173),
 * decode+clip (bbox_transform.py:79-104,177-185), stable descending sort / top pre_nms,
 * NMS(left) & NMS(right), sorted intersection, first post_nms, zero pad, batch index in col 0.
 * probs (B, A, 2), deltas (B, A, 6); level_hw_host: nlevels x {H, W}; rois_* (B, post_nms, 5).
 * LIMITS: specialised to the reference's anchor configuration (config.py: FPN_ANCHOR_SCALES {32,64,128,256,512} one per
 * level, FPN_FEAT_STRIDES {4,8,16,32,64}, ANCHOR_RATIOS {0.5,1,2}, i.e. nlevels == 5 and 3 anchors per location:
 * num_anchors must equal 3 * sum(H_l * W_l) and stay below 4M); pre_nms <= 8192 (radix top-K buffer), post_nms <= pre_nms.  Anything else
 * returns SRCNN_ERR_ARG with the reason in srcnn_last_error(). */
SRCNN_API size_t srcnn_proposal_workspace_bytes(int B, int num_anchors, int pre_nms, int post_nms);
SRCNN_API int srcnn_proposal_layer(const float *probs, const float *deltas, int B, int num_anchors,
                         const int *level_hw_host, int nlevels, const float *im_info /* (B,3) device */,
                         int pre_nms, int post_nms, float nms_thresh,
                         float *rois_left, float *rois_right, int *num_valid /* (B) device, may be NULL */,
                         void *workspace, size_t workspace_bytes, srcnn_stream_t stream);
/* Inspection (tests; not on the hot path): where srcnn_proposal_layer leaves its intermediates in the caller's workspace after
 * a call with the same (B, num_anchors, pre_nms) -- byte offsets of: [0] order (B x n int32: the n = min(pre_nms, A) best
 * anchors, descending score, ties by ascending index), [1] dets (B x 2 x n x 5 float32: decoded + clipped left / right boxes and
 * the score, in that order), [2] keep (B x 2 x n int32: ascending row indices that survive the NMS; the left / right scans of an
 * image stop together once post_nms rows survive in both, so each list is a PREFIX of the full greedy list), [3] num (B x 2
 * int32: entries of each keep list), [4] total bytes.  This is what lets a test compare the kernels' actual discrete decisions
 * with a reference run's (tests/tie_audit.py).  Returns SRCNN_OK, or SRCNN_ERR_ARG for n_offsets < 5. */
SRCNN_API int srcnn_proposal_workspace_layout(int B, int num_anchors, int pre_nms, size_t *offsets, int n_offsets);

/* ------------------------------------------------------------------ heads (A9-A12)
 * cls softmax over n_cls logits (stereo_rcnn.py:257). */
SRCNN_API int srcnn_softmax_rows(const float *x, int rows, int cols, int x_stride, float *y, srcnn_stream_t stream);
/* The box head's stacked fc output (rows, fc_stride >= n_bbox + n_dim + n_cls): columns [RCNN_bbox_pred | RCNN_dim_orien_pred |
 * RCNN_cls_score] (stereo_rcnn.py:251-257) split in ONE launch into the three tensors the forward returns -- the regressions
 * copied bit for bit into contiguous rows, the class logits through the softmax above. */
SRCNN_API int srcnn_box_head_tail(const float *fc, int rows, int n_bbox, int n_dim, int n_cls, int fc_stride, float *bbox_pred,
                                  float *dim_orien_pred, float *cls_prob, srcnn_stream_t stream);
/* keypoint tail (stereo_rcnn.py:262-271): logits (n, G, G, 6) NHWC from kpts_class ->
 * sum over H, softmax over 4*G (kpts) and G (left/right borders).  roi_limit: NULL, or a device int -- rows [0, *roi_limit) only. */
SRCNN_API int srcnn_kpts_tail(const float *logits, int n, int G, float *kpts_prob, float *left_prob, float *right_prob,
                    const int *roi_limit, srcnn_stream_t stream);
/* detection decode (demo.py:144-218, bbox_transform.py:133-155) for B == 1 blocks of n rois. */
/* Keypoint head on the kept detections only ("lazy" form of stereo_rcnn.py:260-271 + demo.py:196-209: the reference computes the
 * keypoint branch for all 300 rois and its scripts then read the rows that survive score threshold + NMS; every roi's
 * keypoint computation is independent of the others).
 * srcnn_gather_rows: dst[r] = src[max(idx[r], 0)] for r < n_idx (rows of `cols` floats; -1 padded index lists repeat row 0).
 * srcnn_decode_kept_kpts: for r < *num_keep, i = keep_idx[r]: kpts[i] (5 floats, as srcnn_decode_detections writes them)
 *   from row r of the kept-order probability arrays and roi i of rois_left; other rows of `kpts` are left alone. */
SRCNN_API int srcnn_gather_rows(const float *src, const int *idx, int n_idx, int cols, float *dst, srcnn_stream_t stream);
SRCNN_API int srcnn_decode_kept_kpts(const float *rois_left, const float *kpts_prob, const float *left_prob,
                                     const float *right_prob, const int *keep_idx, const int *num_keep, const float *im_info,
                                     int n, int G, float *kpts, srcnn_stream_t stream);
SRCNN_API int srcnn_decode_detections(const float *rois_left, const float *rois_right, const float *bbox_pred,
                            const float *dim_orien_pred, const float *kpts_prob, const float *left_prob,
                            const float *right_prob, const float *im_info, int n, int n_cls, int G,
                            float *boxes_left, float *boxes_right, float *dim_orien, float *kpts,
                            srcnn_stream_t stream);
/* per-class filter + sort + NMS (demo.py:231-257): scores (n, n_cls) column j; outputs the kept
 * ORIGINAL roi indices in descending score order and their count, all on the device. */
SRCNN_API size_t srcnn_class_nms_workspace_bytes(int n);
SRCNN_API int srcnn_class_nms(const float *scores, int n, int n_cls, int j, const float *boxes_left /* (n,4*n_cls) */,
                    float score_thresh, float nms_thresh, int *keep_idx, int *num_keep,
                    void *workspace, size_t workspace_bytes, srcnn_stream_t stream);

/* fixed-size, zero-padded detection record of one image: what the 3-D stage works on in place and what the multi-GPU
 * gather moves (new: the reference writes per-image txt files instead, kitti_utils.py:456-460).
 * rec ((n+1), rec_cols) float32, row 0 = [count, 0...], row 1+r = the r-th kept detection (descending score):
 *   0 score | 1-4 left box | 5-8 right box | 9-13 dim_orien (w,h,l,sin,cos) | 14-18 kpts (u,type,prob,left,right border)
 *   19 roi index | 20 4-DoF status | 21-24 x,y,z,theta of the 4-DoF solve (float32, = the reference's poses_all)
 *   25 dense-alignment status | 26 aligned disparity | 27-30 final x,y,z,theta | 31 alpha (poses_all[:,7])
 * srcnn_pack_detections fills columns 0-19 (rec_cols >= 20) and zeroes the rest; the 3-D calls need SRCNN_REC_COLS. */
#define SRCNN_REC_COLS 32
SRCNN_API int srcnn_pack_detections(const float *scores, const float *boxes_left, const float *boxes_right,
                          const float *dim_orien, const float *kpts, const int *keep_idx, const int *num_keep,
                          int n, int n_cls, int j, int rec_cols, float *rec, srcnn_stream_t stream);

/* ------------------------------------------------------------ dense alignment (A15-A16)
 * Replaces lib/model/dense_align/dense_align.py:13-69,175-300 + box_3d.py:12-106 (align_parallel).
 * im_left/right: (3, H, W) planar float32 network-input tensors; the 2x align_corners bilinear
 * upsample of dense_align.py:256-257 is done inside (workspace).  boxes (R,4) and borders (R,2)
 * in ORIGINAL-image pixels (borders = keypoints[:,3:5]), poses (R,7) [x,y,z,w,h,l,theta].
 * scale = im_info[0,2]; p2_00/p2_02/p2_12 = P2 focal/cx/cy; p2_03_minus_p3_03 = P2[0,3]-P3[0,3]
 * (host doubles, as in the reference).  max_pixels bounds the per-object sample count; the reference's lattice has at
 * most 113 x 46 points, so 8192 can never overflow -- an object whose lattice does not fit a smaller bound is reported
 * with status -1 (never silently truncated).  Outputs status (R: 1 ok, 0 no valid pixel, -1 overflow) and best_dis (R)
 * float32 on the device. */
SRCNN_API size_t srcnn_dense_align_workspace_bytes(int H, int W, int R, int max_pixels);
/* Where, inside the workspace of a finished srcnn_dense_align call with the same sizes, the intermediate results of the depth
 * search live (byte offsets; for parity audits of the discrete argmin -- reference dense_align.py:225-232 -- and diagnostics):
 *   offsets[0] = int32 cnt[2 R]: valid lattice pixels per object, then overflow flags
 *   offsets[1], [2], [3] = coarse stage: float depth_enum[50][R], float cost[50][R] (sum over pixels and channels of |L - R|,
 *                          workgroup-reduced in a fixed order), float best_depth[R] (first minimum)
 *   offsets[4], [5], [6] = fine stage: depth_enum[20][R] (rows 20..49 unused), cost[20][R], best_depth[R]
 * n_offsets must be >= 7. */
SRCNN_API int srcnn_dense_align_workspace_layout(int H, int W, int R, int max_pixels, size_t *offsets, int n_offsets);
SRCNN_API int srcnn_dense_align(const float *im_left, const float *im_right, int H, int W, double scale,
                      double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03,
                      const float *boxes, const float *borders, const float *poses,
                      const float *valid /* (R) or NULL: rows with valid <= 0 are skipped (status 0), so a fixed-size
                                            batch can be aligned without compacting it on the host */,
                      int R, int max_pixels,
                      float *status, float *best_dis, void *workspace, size_t workspace_bytes,
                      srcnn_stream_t stream);

/* ------------------------------------------------------------ 3-D stage on the device (A13, A14, A17)
 * All three work IN PLACE on one image's record (see srcnn_pack_detections), asynchronously, no host round trip.
 * srcnn_infer_boundary: kitti_utils.py:398-437 (occlusion "depth line" over the image columns from the left boxes) and
 *   the replacement rule of demo.py:262-265 -- columns 17/18 are overwritten where the regressed borders are narrower
 *   than half the inferred ones.  workspace: srcnn_box3d_workspace_bytes(n, im_w).
 * srcnn_solve_4dof: demo.py:282-302 = box_estimator.py:169-385 for every row with score > eval_thresh: scipy's Newton-CG
 *   (optimize/_optimize.py:_minimize_newtoncg + MINPACK-2 dcsrch / wolfe2 line searches) restated in double precision,
 *   one detection per workgroup, the eight re-projection residuals of every cost / gradient evaluation on eight lanes of its
 *   wavefront, summed in lane order (csrc/box_solver_wave.h).  Writes columns 20-24, 27-31 and state4 (n,4) doubles
 *   (x, y, z, theta).  srcnn_solve_4dof_scalar / srcnn_solve_3dof_scalar: the same iteration with one lane evaluating the
 *   residuals one after another (the form of rounds 2-5) -- bit-identical results, ~4x the time; kept as the lane form's test.
 * srcnn_align_inputs: the arrays align_parallel takes (boxes (n,4), borders (n,2), poses (n,7), valid (n)) from the record.
 * srcnn_solve_3dof: demo.py:311-319 = box_estimator.py:387-545 for rows whose 4-DoF solve and dense alignment
 *   succeeded (align_status / best_dis (n) from srcnn_dense_align, or both NULL = no alignment): columns 25-30 and
 *   state (n,4) doubles (x, y, z = f b / disparity, theta).
 * P2 / P3 entries are host doubles as in the reference (calib.p2[0,0], [0,2], [1,2], p2[0,3] - p3[0,3]).
 *
 * PARITY GRADE of the two device solvers: numerically equivalent to the reference, NOT reference-identical.  They run the
 * reference's iteration (same Newton-CG, same line searches, same stopping rules) with ROCm ocml's cos / sin / atan2 and
 * exact squares where the reference's scipy / numpy path goes through glibc's libm and pow(v, 2).  Those differ in the last
 * bit, and the reference's end point is not a smooth function of its inputs (its gradient is not the gradient of its cost;
 * the iteration stops on scipy's step tolerance, 1e-3..1e-2 short of the optimum): against the host build, on the same
 * inputs, 72 % of the 4-DoF and 98 % of the 3-DoF end points of well-posed cases agree within 1e-4 and the rest jump like
 * the reference itself does when its inputs move by 1e-5 (tests/test_box3d_gpu.py, tests/test_box3d_conditioning.py,
 * DESIGN.md section 7).  Status columns are identical.  Callers that need the reference's boxes bit for bit use the record
 * forms on host memory below (srcnn_solve_*_records_host: the default of stereo_rcnn_amd.pipeline, solver='host').
 *
 * "Bit-identical to the reference's scipy path" is a statement about NumPy >= 2 (NEP 50) scalar promotion, the NumPy of
 * this image: 1050.0f / b[3] in infer_boundary and the float32-row arithmetic of the solvers' box-size tests, start
 * disparity and kpt2alpha ratio are evaluated in float32, as numpy 2 does for python-scalar (op) float32.  Under the
 * reference's own era (numpy 1.x value-based casting) those four expressions are float64; tie cases of depth < pixel and
 * the Newton-CG start point then differ in last bits -- srcnn_solve_4dof_host(..., boxes_are_float32 = 0) gives that
 * arithmetic for the solver. */
SRCNN_API size_t srcnn_box3d_workspace_bytes(int n, int im_w);
SRCNN_API int srcnn_infer_boundary(float *rec, int n, int rec_cols, int im_w, void *workspace, size_t workspace_bytes,
                         srcnn_stream_t stream);
SRCNN_API int srcnn_solve_4dof(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                     double p2_03_minus_p3_03, float eval_thresh, double *state4, srcnn_stream_t stream);
SRCNN_API int srcnn_solve_4dof_scalar(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                     double p2_03_minus_p3_03, float eval_thresh, double *state4, srcnn_stream_t stream);
SRCNN_API int srcnn_align_inputs(const float *rec, int n, int rec_cols, float *boxes, float *borders, float *poses, float *valid,
                       srcnn_stream_t stream);
SRCNN_API int srcnn_solve_3dof(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                     double p2_03_minus_p3_03, const float *align_status, const float *best_dis, double *state,
                     srcnn_stream_t stream);
SRCNN_API int srcnn_solve_3dof_scalar(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                     double p2_03_minus_p3_03, const float *align_status, const float *best_dis, double *state,
                     srcnn_stream_t stream);
/* Record forms on HOST memory (pinned copies of `rec` / `state`): row for row the function the two kernels above run, built
 * for the host and using the host's libm, i.e. bit-identical to the reference's scipy path (scipy Newton-CG, numpy scalar
 * `**2` = pow, glibc cos / sin / atan2).  Rows are spread over host threads: one per 8 rows, at most 16, and at most `threads` when > 0
 * (a budget, not a demand: creating 16 threads for 40 rows costs more than the rows). */
SRCNN_API int srcnn_solve_4dof_records_host(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02,
                                  double p2_12, double p2_03_minus_p3_03, float eval_thresh, double *state4, int threads);
SRCNN_API int srcnn_solve_3dof_records_host(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02,
                                  double p2_12, double p2_03_minus_p3_03, const float *align_status, const float *best_dis,
                                  double *state, int threads);
/* the same solver code compiled for the host (HOST pointers, no GPU needed): the reference's
 * solve_x_y_z_theta_from_kpt(im_shape, calib, alpha, dim, box_left, box_right, kpts) -> (status, state) and
 * solve_x_y_theta_from_kpt(im_shape, calib, alpha, dim, box_left, disparity, kpts) -> (state, z), flattened.
 * newton_status (may be NULL): scipy's OptimizeResult.status (0 ok, 1 maxiter, 2 line search, 3 CG / NaN), -1 = early-out. */
SRCNN_API int srcnn_solve_4dof_host(int im_h, int im_w, double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03,
                          double alpha, const double *dim3, const double *box_left4, const double *box_right4,
                          const double *kpts5, double *state4, int *newton_status,
                          int boxes_are_float32 /* the boxes hold float32 values the reference's numpy evaluates in float32
                                                   (box-size early-outs, start disparity); the device kernels always do */);
                          /* returns status 0 / 1 */
SRCNN_API int srcnn_solve_3dof_host(int im_h, int im_w, double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03,
                          double alpha, const double *dim3, const double *box_left4, double disparity,
                          const double *kpts5, double *state3, double *z, int *newton_status);
/* cost and the reference's (non-)gradient of either problem at (x, y, z, theta): box_right4 NULL = the 3-DoF terms */
SRCNN_API int srcnn_solver_evaluate_host(int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                               double p2_03_minus_p3_03, double alpha, const double *dim3, const double *box_left4,
                               const double *box_right4_or_null, const double *kpts5, const double *xyzt, double *cost,
                               double *grad4);

/* ------------------------------------------------------------------ recorded launch programs
 * The forward is a fixed list of ~230 asynchronous launches over fixed buffers (one list per input size and buffer set).
 * Between srcnn_program_begin and srcnn_program_end every srcnn_* call made BY THE CALLING THREAD records its kernel
 * launches / memsets (function, geometry, stream, by-value arguments) into the program instead of launching them;
 * srcnn_program_run then re-issues the whole list from C on the streams it was recorded with, the main stream replaced by
 * the one given -- no Python frame, descriptor filling or plan lookup per launch.  Dependencies between streams are
 * recorded explicitly: record_event (returns an event id >= 0) on the producing stream, wait_event on the consuming one.
 * The caller keeps every buffer a recorded launch points at alive and unchanged in address (activations, weights,
 * workspaces).  Unlike a hipGraph the side-stream branches stay real concurrent streams at replay. */
SRCNN_API void *srcnn_program_create(void);
SRCNN_API void srcnn_program_destroy(void *prog);
SRCNN_API int srcnn_program_begin(void *prog, srcnn_stream_t main_stream);
SRCNN_API int srcnn_program_end(void *prog);
SRCNN_API int srcnn_program_recording(void);                        /* 1 while the calling thread records */
SRCNN_API int srcnn_program_record_event(void *prog, srcnn_stream_t stream);
SRCNN_API int srcnn_program_wait_event(void *prog, srcnn_stream_t stream, int event_id);
SRCNN_API int srcnn_program_size(void *prog);                       /* recorded nodes */
SRCNN_API int srcnn_program_run(void *prog, srcnn_stream_t main_stream);

/* ------------------------------------------------------------------ streams with their own hardware queue
 * HIP folds the streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues, least-used queue first.  Streams that
 * share a queue execute in submission order and an event wait of one of them holds up everything queued behind it -- with
 * several forwards in flight (one main stream + two side streams each) a side stream of forward A that waits for A's trunk
 * stalls the main stream of forward B that happens to share its queue (profiles/queue_mapping_r04.txt).  A stream created
 * here owns a hardware queue of its own (created with an all-ones CU mask: masked streams are never pooled), so that the
 * placement of the forwards' streams is the caller's decision, not an accident of creation order.
 * dedicated_queue = 0 gives an ordinary non-blocking stream from the shared pool. */
SRCNN_API int srcnn_stream_create(int dedicated_queue, srcnn_stream_t *stream);
/* The same with a caller-chosen CU mask (bit b of word b / 32 enables compute unit b of the device's enumeration; `words`
 * 32-bit words, at least one bit set in them): the kernels of this stream run on those CUs only.  The serving regime gives each
 * of S forwards in flight its own 1/S of every XCD's CUs (stereo_rcnn_amd/streams.py: partition_masks) -- a forward then never
 * shares a CU with another forward's kernels, and its launches meet a device of 256/S CUs that their tile counts fill. */
SRCNN_API int srcnn_stream_create_cu_mask(int words, const unsigned *mask, srcnn_stream_t *stream);
SRCNN_API int srcnn_stream_destroy(srcnn_stream_t stream);
/* Diagnostics: launches `blocks` one-wave workgroups on `stream`; block b writes its XCC_ID register to xcc[b] and its HW_ID
 * register (CU / shader-array / shader-engine fields) to hw_id[b] (device int arrays of `blocks` entries).  Shows where a
 * (masked) stream's workgroups actually run. */
SRCNN_API int srcnn_probe_placement(int blocks, int *xcc, int *hw_id, srcnn_stream_t stream);

/* ------------------------------------------------------------------ profiling hooks
 * When enabled, every conv-engine launch is bracketed by hipEvents on its stream; the
 * accumulated kernel time / algorithmic flops / launch count are read back with
 * srcnn_prof_read (which synchronises those events). Used by bench.py for `roofline`. */
SRCNN_API int srcnn_prof_enable(int on);
SRCNN_API int srcnn_prof_read(double *conv_ms, double *conv_flops, long long *conv_launches);
/* Per-launch form: elapsed ms of each recorded conv launch, in launch order, into ms[0..max) (same synchronisation);
 * returns the number of launches recorded (which may exceed max) or a negative error.  Does not reset the recording --
 * srcnn_prof_read / srcnn_prof_enable do.  Used by bench.py / tools/layer_table.py for the per-layer roofline table. */
SRCNN_API int srcnn_prof_read_launches(float *ms, int max);

#ifdef __cplusplus
}
#endif
#endif /* SRCNN_HIP_H */
