// Shared declarations of the convolution engines (fp32 MFMA and 3xf16 split MFMA).
#pragma once
#include "common.h"

namespace srcnn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;          // floats per K tile (128 B)
constexpr int LDS_ROW = BK + 4; // padded row (floats)

struct ConvArgs {
    const float *x, *w, *bias, *res;
    float *y, *partial;
    const void *w_lo;   // f16x3 engine: w = hi halves, w_lo = lo halves (both (Cout, K) _Float16)
    float out_scale;    // f16x3 engine: 1 / (power-of-two weight scale)
    int x_fmt, y_fmt, res_fmt;    // SRCNN_FMT_F32 / SRCNN_FMT_SPLIT16
    int nimg;                     // images reachable from x (descriptor range of the SPLIT16 engine's DMA)
    int H, W, Cin, xcs;
    int OH, OW, Cout;
    int KH, KW, stride, pad;
    int ycs, yco, rcs, relu, mode;
    int tap_inner;                // SPLIT16 engine, KH*KW > 1: K runs (channel tile, tap) instead of (tap, channel tile)
    int m_fast;                   // SPLIT16 engine: consecutive workgroups walk the M tiles of one N tile (weights > activations)
    int M, K;
    int ctiles;       // Cin / 32
    int nkt;          // K tiles in total
    int kt_per_split; // K tiles per grid.y slice
    int mtiles, ntiles;
    unsigned long long *stamp;   // debug: per-workgroup phase timestamps (conv_f16s only), normally nullptr
    unsigned *range_flag;        // SPLIT16 range guard (see split16_guard); never null for SPLIT16 outputs
    int tag;                     // caller's layer tag (>= 0): the flag keeps the largest tag + 1 that tripped
    const int *m_limit;          // device-side row limit (rows m >= *m_limit * m_limit_mul are not needed) or nullptr
    int m_limit_mul;
    // SPLIT16 engine, 1x1 convs: a second input whose 1x1 / stride2 projection is K-concatenated (weights (Cout, Cin + Cin2)):
    // the ResNet projection shortcut computed by the block's last conv itself.  nullptr = none.
    const float *x2;
    int Cin2, H2, W2, xcs2, stride2;
    // SPLIT16 engine, 256x256 tile: fused 6-channel 1x1 head on the activated output pixel (srcnn_conv_desc.head_w); nullptr = none
    const float *head_w, *head_b;
    float *head_y;
    float head_scale;
    // ... or, MFMA form (srcnn_conv_desc.head_wf; 256x256 and 128x128 8-wave tiles): the head as a second, narrow GEMM on the tile the
    // epilogue holds in LDS.  head_wf = the head's weights split into hi / lo f16 and laid out in MFMA fragment order; head_rows =
    // its outputs padded to a multiple of 8 (<= 32), head_n the real ones; head_parts 0 = the tile owns all channels of its pixels:
    // bias added, head_n floats per pixel stored; > 0 = per-(eye, N tile) partial sums into planes of head_plane floats each.
    const void *head_wf;
    int head_rows, head_n, head_parts;
    long long head_plane;
    // SPLIT16 engine, every tile but 256x256: the FPN top-down addition in the lateral conv's epilogue (srcnn_conv_desc.up_top):
    // y = bilinear_align_corners(up_top (nimg, up_TH, up_TW, Cout) -> (OH, OW)) + (conv + bias); nullptr = none
    const void *up_top;
    int up_fmt, up_TH, up_TW;
    int nt_out;                  // SPLIT16 engine: results leave with non-temporal stores (A/B switch SRCNN_NT_STORES)
};


// ---- "split16" activation format: per pixel, every group of 8 channels is stored as
// [8 x f16 hi][8 x f16 lo] (32 B, same footprint as fp32) with hi = f16(v), lo = f16(v - hi).
// A K tile of 32 channels is one contiguous 128-B run whose 16-B chunks are exactly the MFMA
// operand chunks, so the conv engine can DMA them straight into LDS (global_load_lds).
__device__ __forceinline__ size_t split16_byte_off(size_t pixel, int cstride, int ch)
{
    return pixel * (size_t)cstride * 4 + (size_t)(ch >> 3) * 32 + (size_t)(ch & 7) * 2;
}

__device__ __forceinline__ float act_load(const void *base, int fmt, size_t pixel, int cstride, int ch)
{
    if (fmt == 0) return reinterpret_cast<const float *>(base)[pixel * (size_t)cstride + ch];
    const char *p = reinterpret_cast<const char *>(base) + split16_byte_off(pixel, cstride, ch);
    return (float)*reinterpret_cast<const _Float16 *>(p) + (float)*reinterpret_cast<const _Float16 *>(p + 16);
}

__device__ __forceinline__ void act_store(void *base, int fmt, size_t pixel, int cstride, int ch, float v)
{
    if (fmt == 0) {
        reinterpret_cast<float *>(base)[pixel * (size_t)cstride + ch] = v;
        return;
    }
    char *p = reinterpret_cast<char *>(base) + split16_byte_off(pixel, cstride, ch);
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16 *>(p) = hi;
    *reinterpret_cast<_Float16 *>(p + 16) = (_Float16)(v - (float)hi);
}

// 8-channel group accessors (32 B in either format): the unit the HBM-bound helper kernels work on.
struct float8 {
    float v[8];
};

__device__ __forceinline__ float8 act_load8(const void *base, int fmt, size_t pixel, int cstride, int group)
{
    const char *p = reinterpret_cast<const char *>(base) + (pixel * (size_t)cstride + (size_t)group * 8) * 4;
    float8 r;
    if (fmt == 0) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 16);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 hi = *reinterpret_cast<const h8 *>(p), lo = *reinterpret_cast<const h8 *>(p + 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[e] = (float)hi[e] + (float)lo[e];
    }
    return r;
}

// (NT: non-temporal stores -- results no wave of this launch reads again; behind a kernel boundary the next launch finds nothing in
//  the XCDs' L2s anyway, so the lines only displace the operands this launch is still re-reading)
template <bool NT = false>
__device__ __forceinline__ void act_store8(void *base, int fmt, size_t pixel, int cstride, int group, const float8 &r)
{
    char *p = reinterpret_cast<char *>(base) + (pixel * (size_t)cstride + (size_t)group * 8) * 4;
    if constexpr (NT) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        if (fmt == 0) {
            const f4 a = {r.v[0], r.v[1], r.v[2], r.v[3]}, b = {r.v[4], r.v[5], r.v[6], r.v[7]};
            __builtin_nontemporal_store(a, reinterpret_cast<f4 *>(p));
            __builtin_nontemporal_store(b, reinterpret_cast<f4 *>(p + 16));
        } else {
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = (_Float16)r.v[e];
                lo[e] = (_Float16)(r.v[e] - (float)hi[e]);
            }
            __builtin_nontemporal_store(hi, reinterpret_cast<h8 *>(p));
            __builtin_nontemporal_store(lo, reinterpret_cast<h8 *>(p + 16));
        }
        return;
    }
    if (fmt == 0) {
        *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
        *reinterpret_cast<float4 *>(p + 16) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = (_Float16)r.v[e];
            lo[e] = (_Float16)(r.v[e] - (float)hi[e]);
        }
        *reinterpret_cast<h8 *>(p) = hi;
        *reinterpret_cast<h8 *>(p + 16) = lo;
    }
}

// SPLIT16 range guard.  hi = f16(v) has no scaling: |v| > 65504 (or a NaN) turns into inf and poisons every product it
// enters, where the fp32 engine would carry on.  Every kernel that WRITES the format from fresh arithmetic (the conv
// epilogues, upsample_add) calls this on the 8 values of a group; one atomicMax per offending group leaves (layer tag + 1) in a
// library-owned device word (srcnn_range_flag_read), so that the host can re-run the forward on the exact fp32 engine
// instead of returning garbage.  In range -- the normal case -- it costs 8 max + 1 compare per group and no memory access.
__device__ __forceinline__ void split16_guard(const float8 &r, unsigned *flag, int tag)
{
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(r.v[e]));
    bool bad = !(m <= 65504.f);                 // also true for NaN
#pragma unroll
    for (int e = 0; e < 8; ++e) bad = bad || (r.v[e] != r.v[e]);
    if (bad) atomicMax(flag, (unsigned)(tag + 1));
}

unsigned *range_flag_word();   // core.hip: library-owned, zero-initialised device word

struct Plan {
    int mr, nr, splits, kt_per_split;   // workgroup tile (64*mr) x (64*nr); K slices
    int waves = 4, stages = 2;          // wavefronts per workgroup; LDS ring depth (conv_f16s only)
};

void launch_conv_f16x3(const ConvArgs &a, const Plan &pl, hipStream_t st);   // A operand fp32 in HBM
void launch_conv_f16s(const ConvArgs &a, const Plan &pl, hipStream_t st);    // A operand split16 in HBM
bool conv_f16s_plan_ok(const Plan &pl, const ConvArgs &a);
// conv_mfma.hip: validates a descriptor and fills the kernel arguments (everything but the tile counts / split-K fields)
int conv_fill_args(const srcnn_conv_desc *d, ConvArgs &a);

}  // namespace srcnn
