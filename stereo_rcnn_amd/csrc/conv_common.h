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
    int H, W, Cin, xcs;
    int OH, OW, Cout;
    int KH, KW, stride, pad;
    int ycs, yco, rcs, relu, mode;
    int M, K;
    int ctiles;       // Cin / 32
    int nkt;          // K tiles in total
    int kt_per_split; // K tiles per grid.y slice
    int mtiles, ntiles;
};


struct Plan {
    int mr, nr, splits, kt_per_split;
};

void launch_conv_f16x3(const ConvArgs &a, const Plan &pl, hipStream_t st);

}  // namespace srcnn
