// The 3-D stage of the path on the device: occlusion borders, the 4-DoF box solve and the
// 3-DoF rectification, all working in place on the per-image detection record (srcnn_pack_detections), so that
// demo.py:259-326 runs without a host round trip between the detector and the final 3-D boxes:
//     class NMS -> pack -> [infer_boundary + border replacement] -> [4-DoF solve] -> dense alignment -> [3-DoF solve]
// The solvers themselves are box_solver.h (scipy's Newton-CG restated; the same code also builds for the host and is
// exported as srcnn_solve_*_host for CPU-side callers and tests).  One detection per 64-lane workgroup, so that the objects of
// an image spread over up to 300 CUs instead of diverging inside one wavefront; within the wavefront the eight residuals of
// every cost / gradient evaluation sit on eight lanes (box_solver_wave.h; round 6).  The form of rounds 2-5 -- lane 0 runs
// the scalar code -- is kept as srcnn_solve_*_scalar: the two are bit-identical, which is how the lane form is tested.
#include "common.h"
#include "box_solver_wave.h"
#include <atomic>
#include <thread>
#include <vector>

namespace srcnn {

using namespace boxsolve;

// record columns (include/srcnn_hip.h: SRCNN_REC_*)
enum { C_SCORE = 0, C_BOXL = 1, C_BOXR = 5, C_DIM = 9, C_SIN = 12, C_COS = 13, C_KPT = 14, C_BORDER = 17, C_ROI = 19,
       C_ST4 = 20, C_POSE4 = 21, C_ALIGN = 25, C_DISP = 26, C_POSE = 27, C_ALPHA = 31 };

// kitti_utils.py:398-437 + demo.py:261-265.  One workgroup per image record: phase 1, one thread per image column walks the
// detections in order (a column's "depth line" value only depends on the boxes that cover it, in their order); phase 2, one
// thread per detection scans its columns; the replacement test of demo.py follows.
__global__ void infer_boundary_kernel(float *__restrict__ rec, int n, int cols, int im_w, double *__restrict__ line)
{
    const int k = min((int)rec[0], n);
    for (int col = threadIdx.x; col <= im_w; col += blockDim.x) {
        double pixel = 0.0;
        for (int i = 0; i < k; ++i) {
            const float *b = rec + (size_t)(1 + i) * cols + C_BOXL;
            if (col < (int)b[0] || col > (int)b[2]) continue;
            const double depth = (double)(1050.0f / b[3]);     // Python float / np.float32 -> float32 (as run for the goldens)
            if (pixel == 0.0) pixel = depth;
            else if (depth < pixel) pixel = (depth + pixel) / 2.0;
        }
        line[col] = pixel;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        float *row = rec + (size_t)(1 + i) * cols;
        const float *b = row + C_BOXL;
        const double d = (double)(1050.0f / b[3]);
        const int x1 = min(max((int)b[0], 0), im_w), x2 = min(max((int)b[2], 0), im_w);
        float left = b[0], right = b[2];
        const bool left_visible = !(line[x1] < d), right_visible = !(line[x2] < d);
        if (!right_visible && !left_visible) right = b[0];
        for (int col = x1; col <= x2; ++col) {
            if (left_visible && line[col] >= d) right = (float)col;
            else if (right_visible && line[col] < d) left = (float)col;
        }
        // demo.py:262-265 (Python floats under torch 0.3: exact on float32 operands in double)
        if ((double)row[C_BORDER + 1] - (double)row[C_BORDER] < 0.5 * ((double)right - (double)left)) {
            row[C_BORDER] = left;
            row[C_BORDER + 1] = right;
        }
    }
}

struct Calib {
    double f, cx, cy, base;
    int im_h, im_w;
};

SRCNN_HD inline void load5(const float *p, double *o)
{
    for (int i = 0; i < 5; ++i) o[i] = (double)p[i];
}

// demo.py:282-302 for detection i of a record: (status, x, y, z, theta), poses kept in float32 as `poses_all`.
// Host and device run this same function (the host build through srcnn_solve_4dof_records_host).
SRCNN_HD inline void solve4_row(float *rec, int n, int cols, const Calib &c, float eval_thresh, double *state4, int i)
{
    const int k = (int)rec[0] < n ? (int)rec[0] : n;
    float *row = rec + (size_t)(1 + i) * cols;
    double *out = state4 + (size_t)i * 4;
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (i >= k) return;
    row[C_ST4] = 0.f;
    row[C_ALIGN] = 0.f;
    if (!(row[C_SCORE] > eval_thresh)) return;
    double bl[5], br[5], dim[5], kp[5];
    load5(row + C_BOXL, bl);
    load5(row + C_BOXR, br);
    load5(row + C_DIM, dim);
    load5(row + C_KPT, kp);
    const double alpha = atan2((double)row[C_SIN], (double)row[C_COS]);          // demo.py:288-290
    double st[4];
    const int status = solve_4dof(c.im_h, c.im_w, c.f, c.cx, c.cy, c.base, alpha, dim, bl, br, kp, st, nullptr, true);
    for (int q = 0; q < 4; ++q) out[q] = st[q];
    row[C_ST4] = (float)status;
    for (int q = 0; q < 4; ++q) row[C_POSE4 + q] = (float)st[q];                  // poses[0..2], poses[6]
    for (int q = 0; q < 4; ++q) row[C_POSE + q] = (float)st[q];
    row[C_ALPHA] = (float)alpha;                                                  // poses[7]
}

__global__ void solve4_kernel(float *__restrict__ rec, int n, int cols, Calib c, float eval_thresh, double *__restrict__ state4)
{
    if (threadIdx.x == 0) solve4_row(rec, n, cols, c, eval_thresh, state4, blockIdx.x);
}

// solve4_row with the residuals of every evaluation across the lanes of the workgroup's one wavefront: all 64 lanes run the
// row (same loads, same control flow), lane 0 stores.  __launch_bounds__(64): the optimiser state of four nested line-search
// levels wants ~200 VGPRs; at the default bound (1024 threads, 128 VGPRs) the scalar kernel spills 94 of them to scratch.
__global__ void __launch_bounds__(64) solve4_wave_kernel(float *__restrict__ rec, int n, int cols, Calib c, float eval_thresh,
                                                         double *__restrict__ state4)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    const bool writer = lane == 0;
    const int k = (int)rec[0] < n ? (int)rec[0] : n;
    float *row = rec + (size_t)(1 + i) * cols;
    double *out = state4 + (size_t)i * 4;
    if (writer) out[0] = out[1] = out[2] = out[3] = 0.0;
    if (i >= k) return;
    const bool scored = row[C_SCORE] > eval_thresh;
    double bl[5], br[5], dim[5], kp[5];
    load5(row + C_BOXL, bl);
    load5(row + C_BOXR, br);
    load5(row + C_DIM, dim);
    load5(row + C_KPT, kp);
    const double alpha = atan2((double)row[C_SIN], (double)row[C_COS]);
    if (writer) row[C_ST4] = row[C_ALIGN] = 0.f;
    if (!scored) return;
    double st[4];
    const int status = solve_4dof_wave(c.im_h, c.im_w, c.f, c.cx, c.cy, c.base, alpha, dim, bl, br, kp, st, true, lane);
    if (!writer) return;
    for (int q = 0; q < 4; ++q) out[q] = st[q];
    row[C_ST4] = (float)status;
    for (int q = 0; q < 4; ++q) row[C_POSE4 + q] = (float)st[q];
    for (int q = 0; q < 4; ++q) row[C_POSE + q] = (float)st[q];
    row[C_ALPHA] = (float)alpha;
}

// gather of what align_parallel takes (demo.py:306-308): boxes (n,4), borders (n,2), poses (n,7), valid (n)
__global__ void align_inputs_kernel(const float *__restrict__ rec, int n, int cols, float *__restrict__ boxes,
                                    float *__restrict__ borders, float *__restrict__ poses, float *__restrict__ valid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = min((int)rec[0], n);
    const float *row = rec + (size_t)(1 + i) * cols;
    const bool ok = i < k && row[C_ST4] > 0.f;
    valid[i] = ok ? 1.f : 0.f;
    for (int q = 0; q < 4; ++q) boxes[i * 4 + q] = ok ? row[C_BOXL + q] : 0.f;
    borders[i * 2 + 0] = ok ? row[C_BORDER] : 0.f;
    borders[i * 2 + 1] = ok ? row[C_BORDER + 1] : 0.f;
    poses[i * 7 + 0] = ok ? row[C_POSE4 + 0] : 0.f;
    poses[i * 7 + 1] = ok ? row[C_POSE4 + 1] : 0.f;
    poses[i * 7 + 2] = ok ? row[C_POSE4 + 2] : 1.f;
    poses[i * 7 + 3] = ok ? row[C_DIM + 0] : 1.f;
    poses[i * 7 + 4] = ok ? row[C_DIM + 1] : 1.f;
    poses[i * 7 + 5] = ok ? row[C_DIM + 2] : 1.f;
    poses[i * 7 + 6] = ok ? row[C_POSE4 + 3] : 0.f;
}

// demo.py:311-319: objects the alignment succeeded on are re-solved with z fixed by the aligned disparity
SRCNN_HD inline void solve3_row(float *rec, int n, int cols, const Calib &c, const float *align_status, const float *best_dis,
                                double *state, int i)
{
    const int k = (int)rec[0] < n ? (int)rec[0] : n;
    if (i >= k) return;
    float *row = rec + (size_t)(1 + i) * cols;
    double *out = state + (size_t)i * 4;
    if (!(row[C_ST4] > 0.f)) return;
    const float ast = align_status ? align_status[i] : 0.f;
    row[C_ALIGN] = ast;
    if (!(ast > 0.f)) return;            // (ast < 0: lattice overflow, reported to the host through the record)
    row[C_DISP] = best_dis[i];
    double bl[5], dim[5], kp[5];
    load5(row + C_BOXL, bl);
    load5(row + C_DIM, dim);
    load5(row + C_KPT, kp);
    double st[3];
    // alpha and dim are read back from the float32 `poses_all` (demo.py:313-316)
    const double z = solve_3dof(c.im_h, c.im_w, c.f, c.cx, c.cy, c.base, (double)row[C_ALPHA], dim, bl, (double)best_dis[i], kp,
                                st, nullptr);
    out[0] = st[0]; out[1] = st[1]; out[2] = z; out[3] = st[2];
    for (int q = 0; q < 4; ++q) row[C_POSE + q] = (float)out[q];
}

__global__ void solve3_kernel(float *__restrict__ rec, int n, int cols, Calib c, const float *__restrict__ align_status,
                              const float *__restrict__ best_dis, double *__restrict__ state)
{
    if (threadIdx.x == 0) solve3_row(rec, n, cols, c, align_status, best_dis, state, blockIdx.x);
}

// solve3_row, residuals across lanes (see solve4_wave_kernel)
__global__ void __launch_bounds__(64) solve3_wave_kernel(float *__restrict__ rec, int n, int cols, Calib c,
                                                         const float *__restrict__ align_status,
                                                         const float *__restrict__ best_dis, double *__restrict__ state)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    const bool writer = lane == 0;
    const int k = (int)rec[0] < n ? (int)rec[0] : n;
    if (i >= k) return;
    float *row = rec + (size_t)(1 + i) * cols;
    double *out = state + (size_t)i * 4;
    if (!(row[C_ST4] > 0.f)) return;
    const float ast = align_status ? align_status[i] : 0.f;
    if (writer) row[C_ALIGN] = ast;
    if (!(ast > 0.f)) return;
    const float dis = best_dis[i];
    double bl[5], dim[5], kp[5];
    load5(row + C_BOXL, bl);
    load5(row + C_DIM, dim);
    load5(row + C_KPT, kp);
    const double alpha = (double)row[C_ALPHA];
    double st[3];
    const double z = solve_3dof_wave(c.im_h, c.im_w, c.f, c.cx, c.cy, c.base, alpha, dim, bl, (double)dis, kp, st, lane);
    if (!writer) return;
    row[C_DISP] = dis;
    out[0] = st[0]; out[1] = st[1]; out[2] = z; out[3] = st[2];
    for (int q = 0; q < 4; ++q) row[C_POSE + q] = (float)out[q];
}

// rows [0, n) of a host record over host threads: one per 8 rows, at most 16 -- and at most `threads` when the caller gives a budget
// (> 0).  The budget is an upper bound, not a demand (round 6): a solve of ~40 detections takes 0.4 ms on 5 threads and 0.5-1.1 ms on
// 16, whose creation costs more than the rows they take (profiles/flow3d_async_host_r06.txt).
template <typename F>
static void host_rows(int n, int threads, F &&row_fn)
{
    int want = n / 8;
    want = want < 1 ? 1 : (want > 16 ? 16 : want);
    threads = threads > 0 && threads < want ? threads : want;
    if (threads > n) threads = n > 0 ? n : 1;
    if (threads == 1) {
        for (int i = 0; i < n; ++i) row_fn(i);
        return;
    }
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) row_fn(i);   // rows are independent
        });
    for (auto &th : pool) th.join();
}

static Calib make_calib(int im_h, int im_w, double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03)
{
    Calib c;
    c.f = p2_00;
    c.cx = p2_02;
    c.cy = p2_12;
    c.base = p2_03_minus_p3_03 / p2_00;          // bl = (P2[0,3] - P3[0,3]) / f
    c.im_h = im_h;
    c.im_w = im_w;
    return c;
}

}  // namespace srcnn

extern "C" {

size_t srcnn_box3d_workspace_bytes(int n, int im_w)
{
    using namespace srcnn;
    n = n > 0 ? n : 1;
    return align_up((size_t)(im_w + 2) * sizeof(double), 256) + 4 * align_up((size_t)n * 7 * sizeof(float), 256);
}

int srcnn_infer_boundary(float *rec, int n, int rec_cols, int im_w, void *workspace, size_t workspace_bytes,
                         srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rec && n > 0 && rec_cols >= SRCNN_REC_COLS && im_w > 0, "bad args (rec_cols >= SRCNN_REC_COLS)");
    SRCNN_REQUIRE(workspace && workspace_bytes >= srcnn_box3d_workspace_bytes(n, im_w), "workspace too small");
    SRCNN_LAUNCH(infer_boundary_kernel, dim3(1), dim3(1024), 0, as_stream(stream), rec, n, rec_cols, im_w,
                       static_cast<double *>(workspace));
    return check_launch("srcnn_infer_boundary");
}

static int launch_solve4(bool wave, float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                         double p2_03_minus_p3_03, float eval_thresh, double *state4, srcnn_stream_t stream, const char *who)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rec && state4 && n > 0 && rec_cols >= SRCNN_REC_COLS, "bad args (rec_cols >= SRCNN_REC_COLS)");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    if (wave) SRCNN_LAUNCH(solve4_wave_kernel, dim3(n), dim3(64), 0, as_stream(stream), rec, n, rec_cols, c, eval_thresh, state4);
    else SRCNN_LAUNCH(solve4_kernel, dim3(n), dim3(64), 0, as_stream(stream), rec, n, rec_cols, c, eval_thresh, state4);
    return check_launch(who);
}

int srcnn_solve_4dof(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                     double p2_03_minus_p3_03, float eval_thresh, double *state4, srcnn_stream_t stream)
{
    return launch_solve4(true, rec, n, rec_cols, im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03, eval_thresh, state4, stream,
                         "srcnn_solve_4dof");
}

int srcnn_solve_4dof_scalar(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                            double p2_03_minus_p3_03, float eval_thresh, double *state4, srcnn_stream_t stream)
{
    return launch_solve4(false, rec, n, rec_cols, im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03, eval_thresh, state4, stream,
                         "srcnn_solve_4dof_scalar");
}

int srcnn_align_inputs(const float *rec, int n, int rec_cols, float *boxes, float *borders, float *poses, float *valid,
                       srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rec && boxes && borders && poses && valid && n > 0 && rec_cols >= SRCNN_REC_COLS, "bad args");
    SRCNN_LAUNCH(align_inputs_kernel, dim3(cdiv(n, 64)), dim3(64), 0, as_stream(stream), rec, n, rec_cols, boxes,
                       borders, poses, valid);
    return check_launch("srcnn_align_inputs");
}

static int launch_solve3(bool wave, float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                         double p2_03_minus_p3_03, const float *align_status, const float *best_dis, double *state,
                         srcnn_stream_t stream, const char *who)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rec && state && n > 0 && rec_cols >= SRCNN_REC_COLS, "bad args (rec_cols >= SRCNN_REC_COLS)");
    SRCNN_REQUIRE((align_status == nullptr) == (best_dis == nullptr), "align_status and best_dis come together");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    if (wave) SRCNN_LAUNCH(solve3_wave_kernel, dim3(n), dim3(64), 0, as_stream(stream), rec, n, rec_cols, c, align_status, best_dis, state);
    else SRCNN_LAUNCH(solve3_kernel, dim3(n), dim3(64), 0, as_stream(stream), rec, n, rec_cols, c, align_status, best_dis, state);
    return check_launch(who);
}

int srcnn_solve_3dof(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                     double p2_03_minus_p3_03, const float *align_status, const float *best_dis, double *state,
                     srcnn_stream_t stream)
{
    return launch_solve3(true, rec, n, rec_cols, im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03, align_status, best_dis, state,
                         stream, "srcnn_solve_3dof");
}

int srcnn_solve_3dof_scalar(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                            double p2_03_minus_p3_03, const float *align_status, const float *best_dis, double *state,
                            srcnn_stream_t stream)
{
    return launch_solve3(false, rec, n, rec_cols, im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03, align_status, best_dis, state,
                         stream, "srcnn_solve_3dof_scalar");
}

// ---- the same solvers for host callers (no GPU involved): box_estimator.solve_* signatures flattened
int srcnn_solve_4dof_host(int im_h, int im_w, double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03, double alpha,
                          const double *dim3, const double *box_left4, const double *box_right4, const double *kpts5,
                          double *state4, int *newton_status, int boxes_are_float32)
{
    using namespace srcnn;
    SRCNN_REQUIRE(dim3 && box_left4 && box_right4 && kpts5 && state4, "null pointer");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    return boxsolve::solve_4dof(im_h, im_w, c.f, c.cx, c.cy, c.base, alpha, dim3, box_left4, box_right4, kpts5, state4,
                                newton_status, boxes_are_float32 != 0) ? 1 : 0;
}

int srcnn_solve_3dof_host(int im_h, int im_w, double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03, double alpha,
                          const double *dim3, const double *box_left4, double disparity, const double *kpts5, double *state3,
                          double *z, int *newton_status)
{
    using namespace srcnn;
    SRCNN_REQUIRE(dim3 && box_left4 && kpts5 && state3 && z, "null pointer");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    *z = boxsolve::solve_3dof(im_h, im_w, c.f, c.cx, c.cy, c.base, alpha, dim3, box_left4, disparity, kpts5, state3,
                              newton_status);
    return SRCNN_OK;
}

// The record forms of the two solves on HOST memory: what srcnn_solve_4dof / srcnn_solve_3dof do, row for row (same
// function), with the host's libm -- i.e. bit-identical to the reference's scipy path (tests/test_solvers_cpu.py).  The
// "reference-exact" 3-D flow copies the record down, runs these, and copies it back for the dense alignment.
int srcnn_solve_4dof_records_host(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                                  double p2_03_minus_p3_03, float eval_thresh, double *state4, int threads)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rec && state4 && n > 0 && rec_cols >= SRCNN_REC_COLS, "bad args (rec_cols >= SRCNN_REC_COLS)");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    const int k = (int)rec[0] < n ? (int)rec[0] : n;
    for (int i = k > 0 ? k : 0; i < n; ++i) state4[(size_t)i * 4] = state4[(size_t)i * 4 + 1] = state4[(size_t)i * 4 + 2] = state4[(size_t)i * 4 + 3] = 0.0;
    host_rows(k, threads, [&](int i) { solve4_row(rec, n, rec_cols, c, eval_thresh, state4, i); });
    return SRCNN_OK;
}

int srcnn_solve_3dof_records_host(float *rec, int n, int rec_cols, int im_h, int im_w, double p2_00, double p2_02, double p2_12,
                                  double p2_03_minus_p3_03, const float *align_status, const float *best_dis, double *state,
                                  int threads)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rec && state && n > 0 && rec_cols >= SRCNN_REC_COLS, "bad args (rec_cols >= SRCNN_REC_COLS)");
    SRCNN_REQUIRE((align_status == nullptr) == (best_dis == nullptr), "align_status and best_dis come together");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    const int k = (int)rec[0] < n ? (int)rec[0] : n;
    host_rows(k, threads, [&](int i) { solve3_row(rec, n, rec_cols, c, align_status, best_dis, state, i); });
    return SRCNN_OK;
}

// cost and the reference's gradient at one point (tests: against the reference's own closures)
int srcnn_solver_evaluate_host(int im_h, int im_w, double p2_00, double p2_02, double p2_12, double p2_03_minus_p3_03,
                               double alpha, const double *dim3, const double *box_left4, const double *box_right4_or_null,
                               const double *kpts5, const double *xyzt, double *cost, double *grad4)
{
    using namespace srcnn;
    SRCNN_REQUIRE(dim3 && box_left4 && kpts5 && xyzt && cost && grad4, "null pointer");
    const Calib c = make_calib(im_h, im_w, p2_00, p2_02, p2_12, p2_03_minus_p3_03);
    boxsolve::Problem t;
    boxsolve::setup(t, im_h, im_w, c.f, c.cx, c.cy, c.base, alpha, dim3, box_left4, box_right4_or_null, kpts5);
    *cost = boxsolve::evaluate(t, xyzt[0], xyzt[1], xyzt[2], xyzt[3], true, grad4);
    return SRCNN_OK;
}

}  // extern "C"
