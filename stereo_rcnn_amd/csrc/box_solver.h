// 3-D box solvers (A14 / A17) as plain double-precision scalar code that compiles for the host AND the device.
//
// Replaces, for one detection, what the reference does with scipy on the host:
//   solve_x_y_z_theta_from_kpt  lib/model/utils/box_estimator.py:169-385   (4-DoF: x, y, z, theta)
//   solve_x_y_theta_from_kpt    lib/model/utils/box_estimator.py:387-545   (3-DoF: x, y, theta; z from the aligned disparity)
// Both hand a sum of squared re-projection residuals and a hand-written "gradient" (which is NOT the gradient of that cost:
// the doubled keypoint residual, :264, is differentiated without its factor 2, :311-316) to
// scipy.optimize.minimize(method='Newton-CG').  The point the optimiser stops at is therefore defined by the optimiser's
// own control flow, not by a stationarity condition, so this file restates that optimiser step by
// step -- scipy 1.15 (the version in this image; the reference leaves scipy unpinned, requirements.txt:4):
//   _minimize_newtoncg (optimize/_optimize.py): CG inner loop on finite-difference Hessian-vector products
//       (approx_fhess_p, epsilon = sqrt(DBL_EPSILON)), xtol = n * 1e-5 on the l1 norm of the update, maxiter = 200 n,
//       cg_maxiter = 20 n, curvature tests with 3 * DBL_EPSILON;
//   _line_search_wolfe12: line_search_wolfe1 (MINPACK-2 dcsrch / dcstep, optimize/_dcsrch.py; c1 = 1e-4, c2 = 0.9,
//       amax = 50, amin = 1e-8, xtol = 1e-14, <= 100 iterations) and, when that fails, line_search_wolfe2
//       (scalar_search_wolfe2 + _zoom + _cubicmin / _quadmin, optimize/_linesearch.py; <= 10 + 10 iterations);
//   the observation set-up (viewpoint tables, truncation switches, early-outs) of box_estimator.py:15-167,186-260,402-470.
// Every quantity is a double, as in the reference (under torch 0.3 an indexed tensor element is a Python float).
// What cannot be reproduced bit for bit on ANY other machine is the libm underneath (numpy's cos / sin, OpenBLAS' ddot
// summation order); the end point is chaotic in those last bits for ill-posed boxes, for scipy itself too
// (tests/test_solvers_cpu.py quantifies both).
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define SRCNN_HD __host__ __device__ inline
#else
#define SRCNN_HD inline
#endif

namespace srcnn {
namespace boxsolve {

constexpr double kPi = 3.141592653589793;     // math.pi

// ------------------------------------------------------------------------------------------------ observation set-up
// box_estimator.py:15-41
SRCNN_HD int bb2viewpoint(double alpha)
{
    alpha = alpha * 180.0 / kPi;
    if (alpha > 360) alpha = alpha - 360;
    else if (alpha < -360) alpha = alpha + 360;
    const double thr = 4.0;
    if (alpha >= -90.0 - thr && alpha <= -90.0 + thr) return 0;
    if (alpha >= -180.0 + thr && alpha <= -90.0 - thr) return 1;
    if (alpha >= 180.0 - thr || alpha <= -180.0 + thr) return 2;
    if (alpha >= 90.0 + thr && alpha <= 180.0 - thr) return 3;
    if (alpha >= 90.0 - thr && alpha <= 90.0 + thr) return 4;
    if (alpha >= 0.0 + thr && alpha <= 90.0 - thr) return 5;
    if (alpha >= 0.0 - thr && alpha <= 0.0 + thr) return 6;
    if (alpha >= -90.0 + thr && alpha <= 0.0 - thr) return 7;
    return -1;
}

struct Problem {
    double h, f, bl, z_fixed, alpha;
    double obs[7];        // ul, ur, uk, ul_r, ur_r, vb, vt  (normalised image plane)
    double vw[4], vl[4];  // object-frame (w, l) half extents of the left / right / keypoint / bottom vertex
    bool act[8];          // ul, ur, uk, ul_r, ur_r, vb, vt, alpha  -- residuals the reference zeroes are inactive
    bool truncation;
};

// (left, right, bottom) vertex signs (w, l) for view points 0..7 (box_estimator.py:92-122; -1 falls into the last branch)
SRCNN_HD void side_vertices(int view_point, double w, double l, double *vw, double *vl)
{
    const int sw[8][3] = {{-1, 1, 1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}, {1, -1, -1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
    const int sl[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, -1, -1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}, {-1, 1, 1}, {-1, 1, -1}};
    const int v = (view_point >= 0 && view_point <= 7) ? view_point : 7;
    vw[0] = sw[v][0] * w / 2; vl[0] = sl[v][0] * l / 2;      // left
    vw[1] = sw[v][1] * w / 2; vl[1] = sl[v][1] * l / 2;      // right
    vw[3] = sw[v][2] * w / 2; vl[3] = sl[v][2] * l / 2;      // bottom
}

// box_estimator.py:150-167
// box_f32: `box` is a numpy float32 row and kpt_pos a Python float -- numpy (NEP 50, as executed where the goldens were made)
// keeps the whole quotient in float32
SRCNN_HD double kpt2alpha(double kpt_pos, int kpt_type, double box0, double box2, bool box_f32 = false)
{
    double ratio = (kpt_pos - box0) / (box2 - box0);
    if (box_f32) ratio = (double)(((float)kpt_pos - (float)box0) / ((float)box2 - (float)box0));
    ratio = ratio < 1 ? ratio : 1;         // min(1, .)
    ratio = ratio > -1 ? ratio : -1;       // max(., -1)
    const double base = kpt_type == 0 ? -kPi / 2 : (kpt_type == 1 ? kPi : (kpt_type == 2 ? kPi / 2 : 0.0));
    return base - asin(ratio);
}

// Shared by both solvers (box_estimator.py:188-260 and :402-470).  box_right == nullptr: the 3-DoF problem.
// im_h / im_w: original image size; p2_00 = f, p2_02 = cx, p2_12 = cy, base = (P2[0,3] - P3[0,3]) / f.
SRCNN_HD void setup(Problem &t, int im_h, int im_w, double f, double cx, double cy, double base, double alpha, const double *dim,
                    const double *box_left, const double *box_right, const double *kpts, bool boxes_f32 = false)
{
    const double TB = 10;     // truncate_border
    t.h = dim[1];
    const double w = dim[0], l = dim[2];
    const double ul = box_left[0], vt = box_left[1], ur = box_left[2], vb = box_left[3];
    t.f = f;
    t.bl = base;
    const double kpt_pos = kpts[0];
    int kpt_type = (int)kpts[1];              // int(): truncation toward zero
    if (kpt_type < 0) kpt_type += 4;          // Python's negative index into the 4-entry table
    kpt_type = kpt_type < 0 ? 0 : (kpt_type > 3 ? 3 : kpt_type);
    t.obs[0] = (ul - cx) / f;
    t.obs[1] = (ur - cx) / f;
    t.obs[2] = (kpt_pos - cx) / f;
    t.obs[5] = (vb - cy) / f;
    t.obs[6] = (vt - cy) / f;
    t.truncation = ul < 2.0 * TB || ur > im_w - 2.0 * TB;
    if (!t.truncation) alpha = kpt2alpha(kpt_pos, kpt_type, box_left[0], box_left[2], boxes_f32);
    t.alpha = alpha;
    side_vertices(bb2viewpoint(alpha), w, l, t.vw, t.vl);
    const int kw[4] = {-1, -1, 1, 1}, kl[4] = {-1, 1, 1, -1};       // box_estimator.py:138-146
    t.vw[2] = kw[kpt_type] * w / 2;
    t.vl[2] = kl[kpt_type] * l / 2;
    t.act[0] = !(ul < 2.0 * TB);
    t.act[1] = !(ur > im_w - 2.0 * TB);
    t.act[2] = !t.truncation;
    t.act[7] = t.truncation;
    t.act[6] = !(vt < TB);
    t.act[5] = !(vb > im_h - TB);
    t.act[3] = t.act[4] = false;
    t.obs[3] = t.obs[4] = 0;
    if (box_right) {
        const double ul_r = box_right[0], ur_r = box_right[2];
        t.obs[3] = (ul_r - cx) / f;
        t.obs[4] = (ur_r - cx) / f;
        t.act[3] = t.truncation && !(ul_r < 2.0 * TB);
        t.act[4] = t.truncation && !(ur_r > im_w - 2.0 * TB);
    }
    t.z_fixed = 0;
}

// `v ** 2` of the reference (numpy float64 scalars) is libm's pow(v, 2.0), which is NOT always the correctly rounded v * v
// (glibc: last bit differs in ~0.06 % of the calls) -- and one last bit moves a chaotic Newton-CG end point.  The host build
// therefore calls the same pow (through a volatile exponent: the compiler would fold pow(v, 2.0) into v * v); the device has
// no glibc and squares exactly, which is one of the two reasons its end points differ from the host's (the other: ocml cos/sin).
SRCNN_HD double sq(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return v * v;
#else
    volatile double two = 2.0;
    return pow(v, two);
#endif
}

// Cost (sum of squares, box_estimator.py:253-276 / :463-480) and the REFERENCE's gradient (:278-372 / :482-538) at
// (x, y, z, theta).  g has 4 entries (x, y, z, theta).
SRCNN_HD double evaluate(const Problem &t, double x, double y, double z, double theta, bool want_grad, double *g)
{
    const double ct = cos(theta), st = sin(theta);
    double cost = 0.0;
    if (want_grad) g[0] = g[1] = g[2] = g[3] = 0.0;
    // ul, ur, uk, ul_r, ur_r  ->  (vertex index, x shift, residual scale)
    const int vidx[5] = {0, 1, 2, 0, 1};
    for (int k = 0; k < 5; ++k) {
        if (!t.act[k]) continue;
        const double shift = k >= 3 ? t.bl : 0.0, scale = k == 2 ? 2.0 : 1.0;      // res_uk = 2 * res_uk (:264)
        const double vw = t.vw[vidx[k]], vl = t.vl[vidx[k]];
        const double num = x - shift + ct * vw + st * vl;
        const double den = z - st * vw + ct * vl;
        const double res = scale * (num / den - t.obs[k]);
        cost += sq(res);
        if (want_grad) {
            g[0] += 2.0 * res / den;
            g[2] += -2.0 * res * num / sq(den);
            g[3] += 2.0 * res * ((vl * ct - vw * st) / den + (vw * ct + vl * st) * num / sq(den));
        }
    }
    const double bw = t.vw[3], bl_ = t.vl[3];
    if (t.act[5]) {
        const double den = z - st * bw + ct * bl_;
        const double res = y / den - t.obs[5];
        cost += sq(res);
        if (want_grad) {
            g[1] += 2.0 * res / den;
            g[2] += -2.0 * res * y / sq(den);
            g[3] += 2.0 * res * (y * (bw * ct + bl_ * st)) / sq(den);
        }
    }
    if (t.act[6]) {
        const double den = z + st * bw - ct * bl_;
        const double res = (y - t.h) / den - t.obs[6];
        cost += sq(res);
        if (want_grad) {
            g[1] += 2.0 * res / den;
            g[2] += 2.0 * res * (t.h - y) / sq(den);
            g[3] += 2.0 * res * ((t.h - y) * (bw * ct + bl_ * st)) / sq(den);
        }
    }
    if (t.act[7]) {
        const double res = theta - kPi / 2 + atan2(-x, z) - t.alpha;
        cost += sq(res);
        if (want_grad) {
            const double r = -x / z;
            const double q = 1.0 + sq(r);
            g[0] += 2.0 * res / q * (-1.0 / z);
            g[2] += 2.0 * res / q * (x / (z * z));
            g[3] += 2.0 * res;
        }
    }
    return cost;
}

// N = 4: state (x, y, z, theta).  N = 3: state (x, y, theta), z = t.z_fixed, gradient entries (0, 1, 3).
template <int N>
SRCNN_HD double fun(const Problem &t, const double *s)
{
    double g[4];
    return N == 4 ? evaluate(t, s[0], s[1], s[2], s[3], false, g) : evaluate(t, s[0], s[1], t.z_fixed, s[2], false, g);
}

template <int N>
SRCNN_HD void grad(const Problem &t, const double *s, double *out)
{
    double g[4];
    if (N == 4) {
        evaluate(t, s[0], s[1], s[2], s[3], true, g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[2]; out[3] = g[3];
    } else {
        evaluate(t, s[0], s[1], t.z_fixed, s[2], true, g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[3];
    }
}

// cost AND gradient at one point from one pass over the residuals: the doubles fun<N> and grad<N> return there (evaluate()
// computes the cost the same way whether or not it is asked for the gradient).  The optimiser below asks for both wherever
// scipy calls phi(stp) and derphi(stp) back to back.
template <int N>
SRCNN_HD double fun_grad(const Problem &t, const double *s, double *out)
{
    double g[4], c;
    if (N == 4) {
        c = evaluate(t, s[0], s[1], s[2], s[3], true, g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[2]; out[3] = g[3];
    } else {
        c = evaluate(t, s[0], s[1], t.z_fixed, s[2], true, g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[3];
    }
    return c;
}

// ------------------------------------------------------------------------------------------------ small vector helpers
// np.dot of two short float64 vectors = cblas_ddot of the OpenBLAS bundled with numpy: on every x86 core with FMA (Haswell /
// SkylakeX / Zen kernels) the n < 32 tail is the scalar loop `dot += y[i] * x[i]` compiled to fused multiply-adds
// (measured here: equal to the fma chain on 3000 / 3000 random vectors, to the unfused sum on 61 %).  fma() rounds once on
// the host and on the device alike, so this is also what makes the two builds agree.
template <int N> SRCNN_HD double dot(const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < N; ++i) s = fma(a[i], b[i], s);
    return s;
}
template <int N> SRCNN_HD double norm1(const double *a)
{
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += fabs(a[i]);
    return s;
}
SRCNN_HD double py_max3(double a, double b, double c)          // Python's max(a, b, c)
{
    double m = a;
    if (b > m) m = b;
    if (c > m) m = c;
    return m;
}
SRCNN_HD double py_min(double a, double b) { return b < a ? b : a; }      // min(a, b)
SRCNN_HD double py_max(double a, double b) { return b > a ? b : a; }      // max(a, b)
SRCNN_HD double np_clip(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }   // NaN propagates like np.clip
SRCNN_HD bool fin(double v) { return v - v == 0.0; }                          // np.isfinite (false for inf and NaN)
SRCNN_HD double np_sign(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : (v == 0 ? 0.0 : v)); }

// line phi(s) = f(xk + s * pk), derphi(s) = <grad f(xk + s * pk), pk>.  P: the problem type -- `Problem` (one thread evaluates all
// eight residuals) or box_solver_wave.h's WaveProblem (one residual per lane); fun<N> / grad<N> are overloaded on it.
template <int N, class P>
struct Line {
    const P *t;
    const double *xk, *pk;
    double gval[N];          // gradient at the last derphi() point (line_search_wolfe1/2 hand it back as gfkp1)
    SRCNN_HD double phi(double s) const
    {
        double x[N];
        for (int i = 0; i < N; ++i) x[i] = xk[i] + s * pk[i];
        return fun<N>(*t, x);
    }
    SRCNN_HD double derphi(double s)
    {
        double x[N];
        for (int i = 0; i < N; ++i) x[i] = xk[i] + s * pk[i];
        grad<N>(*t, x, gval);
        return dot<N>(gval, pk);
    }
    // phi(s) and derphi(s) of one point in one evaluation
    SRCNN_HD double phi_derphi(double s, double &der)
    {
        double x[N];
        for (int i = 0; i < N; ++i) x[i] = xk[i] + s * pk[i];
        const double f = fun_grad<N>(*t, x, gval);
        der = dot<N>(gval, pk);
        return f;
    }
};

// ------------------------------------------------------------------------------------------------ MINPACK-2 dcstep
struct DcState {
    double stx, fx, gx, sty, fy, gy, stmin, stmax, width, width1, finit, ginit, gtest;
    bool brackt;
    int stage;
};

SRCNN_HD void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, double fp, double dp,
                     bool &brackt, double stpmin, double stpmax)
{
    const double sgnd = np_sign(dp) * np_sign(dx);
    double stpf;
    if (fp > fx) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma *= -1;
        const double p = (gamma - dx) + theta;
        const double q = ((gamma - dx) + gamma) + dp;
        const double r = p / q;
        const double stpc = stx + r * (stp - stx);
        const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        if (fabs(stpc - stx) <= fabs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {
        const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma *= -1;
        const double p = (gamma - dp) + theta;
        const double q = ((gamma - dp) + gamma) + dx;
        const double r = p / q;
        const double stpc = stp + r * (stx - stp);
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = true;
    } else if (fabs(dp) < fabs(dx)) {
        const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        const double rad = (theta / s) * (theta / s) - (dx / s) * (dp / s);
        double gamma = s * sqrt(rad > 0 ? rad : 0.0);              // max(0, .)
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta;
        const double q = (gamma + (dx - dp)) + gamma;
        const double r = p / q;
        double stpc;
        if (r < 0 && gamma != 0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            if (stp > stx) stpf = py_min(stp + 0.66 * (sty - stp), stpf);
            else stpf = py_max(stp + 0.66 * (sty - stp), stpf);
        } else {
            if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            stpf = np_clip(stpf, stpmin, stpmax);
        }
    } else {
        if (brackt) {
            const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            const double s = py_max3(fabs(theta), fabs(dy), fabs(dp));
            double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            const double p = (gamma - dp) + theta;
            const double q = ((gamma - dp) + gamma) + dy;
            const double r = p / q;
            stpf = stp + r * (sty - stp);
        } else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
    if (fp > fx) {
        sty = stp; fy = fp; dy = dp;
    } else {
        if (sgnd < 0) { sty = stx; fy = fx; dy = dx; }
        stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
}

// one DCSRCH._iterate call after START.  Returns 0 = 'FG' (evaluate at stp, call again), 1 = CONVERGENCE, 2 = WARNING.
SRCNN_HD int dcsrch_iterate(DcState &d, double &stp, double f, double g, double ftol, double gtol, double xtol, double stpmin,
                            double stpmax)
{
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    const double ftest = d.finit + stp * d.gtest;
    if (d.stage == 1 && f <= ftest && g >= 0) d.stage = 2;
    int task = 0;
    if (d.brackt && (stp <= d.stmin || stp >= d.stmax)) task = 2;
    if (d.brackt && d.stmax - d.stmin <= xtol * d.stmax) task = 2;
    if (stp == stpmax && f <= ftest && g <= d.gtest) task = 2;
    if (stp == stpmin && (f > ftest || g >= d.gtest)) task = 2;
    if (f <= ftest && fabs(g) <= gtol * -d.ginit) task = 1;
    if (task) return task;
    if (d.stage == 1 && f <= d.fx && f > ftest) {
        const double fm = f - stp * d.gtest;
        double fxm = d.fx - d.stx * d.gtest, fym = d.fy - d.sty * d.gtest;
        const double gm = g - d.gtest;
        double gxm = d.gx - d.gtest, gym = d.gy - d.gtest;
        dcstep(d.stx, fxm, gxm, d.sty, fym, gym, stp, fm, gm, d.brackt, d.stmin, d.stmax);
        d.fx = fxm + d.stx * d.gtest;
        d.fy = fym + d.sty * d.gtest;
        d.gx = gxm + d.gtest;
        d.gy = gym + d.gtest;
    } else {
        dcstep(d.stx, d.fx, d.gx, d.sty, d.fy, d.gy, stp, f, g, d.brackt, d.stmin, d.stmax);
    }
    if (d.brackt) {
        if (fabs(d.sty - d.stx) >= p66 * d.width1) stp = d.stx + p5 * (d.sty - d.stx);
        d.width1 = d.width;
        d.width = fabs(d.sty - d.stx);
    }
    if (d.brackt) {
        d.stmin = py_min(d.stx, d.sty);
        d.stmax = py_max(d.stx, d.sty);
    } else {
        d.stmin = stp + xtrapl * (stp - d.stx);
        d.stmax = stp + xtrapu * (stp - d.stx);
    }
    stp = np_clip(stp, stpmin, stpmax);
    if ((d.brackt && (stp <= d.stmin || stp >= d.stmax)) || (d.brackt && d.stmax - d.stmin <= xtol * d.stmax)) stp = d.stx;
    return 0;
}

// scalar_search_wolfe1 (optimize/_linesearch.py) = DCSRCH.__call__.  Returns true and (stp, phi1) on success.
template <int N, class P>
SRCNN_HD bool search_wolfe1(Line<N, P> &ln, double phi0, bool have_old, double old_phi0, double derphi0, double &stp_out,
                            double &phi_out)
{
    const double c1 = 1e-4, c2 = 0.9, amax = 50, amin = 1e-8, xtol = 1e-14;
    double alpha1 = 1.0;
    if (have_old && derphi0 != 0) {
        alpha1 = py_min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
        if (alpha1 < 0) alpha1 = 1.0;
    }
    // task START: argument checks
    if (alpha1 < amin || alpha1 > amax || derphi0 >= 0) return false;      // 'ERROR: ...' -> stp = None
    if (!(alpha1 >= amin)) return false;                                    // NaN start step
    DcState d;
    d.brackt = false;
    d.stage = 1;
    d.finit = phi0;
    d.ginit = derphi0;
    d.gtest = c1 * d.ginit;
    d.width = amax - amin;
    d.width1 = d.width / 0.5;
    d.stx = 0.0; d.fx = d.finit; d.gx = d.ginit;
    d.sty = 0.0; d.fy = d.finit; d.gy = d.ginit;
    d.stmin = 0;
    d.stmax = alpha1 + 4.0 * alpha1;
    double stp = alpha1;
    // iteration 0 of the Python loop was the START call (returns 'FG'); 99 more may follow
    if (!fin(stp)) return false;
    double derphi1;
    double phi1 = ln.phi_derphi(stp, derphi1);
    for (int i = 1; i < 100; ++i) {
        const int task = dcsrch_iterate(d, stp, phi1, derphi1, c1, c2, xtol, amin, amax);
        if (!fin(stp)) return false;
        if (task == 0) {
            phi1 = ln.phi_derphi(stp, derphi1);
        } else {
            if (task == 2) return false;        // WARNING -> stp = None
            stp_out = stp;
            phi_out = phi1;
            return true;
        }
    }
    return false;                               // did not converge within max iterations
}

// ------------------------------------------------------------------------------------------------ wolfe2 / zoom
SRCNN_HD bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double &xmin)
{
    const double C = fpa;
    const double db = b - a, dc = c - a;
    const double denom = (db * dc) * (db * dc) * (db - dc);
    const double d00 = dc * dc, d01 = -(db * db), d10 = -(dc * dc * dc), d11 = db * db * db;
    const double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
    double A = d00 * v0 + d01 * v1, B = d10 * v0 + d11 * v1;
    if (denom == 0 || !fin(denom) || !fin(A) || !fin(B)) return false;      // np.errstate(... = 'raise')
    A /= denom;
    B /= denom;
    const double radical = B * B - 3 * A * C;
    if (!fin(A) || !fin(B) || !(radical >= 0) || A == 0) return false;
    xmin = a + (-B + sqrt(radical)) / (3 * A);
    return fin(xmin);
}

SRCNN_HD bool quadmin(double a, double fa, double fpa, double b, double fb, double &xmin)
{
    const double D = fa, C = fpa;
    const double db = b - a * 1.0;
    if (db * db == 0) return false;
    const double B = (fb - D - C * db) / (db * db);
    if (!fin(B) || 2.0 * B == 0) return false;
    xmin = a - C / (2.0 * B);
    return fin(xmin);
}

template <int N, class P>
SRCNN_HD bool zoom(Line<N, P> &ln, double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo, double phi0,
                   double derphi0, double c1, double c2, double &a_star, double &val_star)
{
    const int maxiter = 10;
    int i = 0;
    const double delta1 = 0.2, delta2 = 0.1;
    double phi_rec = phi0, a_rec = 0;
    for (;;) {
        const double dalpha = a_hi - a_lo;
        double a, b;
        if (dalpha < 0) { a = a_hi; b = a_lo; }
        else { a = a_lo; b = a_hi; }
        double a_j = 0, cchk = 0;
        bool have = false;
        if (i > 0) {
            cchk = delta1 * dalpha;
            have = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, a_j);
        }
        if (i == 0 || !have || a_j > b - cchk || a_j < a + cchk) {
            const double qchk = delta2 * dalpha;
            have = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_j);
            if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
        }
        const double phi_aj = ln.phi(a_j);
        if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
            phi_rec = phi_hi; a_rec = a_hi;
            a_hi = a_j; phi_hi = phi_aj;
        } else {
            const double derphi_aj = ln.derphi(a_j);
            if (fabs(derphi_aj) <= -c2 * derphi0) {
                a_star = a_j;
                val_star = phi_aj;
                return true;
            }
            if (derphi_aj * (a_hi - a_lo) >= 0) {
                phi_rec = phi_hi; a_rec = a_hi;
                a_hi = a_lo; phi_hi = phi_lo;
            } else {
                phi_rec = phi_lo; a_rec = a_lo;
            }
            a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
        }
        i += 1;
        if (i > maxiter) return false;
    }
}

// scalar_search_wolfe2 with amax = None, extra_condition = None, maxiter = 10.  Returns true and (alpha, phi) when the caller
// (_line_search_wolfe12) would accept the step: alpha_star is not None -- which includes the "did not converge" exit of the
// for-else branch (alpha_star = alpha1, derphi_star = None).
template <int N, class P>
SRCNN_HD bool search_wolfe2(Line<N, P> &ln, double phi0, bool have_old, double old_phi0, double derphi0, double &alpha_star,
                            double &phi_star)
{
    const double c1 = 1e-4, c2 = 0.9;
    double alpha0 = 0, alpha1 = 1.0;
    if (have_old && derphi0 != 0) alpha1 = py_min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
    if (alpha1 < 0) alpha1 = 1.0;
    double phi_a1 = ln.phi(alpha1);
    double phi_a0 = phi0, derphi_a0 = derphi0;
    for (int i = 0; i < 10; ++i) {
        if (alpha1 == 0) return false;                    // alpha_star = None
        if (phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && i > 0))
            return zoom<N, P>(ln, alpha0, alpha1, phi_a0, phi_a1, derphi_a0, phi0, derphi0, c1, c2, alpha_star, phi_star);
        const double derphi_a1 = ln.derphi(alpha1);
        if (fabs(derphi_a1) <= -c2 * derphi0) {
            alpha_star = alpha1;
            phi_star = phi_a1;
            return true;
        }
        if (derphi_a1 >= 0)
            return zoom<N, P>(ln, alpha1, alpha0, phi_a1, phi_a0, derphi_a1, phi0, derphi0, c1, c2, alpha_star, phi_star);
        const double alpha2 = 2 * alpha1;
        alpha0 = alpha1;
        alpha1 = alpha2;
        phi_a0 = phi_a1;
        phi_a1 = ln.phi(alpha1);
        derphi_a0 = derphi_a1;
    }
    alpha_star = alpha1;                                  // maxiter reached: alpha_star = alpha1, accepted by the caller
    phi_star = phi_a1;
    return true;
}

// ------------------------------------------------------------------------------------------------ Newton-CG
// _minimize_newtoncg with default options.  xk: start point in, end point out.  Returns scipy's status (0 success,
// 1 maxiter, 2 line search failed, 3 CG did not converge / NaN); the reference ignores it and takes res.x.
template <int N, class P>
SRCNN_HD int newton_cg(const P &t, double *xk, int *iterations = nullptr)
{
    const double avextol = 1e-5, epsilon = 1.4901161193847656e-08;     // sqrt(np.finfo(float).eps)
    const int maxiter = N * 200, cg_maxiter = 20 * N;
    const double xtol = N * avextol;
    double update_l1norm = DBL_MAX;
    int k = 0;
    // The gradient at the new xk is the one the MINPACK line search evaluated last (it returns the step it evaluated last, and
    // xk + alphak * pk is formed the same way in Line and below): scipy recomputes it, fprime(xk), to the same doubles; here it
    // is carried over.  After a wolfe2 search it is recomputed (its for-else exit returns a step whose gradient was not taken).
    double gfk[N];
    double old_fval = fun_grad<N>(t, xk, gfk), old_old_fval = 0;
    bool have_old = false, have_gfk = true;
    int status = 0;
    while (update_l1norm > xtol) {
        if (k >= maxiter) { status = 1; break; }
        double b[N], xsupi[N], ri[N], psupi[N];
        if (!have_gfk) grad<N>(t, xk, gfk);
        for (int i = 0; i < N; ++i) b[i] = -gfk[i];
        const double maggrad = norm1<N>(b);
        const double eta = py_min(0.5, sqrt(maggrad));
        const double termcond = eta * maggrad;
        for (int i = 0; i < N; ++i) {
            xsupi[i] = 0.0;
            ri[i] = -b[i];
            psupi[i] = -ri[i];
        }
        int i_cg = 0;
        double dri0 = dot<N>(ri, ri);
        bool broke = false;
        for (int k2 = 0; k2 < cg_maxiter; ++k2) {
            if (norm1<N>(ri) <= termcond) { broke = true; break; }
            double xp[N], f2[N], Ap[N];
            for (int i = 0; i < N; ++i) xp[i] = xk[i] + epsilon * psupi[i];
            grad<N>(t, xp, f2);
            for (int i = 0; i < N; ++i) Ap[i] = (f2[i] - gfk[i]) / epsilon;
            const double curv = dot<N>(psupi, Ap);
            if (0 <= curv && curv <= 3 * DBL_EPSILON) { broke = true; break; }
            else if (curv < 0) {
                if (i_cg > 0) { broke = true; break; }
                const double s = dri0 / (-curv);
                for (int i = 0; i < N; ++i) xsupi[i] = s * b[i];
                broke = true;
                break;
            }
            const double alphai = dri0 / curv;
            for (int i = 0; i < N; ++i) xsupi[i] += alphai * psupi[i];
            for (int i = 0; i < N; ++i) ri[i] += alphai * Ap[i];
            const double dri1 = dot<N>(ri, ri);
            const double betai = dri1 / dri0;
            for (int i = 0; i < N; ++i) psupi[i] = -ri[i] + betai * psupi[i];
            i_cg += 1;
            dri0 = dri1;
        }
        if (!broke) { status = 3; break; }             // "CG iterations didn't converge"
        const double *pk = xsupi;
        Line<N, P> ln;
        ln.t = &t;
        ln.xk = xk;
        ln.pk = pk;
        const double derphi0 = dot<N>(gfk, pk);
        double alphak = 0, new_fval = 0;
        bool ok = search_wolfe1<N, P>(ln, old_fval, have_old, old_old_fval, derphi0, alphak, new_fval);
        have_gfk = ok;
        if (!ok) ok = search_wolfe2<N, P>(ln, old_fval, have_old, old_old_fval, derphi0, alphak, new_fval);
        if (!ok) { status = 2; break; }                // _LineSearchError: "precision loss"
        if (have_gfk)
            for (int i = 0; i < N; ++i) gfk[i] = ln.gval[i];
        old_old_fval = old_fval;
        have_old = true;
        old_fval = new_fval;
        double update[N];
        for (int i = 0; i < N; ++i) update[i] = alphak * pk[i];
        for (int i = 0; i < N; ++i) xk[i] += update[i];
        k += 1;
        update_l1norm = norm1<N>(update);
    }
    if (status == 0 && (old_fval != old_fval || update_l1norm != update_l1norm)) status = 3;
    if (iterations) *iterations = k;
    return status;
}

// ------------------------------------------------------------------------------------------------ the two entry points
// solve_x_y_z_theta_from_kpt (box_estimator.py:169-385).  Returns status (0 failed, 1 normal); state = (x, y, z, theta).
// boxes_f32: box_left / box_right hold float32 values and are numpy float32 arrays in the caller being mirrored (demo.py:284-285
// passes `.cpu().numpy()` rows): numpy then evaluates the box-size tests (:186), the start disparity (:374) and the keypoint
// ratio of kpt2alpha (:160) in FLOAT32 arithmetic, everything else is promoted to double by the float64 calibration entries.
// kpts is a torch tensor row there (Python floats) and dim only enters through products with doubles.
// prepare_4dof: everything before the optimiser (early-outs :186-187, observation set-up, start point :374-378).  false = the
// reference returns (zeros, status 0) without solving.
SRCNN_HD bool prepare_4dof(Problem &t, int im_h, int im_w, double f, double cx, double cy, double base, double alpha, const double *dim,
                           const double *box_left, const double *box_right, const double *kpts, double *state, bool boxes_f32)
{
    state[0] = state[1] = state[2] = state[3] = 0;
    double bw = box_left[2] - box_left[0], bh = box_left[3] - box_left[1];
    double disparity = (box_left[0] + box_left[2]) / 2 - (box_right[0] + box_right[2]) / 2;
    if (boxes_f32) {
        const float l0 = (float)box_left[0], l1 = (float)box_left[1], l2 = (float)box_left[2], l3 = (float)box_left[3];
        const float r0 = (float)box_right[0], r2 = (float)box_right[2];
        bw = (double)(l2 - l0);
        bh = (double)(l3 - l1);
        const float sl = (l0 + l2) / 2, sr = (r0 + r2) / 2;
        disparity = (double)(sl - sr);
    }
    if (kpts[4] - kpts[3] < 3 || bw < 10 || bh < 10) return false;                                               // :186-187
    setup(t, im_h, im_w, f, cx, cy, base, alpha, dim, box_left, box_right, kpts, boxes_f32);
    const double init_z = t.f * t.bl / disparity;
    const double init_x = init_z * (t.obs[0] + t.obs[1]) / 2.0;
    const double init_y = init_z * (t.obs[5] + t.obs[6]) / 2.0 + t.h / 2.0;
    const double init_theta = t.alpha + kPi / 2 - atan2(-init_x, init_z);
    state[0] = init_x; state[1] = init_y; state[2] = init_z; state[3] = init_theta;
    return true;
}

SRCNN_HD int solve_4dof(int im_h, int im_w, double f, double cx, double cy, double base, double alpha, const double *dim,
                        const double *box_left, const double *box_right, const double *kpts, double *state, int *newton_status,
                        bool boxes_f32 = false)
{
    if (newton_status) *newton_status = -1;
    Problem t;
    if (!prepare_4dof(t, im_h, im_w, f, cx, cy, base, alpha, dim, box_left, box_right, kpts, state, boxes_f32)) return 0;
    const int st = newton_cg<4>(t, state);
    if (newton_status) *newton_status = st;
    return state[2] > 100 ? 0 : 1;                                                                                 // :383-385
}

// solve_x_y_theta_from_kpt (box_estimator.py:387-545).  state = (x, y, theta); returns z.
SRCNN_HD double prepare_3dof(Problem &t, int im_h, int im_w, double f, double cx, double cy, double base, double alpha,
                             const double *dim, const double *box_left, double disparity, const double *kpts, double *state)
{
    setup(t, im_h, im_w, f, cx, cy, base, alpha, dim, box_left, nullptr, kpts);
    const double z = t.f * t.bl / disparity;
    t.z_fixed = z;
    const double init_x = z * (t.obs[0] + t.obs[1]) / 2.0;
    const double init_y = z * (t.obs[5] + t.obs[6]) / 2.0 + t.h / 2.0;
    const double init_theta = t.alpha + kPi / 2 - atan2(-init_x, z);
    state[0] = init_x; state[1] = init_y; state[2] = init_theta;
    return z;
}

SRCNN_HD double solve_3dof(int im_h, int im_w, double f, double cx, double cy, double base, double alpha, const double *dim,
                           const double *box_left, double disparity, const double *kpts, double *state, int *newton_status)
{
    Problem t;
    const double z = prepare_3dof(t, im_h, im_w, f, cx, cy, base, alpha, dim, box_left, disparity, kpts, state);
    const int st = newton_cg<3>(t, state);
    if (newton_status) *newton_status = st;
    return z;
}

}  // namespace boxsolve
}  // namespace srcnn
