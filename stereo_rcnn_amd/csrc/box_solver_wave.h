// The 3-D box solvers with ONE WAVEFRONT PER DETECTION (device only).
//
// box_solver.h restates scipy's Newton-CG for one detection as scalar double code; nearly all of its time is spent evaluating
// the cost / the reference's gradient (box_estimator.py:253-372 / :463-538): two fp64 trig calls and EIGHT independent
// re-projection residuals with four or five fp64 divisions each, evaluated ~10 times per Newton iteration (one per CG step
// for the finite-difference Hessian-vector product, two per line-search trial).  Here the eight residuals of one evaluation
// sit on eight lanes:
//     lane k & 7:  0 ul   1 ur   2 uk   3 ul_r   4 ur_r   5 vb   6 vt   7 alpha
// every lane carries the whole optimiser state (x, the CG vectors, the line-search brackets) redundantly, so the control flow
// of newton_cg<N> stays wave-uniform and needs no broadcast; only the residual arithmetic differs per lane, and the five
// sums (cost, g_x, g_y, g_z, g_theta) are taken over the residual lanes IN LANE ORDER (group_prefix_sums), which is the order
// the scalar code accumulates them in -- so this form is bit-identical to the scalar device build of box_solver.h
// (tests/test_box3d_gpu.py), and differs from the host build exactly where that one does (ocml cos / sin / atan2, sq()).
//
// The residual families are brought to one form so that lanes 0..6 share one instruction stream:
//     num = (b - shift) [+ ct * nw + st * nl]        b = x (u residuals) or y (v residuals); shift = 0, baseline or h
//     den = (z - st * dw) + ct * dl                  (dw, dl) = the vertex; NEGATED for vt, whose den is z + st * bw - ct * bl
//     res = scale * (num / den - obs)
// Negating both factors of a product, or writing a + (-b) for a - b, is exact in IEEE arithmetic, so each lane computes the very
// doubles the scalar code computes for its residual.
#pragma once
#include "box_solver.h"

namespace srcnn {
namespace boxsolve {

struct WaveProblem {
    double shift, nw, nl, dw, dl, scale, obs, alpha, z_fixed;
    int kind;        // this lane's residual: 0 = u (ul, ur, uk, ul_r, ur_r), 1 = v (vb, vt), 2 = alpha
    int group;       // lane >> 3: the quantity this lane carries through the sums (0..3 gradient entries, 4 cost)
    bool act;        // false: the reference zeroes this residual
    bool first;      // lane & 7 == 0
};

__device__ inline WaveProblem make_wave(const Problem &t, int lane)
{
    WaveProblem w;
    const int k = lane & 7;
    w.group = lane >> 3;
    w.first = k == 0;
    w.alpha = t.alpha;
    w.z_fixed = t.z_fixed;
    w.shift = 0.0;
    w.scale = 1.0;
    w.nw = w.nl = 0.0;
    w.kind = k < 5 ? 0 : (k < 7 ? 1 : 2);
    switch (k) {
    case 0: w.act = t.act[0]; w.obs = t.obs[0]; w.dw = t.vw[0]; w.dl = t.vl[0]; break;
    case 1: w.act = t.act[1]; w.obs = t.obs[1]; w.dw = t.vw[1]; w.dl = t.vl[1]; break;
    case 2: w.act = t.act[2]; w.obs = t.obs[2]; w.dw = t.vw[2]; w.dl = t.vl[2]; w.scale = 2.0; break;      // res_uk = 2 * res_uk (:264)
    case 3: w.act = t.act[3]; w.obs = t.obs[3]; w.dw = t.vw[0]; w.dl = t.vl[0]; w.shift = t.bl; break;
    case 4: w.act = t.act[4]; w.obs = t.obs[4]; w.dw = t.vw[1]; w.dl = t.vl[1]; w.shift = t.bl; break;
    case 5: w.act = t.act[5]; w.obs = t.obs[5]; w.dw = t.vw[3]; w.dl = t.vl[3]; break;
    case 6: w.act = t.act[6]; w.obs = t.obs[6]; w.dw = -t.vw[3]; w.dl = -t.vl[3]; w.shift = t.h; break;
    default: w.act = t.act[7]; w.obs = 0.0; w.dw = w.dl = 0.0; break;
    }
    if (w.kind == 0) { w.nw = w.dw; w.nl = w.dl; }
    return w;
}

// Ordered sums over the eight residual lanes, several quantities at once.  The wavefront is eight groups of eight lanes; lane l
// evaluates residual l & 7 (all groups redundantly) and group l >> 3 carries ONE quantity's contributions: v = that quantity of
// this lane's residual.  Seven steps of
//     s[l] = (l & 7 ? s[l - 1] : 0.0) + v[l]          (row_shr:1 DPP moves; s starts as 0.0 + v)
// leave 0.0 + v0 + v1 + ... + vk, added in that order, in lane k of every group from step k on (lane 0 holds 0.0 + v0 throughout,
// lane k at step j >= k adds v[k] to lane k - 1's finished prefix): the scalar code's `sum = 0.0; sum += ...` for every
// quantity in 7 x 5 instructions instead of 8 x 3 per quantity through v_readlane.  Lane 7 of group q then holds quantity q.
__device__ inline double group_prefix_sums(double v, bool first)
{
    double s = 0.0 + v;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(s), 0x111, 0xf, 0xf, true);      // row_shr:1, zero fill
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(s), 0x111, 0xf, 0xf, true);
        const double t = first ? 0.0 : __hiloint2double(hi, lo);
        s = t + v;
    }
    return s;
}

__device__ inline double group_total(double s, int group)          // the finished sum of `group`, the same value in every lane
{
    const int l = group * 8 + 7;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), l), __builtin_amdgcn_readlane(__double2loint(s), l));
}

// evaluate() of box_solver.h, one residual per lane.  g: (x, y, z, theta), as there.
template <bool GRAD>
__device__ inline double evaluate_wave(const WaveProblem &t, double x, double y, double z, double theta, double *g)
{
    double st, ct;
    sincos(theta, &st, &ct);          // ocml: one argument reduction, the doubles sin() and cos() return
    double c = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
    if (t.act) {
        if (t.kind < 2) {
            double num = (t.kind == 0 ? x : y) - t.shift;
            if (t.kind == 0) num = num + ct * t.nw + st * t.nl;
            const double den = z - st * t.dw + ct * t.dl;
            const double res = t.scale * (num / den - t.obs);
            c = sq(res);
            if (GRAD) {
                const double gq = 2.0 * res / den;
                c2 = -2.0 * res * num / sq(den);
                if (t.kind == 0) {
                    c0 = gq;
                    c3 = 2.0 * res * ((t.dl * ct - t.dw * st) / den + (t.dw * ct + t.dl * st) * num / sq(den));
                } else {
                    c1 = gq;
                    c3 = 2.0 * res * (num * (t.dw * ct + t.dl * st)) / sq(den);
                }
            }
        } else {
            const double res = theta - kPi / 2 + atan2(-x, z) - t.alpha;
            c = sq(res);
            if (GRAD) {
                const double r = -x / z;
                const double q = 1.0 + sq(r);
                c0 = 2.0 * res / q * (-1.0 / z);
                c2 = 2.0 * res / q * (x / (z * z));
                c3 = 2.0 * res;
            }
        }
    }
    if (GRAD) {
        const int q = t.group;
        const double s = group_prefix_sums(q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : (q == 3 ? c3 : c))), t.first);
        g[0] = group_total(s, 0);
        g[1] = group_total(s, 1);
        g[2] = group_total(s, 2);
        g[3] = group_total(s, 3);
        return group_total(s, 4);
    }
    return group_total(group_prefix_sums(c, t.first), 0);
}

template <int N>
__device__ inline double fun(const WaveProblem &t, const double *s)
{
    double g[4];
    return N == 4 ? evaluate_wave<false>(t, s[0], s[1], s[2], s[3], g) : evaluate_wave<false>(t, s[0], s[1], t.z_fixed, s[2], g);
}

template <int N>
__device__ inline void grad(const WaveProblem &t, const double *s, double *out)
{
    double g[4];
    if (N == 4) {
        evaluate_wave<true>(t, s[0], s[1], s[2], s[3], g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[2]; out[3] = g[3];
    } else {
        evaluate_wave<true>(t, s[0], s[1], t.z_fixed, s[2], g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[3];
    }
}

template <int N>
__device__ inline double fun_grad(const WaveProblem &t, const double *s, double *out)
{
    double g[4], c;
    if (N == 4) {
        c = evaluate_wave<true>(t, s[0], s[1], s[2], s[3], g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[2]; out[3] = g[3];
    } else {
        c = evaluate_wave<true>(t, s[0], s[1], t.z_fixed, s[2], g);
        out[0] = g[0]; out[1] = g[1]; out[2] = g[3];
    }
    return c;
}

// solve_4dof / solve_3dof of box_solver.h for a whole wavefront: every lane passes the same arguments and gets the same results.
__device__ inline int solve_4dof_wave(int im_h, int im_w, double f, double cx, double cy, double base, double alpha, const double *dim,
                                      const double *box_left, const double *box_right, const double *kpts, double *state,
                                      bool boxes_f32, int lane)
{
    Problem t;
    if (!prepare_4dof(t, im_h, im_w, f, cx, cy, base, alpha, dim, box_left, box_right, kpts, state, boxes_f32)) return 0;
    const WaveProblem w = make_wave(t, lane);
    newton_cg<4>(w, state);
    return state[2] > 100 ? 0 : 1;
}

__device__ inline double solve_3dof_wave(int im_h, int im_w, double f, double cx, double cy, double base, double alpha,
                                         const double *dim, const double *box_left, double disparity, const double *kpts,
                                         double *state, int lane)
{
    Problem t;
    const double z = prepare_3dof(t, im_h, im_w, f, cx, cy, base, alpha, dim, box_left, disparity, kpts, state);
    const WaveProblem w = make_wave(t, lane);
    newton_cg<3>(w, state);
    return z;
}

}  // namespace boxsolve
}  // namespace srcnn
