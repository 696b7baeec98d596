// wm_gicp_quad.hpp -- GICP's objective as SUFFICIENT STATISTICS: one pass over the pairs per outer iteration,
// after which every evaluation of f and its gradient that the optimiser asks for is scalar work on 74 numbers.
//
// What PCL does (registration/impl/gicp.hpp, OptimizationFunctorWithIndices::fdf -- the reference reaches it through
// wave_matching/src/gicp.cpp:58 -> align -> estimateRigidTransformationBFGS): for every trial point x of the line
// search it applies the float transform T(x) to every source point, forms the residual r = T(x) p - q and adds
// r^T M r, M r and p (M r)^T over all pairs -- ~40 passes over the pairs per outer iteration, ~170 per registration.
// But between two correspondence searches the pairs (p, q) and their Mahalanobis matrices M are FIXED, and the
// residual is affine in the twelve entries W of the transform: with z = (p, 1),
//     r(W) = W z - q = r0 + D z,      r0 = T0 z - q,   D = W - T0     (T0: the transform the pairs were found under)
//     sum r^T M r   = C0 + sum_aj D_aj (B0_aj + G_aj)
//     G_aj = sum (M r)_a z_j = B0_aj + sum_ck D_ck A_(ac)(jk)
// with A_(ac)(jk) = sum M_ac z_j z_k (6 x 10 by symmetry), B0_aj = sum (M r0)_a z_j (12), C0 = sum r0^T M r0 and the
// pair count: 74 sums, formed ONCE per outer iteration (k_gicp_quad, wm_gicp.hip; gs_statistics, wm_gicp_small.hip).
// f, the translation gradient (G's last column) and PCL's rotation accumulator (sum p_base (M r)^T = B G^T) follow
// for any x without touching the pairs again.  The expansion is around T0, where the optimiser starts: D is small,
// nothing cancels (the constant term IS the objective at the starting point).
//
// What is kept of PCL's arithmetic: r0 is PCL's residual -- the float transform, the float subtraction -- so at
// x = x0 the value is the one PCL computes, and T(x) is PCL's float matrix (applyState).  What is not: away from x0
// PCL rounds T(x) p to float per point (3e-6 m of rounding dust on a 50 m coordinate, a relative 4e-7 of f on a 500k
// pair); here D z is exact.  Its BFGS stops at a gradient tolerance of 1e-2 wherever the line search lands, so that
// dust -- like a summation order, like any libm -- moves PCL's own stopping point by millimetres on noisy pairs and by
// nothing on pairs that register sharply (the reference's test cases): the measured spread is in
// tests/test_gicp_quad_gpu.py.  Because that spread exceeds north_star's 1e-4 m on noisy pairs this form is an OPT-IN
// (wm_gicp_params::objective = WM_GICP_OBJECTIVE_STATISTICS); the default, WM_GICP_OBJECTIVE_PCL_SUMS, is the per-pair float path.
//
// The 74 sums are accumulated in double-double (error-free TwoSum: the correctly rounded exact sum, whatever the
// order), and this evaluator is ONE source for the host (one registration on the whole device), the device (one
// registration per workgroup) and -- restated in C -- the oracle (oracle/gicp.c, objective mode 1): the three produce
// the same bits, so the batched path still EQUALS the one-pair path and both equal the oracle's restatement of this
// objective (tests/test_gicp_gpu.py, tests/test_gicp_batch_gpu.py).
#pragma once
#include "wm_bfgs.hpp"
#include "wm_math.hpp"

namespace wm {

constexpr int kQuadA = 60, kQuadB = 12;
constexpr int kQuadN = 74;  // A[60] | B0[12] at 60 | C0 at 72 | pair count at 73
constexpr int kQuadOffB = 60, kQuadOffC = 72, kQuadOffCount = 73;

// index of the unordered pair (a, c), a, c in 0..2: (0,0) 0, (0,1) 1, (0,2) 2, (1,1) 3, (1,2) 4, (2,2) 5
WM_HD int quad_s6(int a, int c) {
    const int lo = a < c ? a : c, hi = a < c ? c : a;
    return lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
}
// ... of (j, k), j, k in 0..3: (0,0) 0, (0,1) 1, (0,2) 2, (0,3) 3, (1,1) 4, (1,2) 5, (1,3) 6, (2,2) 7, (2,3) 8, (3,3) 9
WM_HD int quad_s10(int j, int k) {
    const int lo = j < k ? j : k, hi = j < k ? k : j;
    return lo == 0 ? hi : (lo == 1 ? 3 + hi : (lo == 2 ? 5 + hi : 9));
}

// f and (g != nullptr) its gradient at state x from the statistics Q of the pairs found under T0.
// base = base_transformation_ (the guess align() was given; identity in wm_gicp_align).
// Fixed operation order, no contraction: the host, the device and the oracle's C restatement give the same bits.
WM_HD double gicp_quad_eval(const double (&Q)[kQuadN], const float (&T0)[12], const double base[16], const double x[6],
                            double g[6]) {
    float T[16];
    state_to_matrix_f(base, x, T);
    double D[12];
    for (int k = 0; k < 12; ++k) D[k] = (double) T[k] - (double) T0[k];  // (a difference of two floats: exact)
    double G[12];
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < 4; ++j) {
            double s = Q[kQuadOffB + a * 4 + j];
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 4; ++k) s += D[c * 4 + k] * Q[quad_s6(a, c) * 10 + quad_s10(j, k)];
            G[a * 4 + j] = s;
        }
    double fm = Q[kQuadOffC];
    for (int k = 0; k < 12; ++k) fm += D[k] * (Q[kQuadOffB + k] + G[k]);
    const double m = Q[kQuadOffCount];
    if (g) {
        double Racc[9];
        for (int a = 0; a < 3; ++a) g[a] = G[a * 4 + 3] * 2.0 / m;
        // PCL: R += p_base_src * temp^T with p_base_src = base * p, temp = M r
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0.0;
                for (int j = 0; j < 4; ++j) s += base[a * 4 + j] * G[b * 4 + j];
                Racc[a * 3 + b] = s * 2.0 / m;
            }
        r_derivative(x, Racc, g);
    }
    return fm / m;
}

#if defined(__HIPCC__)
// one matched pair's 74 terms (p: the source point, q: its match, M: the pair's Mahalanobis matrix, T0: the float
// transform the pair was found under), handed to `add(index, term)` in index order
template <class Add>
__device__ __forceinline__ void gicp_quad_terms(const float (&T0)[12], float px, float py, float pz, float qx, float qy,
                                                float qz, const double (&M)[9], Add add) {
    // PCL's residual at T0: the float transform ((m00 x + m01 y) + m02 z) + m03, the float subtraction, then widened
    const float ppx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T0[0], px), __fmul_rn(T0[1], py)), __fmul_rn(T0[2], pz)), T0[3]);
    const float ppy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T0[4], px), __fmul_rn(T0[5], py)), __fmul_rn(T0[6], pz)), T0[7]);
    const float ppz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T0[8], px), __fmul_rn(T0[9], py)), __fmul_rn(T0[10], pz)), T0[11]);
    const double r0[3] = {(double) __fsub_rn(ppx, qx), (double) __fsub_rn(ppy, qy), (double) __fsub_rn(ppz, qz)};
    // the symmetric part of M (M is symmetric up to the rounding of its inverse; the quadratic form sees only this)
    const double Ms[6] = {M[0], 0.5 * (M[1] + M[3]), 0.5 * (M[2] + M[6]), M[4], 0.5 * (M[5] + M[7]), M[8]};
    const double z[4] = {(double) px, (double) py, (double) pz, 1.0};
    double zz[10];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = j; k < 4; ++k) zz[quad_s10(j, k)] = z[j] * z[k];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int t = 0; t < 10; ++t) add(s * 10 + t, Ms[s] * zz[t]);
    double t0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) t0[a] = (Ms[quad_s6(a, 0)] * r0[0] + Ms[quad_s6(a, 1)] * r0[1]) + Ms[quad_s6(a, 2)] * r0[2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) add(kQuadOffB + a * 4 + j, t0[a] * z[j]);
    add(kQuadOffC, (r0[0] * t0[0] + r0[1] * t0[1]) + r0[2] * t0[2]);
    add(kQuadOffCount, 1.0);
}

// ---- double-double sums of N components across a wavefront (recursive halving, as dd_halve of wm_gicp_dev.hpp for
// the thirteen sums of the per-pair objective: at the step for lane bit M a lane keeps one half of its pairs and
// gives the other half to lane ^ M).  Afterwards component k is in (hi[0], lo[0]) of the lane ddn_comp_of_lane names.
__device__ __forceinline__ void ddn_add(double &hi, double &lo, double x) {
    const double s = hi + x;
    const double bb = s - hi;
    lo += (hi - (s - bb)) + (x - bb);
    hi = s;
}
template <int N, int C, int M>
__device__ __forceinline__ void ddn_halve(double (&hi)[N], double (&lo)[N], unsigned lane) {
    constexpr int H = (C + 1) / 2;
    const bool up = (lane & (unsigned) M) != 0u;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double h_lo = hi[i], l_lo = lo[i];
        const double h_hi = (H + i < C) ? hi[H + i] : 0.0, l_hi = (H + i < C) ? lo[H + i] : 0.0;
        const double sh = up ? h_lo : h_hi, sl = up ? l_lo : l_hi;  // the half this lane gives away
        double kh = up ? h_hi : h_lo, kl = up ? l_hi : l_lo;        // the half it keeps
        const double rh = __shfl_xor(sh, M), rl = __shfl_xor(sl, M);
        ddn_add(kh, kl, rh);
        kl += rl;
        hi[i] = kh;
        lo[i] = kl;
    }
    if constexpr (M > 1) ddn_halve<N, H, M / 2>(hi, lo, lane);
}
template <int N>
__device__ __forceinline__ int ddn_comp_of_lane(unsigned lane) {
    int c = N, base = 0, valid = N;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const int h = (c + 1) / 2;
        if (lane & (unsigned) m) {
            base += h;
            valid -= h;
        } else {
            valid = valid < h ? valid : h;
        }
        c = h;
    }
    return valid >= 1 ? base : -1;
}
#endif  // __HIPCC__

}  // namespace wm
