// wm_math.hpp -- small fixed-size double-precision linear algebra shared by the
// host solvers and the single-thread device "solve" kernels (3x3 SVD for the
// Umeyama step, 6x6 solves for GN / NDT / LUM).  Header-only; every function is
// usable from host and device code.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WM_HD __host__ __device__ inline
#else
#define WM_HD inline
#endif

namespace wm {

// statistics layout (see include/wavematch.h)
constexpr int kStatsLen = 32;
constexpr int kSvdN = 0, kSvdSp = 1, kSvdSq = 4, kSvdSqp = 7, kSvdSd2 = 16;
constexpr int kGnN = 0, kGnSd2 = 1, kGnH = 2, kGnG = 23;

WM_HD void mat4_identity(double *T) {
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

WM_HD void mat4_mul(const double *A, const double *B, double *C) {
    double t[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            t[i * 4 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) C[i] = t[i];
}

WM_HD double det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// One-sided Jacobi SVD of a 3x3 matrix: A = U diag(S) V^T, S descending, U
// completed to an orthonormal basis when A is rank deficient.  Written with
// compile-time indices only (every loop fully unrolled) so that on the GPU the
// working set lives in registers, not scratch.
namespace detail {
// 1 / sqrt(x) and 1 / x for x well inside the normal range.  On the GPU these run in ONE lane of the
// solve kernel, where a correctly rounded f64 division or square root is a ~60-instruction dependent
// chain (scale, estimate, refine, fix up): the hardware estimate plus two Newton steps gives full
// double precision in a dozen.  A Jacobi rotation does not need more (any (c, s) with
// c^2 + s^2 = 1 to rounding is a valid rotation; t only steers convergence).
WM_HD double fast_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - (0.5 * x) * (y * y));
    y = y * (1.5 - (0.5 * x) * (y * y));
    return y;
#else
    return 1.0 / sqrt(x);
#endif
}
WM_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
#else
    return 1.0 / x;
#endif
}
// FAST: the reciprocal / square-root estimates above (the ICP solve, one GPU lane).  !FAST: IEEE
// divisions and square roots only -- every operation correctly rounded, so the CPU oracle, which
// states the same algorithm in the same order (oracle/linalg.c: wmo_svd3_jacobi), reproduces the
// result bit for bit (the GICP covariances, whose last bits steer PCL's loosely converged BFGS).
template <int I, int J, bool FAST>
WM_HD bool jacobi_pair(double *W, double *V) {
    double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        alpha += W[k * 3 + I] * W[k * 3 + I];
        beta += W[k * 3 + J] * W[k * 3 + J];
        gamma += W[k * 3 + I] * W[k * 3 + J];
    }
    // columns already orthogonal to working precision?  (squared form: no square root)
    if (fabs(gamma) <= 1e-300 || gamma * gamma <= (2.3e-16 * 2.3e-16) * (alpha * beta)) return false;
    // tan of the rotation angle, smaller root: with zeta = (beta - alpha) / (2 gamma),
    //   t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) = sign(d g) |g| / (|d| + sqrt(d^2 + g^2)),
    // d = beta - alpha, g = 2 gamma
    const double d = beta - alpha, g = 2.0 * gamma;
    const double h2 = d * d + g * g;
    double t, c;
    if (FAST && h2 > 1e-280 && h2 < 1e280) {
        const double hyp = h2 * fast_rsqrt(h2);
        const double ta = fabs(g) * fast_rcp(fabs(d) + hyp);
        t = ((d >= 0) == (g >= 0)) ? ta : -ta;
        c = fast_rsqrt(1.0 + t * t);
    } else {  // far out of range: the slow, scale-safe forms
        t = ((d >= 0) == (g >= 0) ? fabs(g) : -fabs(g)) / (fabs(d) + sqrt(d * d + g * g));
        c = 1.0 / sqrt(1.0 + t * t);
    }
    const double s = c * t;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double wi = W[k * 3 + I], wj = W[k * 3 + J];
        W[k * 3 + I] = c * wi - s * wj;
        W[k * 3 + J] = s * wi + c * wj;
        const double vi = V[k * 3 + I], vj = V[k * 3 + J];
        V[k * 3 + I] = c * vi - s * vj;
        V[k * 3 + J] = s * vi + c * vj;
    }
    return true;
}
template <int I, int J>
WM_HD void swap_cols_if_less(double *sv, double *W, double *V) {  // ensure sv[I] >= sv[J]
    if (sv[J] > sv[I]) {
        double t = sv[I];
        sv[I] = sv[J];
        sv[J] = t;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            t = W[k * 3 + I];
            W[k * 3 + I] = W[k * 3 + J];
            W[k * 3 + J] = t;
            t = V[k * 3 + I];
            V[k * 3 + I] = V[k * 3 + J];
            V[k * 3 + J] = t;
        }
    }
}
}  // namespace detail

// V0 (may be null): an orthogonal matrix to start from -- the V of a nearby matrix's SVD (the
// previous ICP iteration's): the columns of A V0 are then almost orthogonal already and one sweep
// plus the checking sweep do, instead of five or six.
template <bool FAST = true>
WM_HD void svd3(const double *A, double *U, double *S, double *V, const double *V0 = nullptr) {
    double W[9];
    if (V0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = V0[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                W[i * 3 + j] = A[i * 3] * V0[j] + A[i * 3 + 1] * V0[3 + j] + A[i * 3 + 2] * V0[6 + j];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            W[i] = A[i];
            V[i] = (i % 4 == 0) ? 1.0 : 0.0;
        }
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool r0 = detail::jacobi_pair<0, 1, FAST>(W, V);
        bool r1 = detail::jacobi_pair<0, 2, FAST>(W, V);
        bool r2 = detail::jacobi_pair<1, 2, FAST>(W, V);
        if (!(r0 || r1 || r2)) break;
    }
    double sv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double n2 = W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j];
        sv[j] = (FAST && n2 > 1e-280 && n2 < 1e280) ? n2 * detail::fast_rsqrt(n2) : sqrt(n2);
    }
    detail::swap_cols_if_less<0, 1>(sv, W, V);
    detail::swap_cols_if_less<0, 2>(sv, W, V);
    detail::swap_cols_if_less<1, 2>(sv, W, V);
    const double smax = sv[0];
    const bool h0 = (sv[0] > 1e-300);
    const bool h1 = h0 && (sv[1] > 1e-300 && sv[1] > 1e-14 * smax);
    const bool h2 = h1 && (sv[2] > 1e-300 && sv[2] > 1e-14 * smax);
    if (FAST) {
        // (one reciprocal per column, not one division per entry: this runs in a single GPU lane)
        const double i0 = h0 ? detail::fast_rcp(sv[0]) : 0.0, i1 = h1 ? detail::fast_rcp(sv[1]) : 0.0,
                     i2 = h2 ? detail::fast_rcp(sv[2]) : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            S[k] = sv[k];
            U[k * 3 + 0] = W[k * 3 + 0] * i0;
            U[k * 3 + 1] = W[k * 3 + 1] * i1;
            U[k * 3 + 2] = W[k * 3 + 2] * i2;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            S[k] = sv[k];
            U[k * 3 + 0] = h0 ? W[k * 3 + 0] / sv[0] : 0.0;
            U[k * 3 + 1] = h1 ? W[k * 3 + 1] / sv[1] : 0.0;
            U[k * 3 + 2] = h2 ? W[k * 3 + 2] / sv[2] : 0.0;
        }
    }
    if (!h0) {  // zero matrix
#pragma unroll
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    if (!h1) {  // any unit vector orthogonal to u0: drop the smallest component
        const double a0 = U[0], a1 = U[3], a2 = U[6];
        double e0 = 0, e1 = 0, e2 = 0, d;
        if (fabs(a0) <= fabs(a1) && fabs(a0) <= fabs(a2)) {
            e0 = 1;
            d = a0;
        } else if (fabs(a1) <= fabs(a2)) {
            e1 = 1;
            d = a1;
        } else {
            e2 = 1;
            d = a2;
        }
        const double v0 = e0 - d * a0, v1 = e1 - d * a1, v2 = e2 - d * a2;
        const double n = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        U[1] = v0 / n;
        U[4] = v1 / n;
        U[7] = v2 / n;
    }
    if (!h2) {
        const double a0 = U[0], a1 = U[3], a2 = U[6], b0 = U[1], b1 = U[4], b2 = U[7];
        U[2] = a1 * b2 - a2 * b1;
        U[5] = a2 * b0 - a0 * b2;
        U[8] = a0 * b1 - a1 * b0;
    }
}

// Cholesky solve of a symmetric positive definite system (compile-time indices).
// returns false when A is not positive definite.
template <int N>
WM_HD bool chol_solve(const double *A, const double *b, double *x) {
    double L[N * N];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j > i) continue;
            double s = A[i * N + j];
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (k < j) s -= L[i * N + k] * L[j * N + k];
            if (i == j) {
                if (!(s > 0.0)) {
                    ok = false;
                    s = 1.0;
                }
                L[i * N + i] = sqrt(s);
            } else {
                L[i * N + j] = s / L[j * N + j];
            }
        }
    }
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (k < i) s -= L[i * N + k] * y[k];
        y[i] = s / L[i * N + i];
    }
#pragma unroll
    for (int ii = 0; ii < N; ++ii) {
        const int i = N - 1 - ii;
        double s = y[i];
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (k > i) s -= L[k * N + i] * x[k];
        x[i] = s / L[i * N + i];
    }
    return ok;
}

// Gauss-Jordan inverse with partial pivoting, N <= 6.  returns false if singular.
template <int N>
WM_HD bool inverse(const double *A, double *Ainv) {
    double M[N * 2 * N];
    const int w = 2 * N;
    bool ok = true;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            M[i * w + j] = A[i * N + j];
            M[i * w + N + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int k = 0; k < N; ++k) {
        int piv = k;
        double best = fabs(M[k * w + k]);
        for (int i = k + 1; i < N; ++i)
            if (fabs(M[i * w + k]) > best) {
                best = fabs(M[i * w + k]);
                piv = i;
            }
        if (best == 0.0) ok = false;
        if (piv != k)
            for (int j = 0; j < w; ++j) {
                double t = M[k * w + j];
                M[k * w + j] = M[piv * w + j];
                M[piv * w + j] = t;
            }
        double d = 1.0 / M[k * w + k];
        for (int j = 0; j < w; ++j) M[k * w + j] *= d;
        for (int i = 0; i < N; ++i) {
            if (i == k) continue;
            double f = M[i * w + k];
            if (f == 0.0) continue;
            for (int j = 0; j < w; ++j) M[i * w + j] -= f * M[k * w + j];
        }
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) Ainv[i * N + j] = M[i * w + N + j];
    return ok;
}

template <int N>
WM_HD void mat_mul(const double *A, const double *B, double *C) {
    double t[N * N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += A[i * N + k] * B[k * N + j];
            t[i * N + j] = s;
        }
#pragma unroll
    for (int i = 0; i < N * N; ++i) C[i] = t[i];
}

WM_HD void rodrigues(const double *w, double *R) {
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double a, b;
    if (th < 1e-12) {
        a = 1.0;
        b = 0.5;
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / (th * th);
    }
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    mat_mul<3>(K, K, K2);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// pcl::umeyama(src, dst, with_scaling=false) from the 17 sufficient statistics:
// sigma = (1/n) sum (q-qm)(p-pm)^T = Sqp/n - qm pm^T; R = U diag(1,1,det(U)det(V)) V^T;
// t = qm - R pm.  [PCL registration/impl/transformation_estimation_svd.hpp ->
// Eigen/src/Geometry/Umeyama.h], driven by wave_matching/src/icp.cpp:126.
// Vwarm (may be null): in, an orthogonal starting basis for the SVD (see svd3) unless Vwarm[9]
// is 0; out, this call's V and Vwarm[9] = 1.
WM_HD void umeyama_from_stats(const double *st, double *T, double *Vwarm = nullptr) {
    const double n = st[kSvdN];
    double pm[3], qm[3], sigma[9], U[9], S[3], V[9], R[9];
    const double one_over_n = 1.0 / n;  // as Eigen's umeyama does (means and sigma are scaled by it)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pm[i] = st[kSvdSp + i] * one_over_n;
        qm[i] = st[kSvdSq + i] * one_over_n;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sigma[i * 3 + j] = st[kSvdSqp + i * 3 + j] * one_over_n - qm[i] * pm[j];
    svd3(sigma, U, S, V, (Vwarm && Vwarm[9] != 0.0) ? Vwarm : nullptr);
    if (Vwarm) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Vwarm[i] = V[i];
        Vwarm[9] = 1.0;
    }
    const double s2 = (det3(U) * det3(V) < 0) ? -1.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += U[i * 3 + k] * (k == 2 ? s2 : 1.0) * V[j * 3 + k];
            R[i * 3 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = qm[i] - (R[i * 3 + 0] * pm[0] + R[i * 3 + 1] * pm[1] + R[i * 3 + 2] * pm[2]);
    }
    T[15] = 1;
}

// Gauss-Newton step: (J^T J) delta = -J^T r, T_k = [exp(dw) | dt].
WM_HD bool gn6_from_stats(const double *st, double *T) {
    double H[36], delta[6], rhs[6], R[9];
    {
        int k = 0;  // upper triangle, row-major
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j >= i) {
                    H[i * 6 + j] = st[kGnH + k];
                    H[j * 6 + i] = st[kGnH + k];
                    ++k;
                }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) rhs[i] = -st[kGnG + i];
    const bool ok = chol_solve<6>(H, rhs, delta);
    rodrigues(delta + 3, R);
    mat4_identity(T);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = delta[i];
    }
    return ok;
}

}  // namespace wm
