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
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

WM_HD void mat4_mul(const double *A, const double *B, double *C) {
    double t[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            t[i * 4 + j] = s;
        }
    for (int i = 0; i < 16; ++i) C[i] = t[i];
}

WM_HD double det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// One-sided Jacobi SVD of a 3x3 matrix: A = U diag(S) V^T, S descending, U
// completed to an orthonormal basis when A is rank deficient.
WM_HD void svd3(const double *A, double *U, double *S, double *V) {
    double W[9];
    for (int i = 0; i < 9; ++i) {
        W[i] = A[i];
        V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += W[k * 3 + i] * W[k * 3 + i];
                    beta += W[k * 3 + j] * W[k * 3 + j];
                    gamma += W[k * 3 + i] * W[k * 3 + j];
                }
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.3e-16 * sqrt(alpha * beta)) continue;
                rotated = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    double wi = W[k * 3 + i], wj = W[k * 3 + j];
                    W[k * 3 + i] = c * wi - s * wj;
                    W[k * 3 + j] = s * wi + c * wj;
                    double vi = V[k * 3 + i], vj = V[k * 3 + j];
                    V[k * 3 + i] = c * vi - s * vj;
                    V[k * 3 + j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    double sv[3];
    int ord[3] = {0, 1, 2};
    double smax = 0;
    for (int j = 0; j < 3; ++j) {
        sv[j] = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
        if (sv[j] > smax) smax = sv[j];
    }
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[ord[j]] > sv[ord[i]]) {
                int t = ord[i];
                ord[i] = ord[j];
                ord[j] = t;
            }
    double Vt[9];
    for (int i = 0; i < 9; ++i) Vt[i] = V[i];
    bool have[3];
    for (int j = 0; j < 3; ++j) {
        int o = ord[j];
        S[j] = sv[o];
        have[j] = (sv[o] > 1e-300 && sv[o] > 1e-14 * smax);
        for (int k = 0; k < 3; ++k) {
            V[k * 3 + j] = Vt[k * 3 + o];
            U[k * 3 + j] = have[j] ? W[k * 3 + o] / sv[o] : 0.0;
        }
    }
    // complete U (cross products / canonical fallbacks)
    if (!have[0]) {  // zero matrix
        for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    if (!have[1]) {
        // any unit vector orthogonal to u0
        double u0[3] = {U[0], U[3], U[6]};
        int m = 0;
        if (fabs(u0[1]) < fabs(u0[m])) m = 1;
        if (fabs(u0[2]) < fabs(u0[m])) m = 2;
        double e[3] = {0, 0, 0};
        e[m] = 1;
        double d = u0[m];
        double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        U[1] = v[0] / n;
        U[4] = v[1] / n;
        U[7] = v[2] / n;
        have[1] = true;
    }
    if (!have[2]) {
        double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
        U[2] = a[1] * b[2] - a[2] * b[1];
        U[5] = a[2] * b[0] - a[0] * b[2];
        U[8] = a[0] * b[1] - a[1] * b[0];
    }
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
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < N; ++k) s += A[i * N + k] * B[k * N + j];
            t[i * N + j] = s;
        }
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
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// pcl::umeyama(src, dst, with_scaling=false) from the 17 sufficient statistics:
// sigma = (1/n) sum (q-qm)(p-pm)^T = Sqp/n - qm pm^T; R = U diag(1,1,det(U)det(V)) V^T;
// t = qm - R pm.  [PCL registration/impl/transformation_estimation_svd.hpp ->
// Eigen/src/Geometry/Umeyama.h], driven by wave_matching/src/icp.cpp:126.
WM_HD void umeyama_from_stats(const double *st, double *T) {
    const double n = st[kSvdN];
    double pm[3], qm[3], sigma[9], U[9], S[3], V[9], R[9];
    for (int i = 0; i < 3; ++i) {
        pm[i] = st[kSvdSp + i] / n;
        qm[i] = st[kSvdSq + i] / n;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sigma[i * 3 + j] = st[kSvdSqp + i * 3 + j] / n - qm[i] * pm[j];
    svd3(sigma, U, S, V);
    const double s2 = (det3(U) * det3(V) < 0) ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U[i * 3 + k] * (k == 2 ? s2 : 1.0) * V[j * 3 + k];
            R[i * 3 + j] = s;
        }
    for (int i = 0; i < 16; ++i) T[i] = 0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = qm[i] - (R[i * 3 + 0] * pm[0] + R[i * 3 + 1] * pm[1] + R[i * 3 + 2] * pm[2]);
    }
    T[15] = 1;
}

// Gauss-Newton step: (J^T J) delta = -J^T r, T_k = [exp(dw) | dt].
WM_HD bool gn6_from_stats(const double *st, double *T) {
    double H[36], Hinv[36], delta[6], R[9];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
            H[i * 6 + j] = st[kGnH + k];
            H[j * 6 + i] = st[kGnH + k];
            ++k;
        }
    bool ok = inverse<6>(H, Hinv);
    for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += Hinv[i * 6 + j] * st[kGnG + j];
        delta[i] = -s;
    }
    rodrigues(delta + 3, R);
    mat4_identity(T);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = delta[i];
    }
    return ok;
}

}  // namespace wm
