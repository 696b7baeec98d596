/*
 * linalg.c -- ORACLE (test infrastructure): small dense routines standing in for
 * the Eigen calls PCL / libwave make on this path (JacobiSVD, SelfAdjointEigenSolver,
 * PartialPivLU::inverse, umeyama, eulerAngles).  Eigen is not vendored in
 * /root/reference; the algorithms are restated from their published definitions.
 */
#include "wm_oracle.h"
#include "wmo_internal.h"

#include <math.h>
#include <string.h>

/* One-sided (Hestenes) Jacobi SVD, A = U diag(S) V^T, n <= WMO_MAXN.
 * Singular values sorted descending; U is completed to an orthonormal basis
 * when A is rank deficient. */
void wmo_svd(int n, const double *A, double *U, double *S, double *V) {
    double W[WMO_MAXN * WMO_MAXN];
    int i, j, k, sweep;
    memcpy(W, A, sizeof(double) * n * n);
    for (i = 0; i < n; ++i)
        for (j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (i = 0; i < n - 1; ++i) {
            for (j = i + 1; j < n; ++j) {
                double alpha = 0, beta = 0, gamma = 0;
                for (k = 0; k < n; ++k) {
                    alpha += W[k * n + i] * W[k * n + i];
                    beta += W[k * n + j] * W[k * n + j];
                    gamma += W[k * n + i] * W[k * n + j];
                }
                if (gamma == 0.0 || fabs(gamma) <= 1e-300) continue;
                if (fabs(gamma) <= 2.3e-16 * sqrt(alpha * beta)) continue;
                rotated = 1;
                {
                    double zeta = (beta - alpha) / (2.0 * gamma);
                    double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    for (k = 0; k < n; ++k) {
                        double wi = W[k * n + i], wj = W[k * n + j];
                        W[k * n + i] = c * wi - s * wj;
                        W[k * n + j] = s * wi + c * wj;
                        wi = V[k * n + i];
                        wj = V[k * n + j];
                        V[k * n + i] = c * wi - s * wj;
                        V[k * n + j] = s * wi + c * wj;
                    }
                }
            }
        }
        if (!rotated) break;
    }
    /* singular values and ordering */
    {
        int order[WMO_MAXN];
        double sv[WMO_MAXN], Vt[WMO_MAXN * WMO_MAXN];
        double smax = 0;
        for (j = 0; j < n; ++j) {
            double s2 = 0;
            for (k = 0; k < n; ++k) s2 += W[k * n + j] * W[k * n + j];
            sv[j] = sqrt(s2);
            order[j] = j;
            if (sv[j] > smax) smax = sv[j];
        }
        for (i = 0; i < n - 1; ++i)
            for (j = i + 1; j < n; ++j)
                if (sv[order[j]] > sv[order[i]]) {
                    int t = order[i];
                    order[i] = order[j];
                    order[j] = t;
                }
        memcpy(Vt, V, sizeof(double) * n * n);
        for (j = 0; j < n; ++j) {
            int o = order[j];
            S[j] = sv[o];
            for (k = 0; k < n; ++k) {
                V[k * n + j] = Vt[k * n + o];
                U[k * n + j] = (sv[o] > 1e-300 && sv[o] > 1e-14 * smax) ? W[k * n + o] / sv[o] : 0.0;
            }
        }
        /* complete U by Gram-Schmidt against the canonical basis where needed */
        for (j = 0; j < n; ++j) {
            double nrm = 0;
            for (k = 0; k < n; ++k) nrm += U[k * n + j] * U[k * n + j];
            if (nrm > 0.5) continue;
            {
                int e, done = 0;
                for (e = 0; e < n && !done; ++e) {
                    double v[WMO_MAXN];
                    int c;
                    for (k = 0; k < n; ++k) v[k] = (k == e);
                    for (c = 0; c < n; ++c) {
                        double dot = 0, cn = 0;
                        if (c == j) continue;
                        for (k = 0; k < n; ++k) cn += U[k * n + c] * U[k * n + c];
                        if (cn < 0.5) continue;
                        for (k = 0; k < n; ++k) dot += U[k * n + c] * v[k];
                        for (k = 0; k < n; ++k) v[k] -= dot * U[k * n + c];
                    }
                    nrm = 0;
                    for (k = 0; k < n; ++k) nrm += v[k] * v[k];
                    if (nrm > 1e-6) {
                        nrm = sqrt(nrm);
                        for (k = 0; k < n; ++k) U[k * n + j] = v[k] / nrm;
                        done = 1;
                    }
                }
            }
        }
    }
}

/* Cyclic two-sided Jacobi for a symmetric matrix.  evals ascending (as
 * Eigen::SelfAdjointEigenSolver), evecs columns. */
void wmo_sym_eig(int n, const double *Ain, double *evals, double *evecs) {
    double A[WMO_MAXN * WMO_MAXN];
    int i, j, k, sweep;
    memcpy(A, Ain, sizeof(double) * n * n);
    for (i = 0; i < n; ++i)
        for (j = 0; j < n; ++j) evecs[i * n + j] = (i == j);
    for (sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (i = 0; i < n - 1; ++i) {
            for (j = i + 1; j < n; ++j) {
                double apq = A[i * n + j];
                if (fabs(apq) < 1e-300) continue;
                {
                    double theta = (A[j * n + j] - A[i * n + i]) / (2.0 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                    for (k = 0; k < n; ++k) {
                        double akp = A[k * n + i], akq = A[k * n + j];
                        A[k * n + i] = c * akp - s * akq;
                        A[k * n + j] = s * akp + c * akq;
                    }
                    for (k = 0; k < n; ++k) {
                        double apk = A[i * n + k], aqk = A[j * n + k];
                        A[i * n + k] = c * apk - s * aqk;
                        A[j * n + k] = s * apk + c * aqk;
                    }
                    for (k = 0; k < n; ++k) {
                        double vkp = evecs[k * n + i], vkq = evecs[k * n + j];
                        evecs[k * n + i] = c * vkp - s * vkq;
                        evecs[k * n + j] = s * vkp + c * vkq;
                    }
                }
            }
        }
    }
    for (i = 0; i < n; ++i) evals[i] = A[i * n + i];
    /* ascending sort */
    for (i = 0; i < n - 1; ++i)
        for (j = i + 1; j < n; ++j)
            if (evals[j] < evals[i]) {
                double t = evals[i];
                evals[i] = evals[j];
                evals[j] = t;
                for (k = 0; k < n; ++k) {
                    t = evecs[k * n + i];
                    evecs[k * n + i] = evecs[k * n + j];
                    evecs[k * n + j] = t;
                }
            }
}

/* Gauss-Jordan with partial pivoting (Eigen's MatrixBase::inverse for n>4 is
 * PartialPivLU).  returns 0 ok, 1 singular (result then holds inf/nan like Eigen) */
int wmo_inverse(int n, const double *A, double *Ainv) {
    double M[WMO_MAXN * 2 * WMO_MAXN];
    int i, j, k, singular = 0;
    int w = 2 * n;
    for (i = 0; i < n; ++i)
        for (j = 0; j < n; ++j) {
            M[i * w + j] = A[i * n + j];
            M[i * w + n + j] = (i == j);
        }
    for (k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(M[k * w + k]);
        for (i = k + 1; i < n; ++i)
            if (fabs(M[i * w + k]) > best) {
                best = fabs(M[i * w + k]);
                piv = i;
            }
        if (best == 0.0) singular = 1;
        if (piv != k)
            for (j = 0; j < w; ++j) {
                double t = M[k * w + j];
                M[k * w + j] = M[piv * w + j];
                M[piv * w + j] = t;
            }
        {
            double d = 1.0 / M[k * w + k];
            for (j = 0; j < w; ++j) M[k * w + j] *= d;
        }
        for (i = 0; i < n; ++i) {
            double f;
            if (i == k) continue;
            f = M[i * w + k];
            if (f == 0.0) continue;
            for (j = 0; j < w; ++j) M[i * w + j] -= f * M[k * w + j];
        }
    }
    for (i = 0; i < n; ++i)
        for (j = 0; j < n; ++j) Ainv[i * n + j] = M[i * w + n + j];
    return singular;
}

void wmo_mat_mul(int n, const double *A, const double *B, double *C) {
    double T[WMO_MAXN * WMO_MAXN];
    int i, j, k;
    for (i = 0; i < n; ++i)
        for (j = 0; j < n; ++j) {
            double s = 0;
            for (k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
            T[i * n + j] = s;
        }
    memcpy(C, T, sizeof(double) * n * n);
}

double wmo_det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* SVD-based least squares x = V S^+ U^T b (Eigen JacobiSVD::solve with the
 * default rank threshold n*eps*smax) */
void wmo_svd_solve(int n, const double *A, const double *b, double *x) {
    double U[WMO_MAXN * WMO_MAXN], S[WMO_MAXN], V[WMO_MAXN * WMO_MAXN], y[WMO_MAXN];
    int i, j;
    double thr;
    wmo_svd(n, A, U, S, V);
    thr = S[0] * n * 2.220446049250313e-16;
    for (j = 0; j < n; ++j) {
        double s = 0;
        for (i = 0; i < n; ++i) s += U[i * n + j] * b[i];
        y[j] = (S[j] > thr) ? s / S[j] : 0.0;
    }
    for (i = 0; i < n; ++i) {
        double s = 0;
        for (j = 0; j < n; ++j) s += V[i * n + j] * y[j];
        x[i] = s;
    }
}

/* From the 17 sufficient statistics to the rigid fit (Eigen::umeyama,
 * with_scaling=false): sigma = (1/n) sum (q-qm)(p-pm)^T, R = U S V^T,
 * S = diag(1,1,det(U)det(V)), t = qm - R pm.  p = src, q = dst. */
void wmo_umeyama_from_stats(double n, const double sp[3], const double sq[3],
                            const double sqp[9], double T[16]) {
    double pm[3], qm[3], sigma[9], U[9], S[3], V[9], R[9];
    int i, j, k;
    for (i = 0; i < 3; ++i) {
        pm[i] = sp[i] / n;
        qm[i] = sq[i] / n;
    }
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) sigma[i * 3 + j] = sqp[i * 3 + j] / n - qm[i] * pm[j];
    wmo_svd(3, sigma, U, S, V);
    {
        double d = wmo_det3(U) * wmo_det3(V);
        double s2 = (d < 0) ? -1.0 : 1.0;
        for (i = 0; i < 3; ++i)
            for (j = 0; j < 3; ++j) {
                double s = 0;
                for (k = 0; k < 3; ++k) s += U[i * 3 + k] * (k == 2 ? s2 : 1.0) * V[j * 3 + k];
                R[i * 3 + j] = s;
            }
    }
    for (i = 0; i < 16; ++i) T[i] = 0;
    for (i = 0; i < 3; ++i) {
        for (j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = qm[i] - (R[i * 3 + 0] * pm[0] + R[i * 3 + 1] * pm[1] + R[i * 3 + 2] * pm[2]);
    }
    T[15] = 1;
}

/* Eigen::umeyama(src, dst, false) on n point pairs.  float_sums=1 mirrors
 * PCL's Scalar=float: float means, float demeaned outer-product accumulation. */
void wmo_umeyama(const float *src, const float *dst, int n, int float_sums, double T[16]) {
    int i, a, b;
    if (float_sums) {
        float pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0}, sig[9] = {0};
        double sp[3], sq[3], sqp[9];
        for (i = 0; i < n; ++i)
            for (a = 0; a < 3; ++a) {
                pm[a] += src[3 * i + a];
                qm[a] += dst[3 * i + a];
            }
        for (a = 0; a < 3; ++a) {
            pm[a] /= (float) n;
            qm[a] /= (float) n;
        }
        for (i = 0; i < n; ++i) {
            float dp[3], dq[3];
            for (a = 0; a < 3; ++a) {
                dp[a] = src[3 * i + a] - pm[a];
                dq[a] = dst[3 * i + a] - qm[a];
            }
            for (a = 0; a < 3; ++a)
                for (b = 0; b < 3; ++b) sig[a * 3 + b] += dq[a] * dp[b];
        }
        /* rebuild raw statistics so one solver serves both paths */
        for (a = 0; a < 3; ++a) {
            sp[a] = (double) pm[a] * n;
            sq[a] = (double) qm[a] * n;
        }
        for (a = 0; a < 3; ++a)
            for (b = 0; b < 3; ++b)
                sqp[a * 3 + b] = (double) sig[a * 3 + b] + (double) qm[a] * (double) pm[b] * n;
        wmo_umeyama_from_stats((double) n, sp, sq, sqp, T);
    } else {
        double sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, sqp[9] = {0};
        for (i = 0; i < n; ++i) {
            for (a = 0; a < 3; ++a) {
                sp[a] += src[3 * i + a];
                sq[a] += dst[3 * i + a];
            }
            for (a = 0; a < 3; ++a)
                for (b = 0; b < 3; ++b) sqp[a * 3 + b] += (double) dst[3 * i + a] * (double) src[3 * i + b];
        }
        wmo_umeyama_from_stats((double) n, sp, sq, sqp, T);
    }
}

/* Eigen 3.3 MatrixBase::eulerAngles(0,1,2) (reference call:
 * wave_matching/src/icp.cpp:175).  R row-major 3x3. */
void wmo_euler_angles_012(const double R[9], double e[3]) {
    const int i = 0, j = 1, k = 2; /* odd = 0 */
    double c2, s1, c1;
    e[0] = atan2(R[j * 3 + k], R[k * 3 + k]);
    c2 = sqrt(R[i * 3 + i] * R[i * 3 + i] + R[i * 3 + j] * R[i * 3 + j]);
    if (e[0] > 0.0) { /* (!odd) && res[0] > 0 */
        if (e[0] > 0.0)
            e[0] -= M_PI;
        else
            e[0] += M_PI;
        e[1] = atan2(-R[i * 3 + k], -c2);
    } else {
        e[1] = atan2(-R[i * 3 + k], c2);
    }
    s1 = sin(e[0]);
    c1 = cos(e[0]);
    e[2] = atan2(s1 * R[k * 3 + i] - c1 * R[j * 3 + i], c1 * R[j * 3 + j] - s1 * R[k * 3 + j]);
    e[0] = -e[0];
    e[1] = -e[1];
    e[2] = -e[2];
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3 matrix, cyclic sweeps over the column pairs (0,1), (0,2),
 * (1,2): A = U diag(S) V^T, S descending, U completed to an orthonormal basis when A is rank
 * deficient.  Stated with IEEE operations only and in a fixed order, so that the HIP build of the
 * same published algorithm (libwave_amd/csrc/wm_math.hpp: svd3<false>) reproduces it bit for bit:
 * GICP's covariances go through it, and their last bits steer PCL's loosely converged BFGS
 * (gradient tolerance 1e-2) -- see tests/test_gicp_gpu.py.  (Eigen's JacobiSVD, which PCL calls, is
 * a two-sided Jacobi with its own operation order: all three agree to ~1e-15.) */
static int jacobi_pair3(double *W, double *V, int I, int J) {
    double alpha = 0, beta = 0, gamma = 0, d, g, t, c, s;
    int k;
    for (k = 0; k < 3; ++k) {
        alpha += W[k * 3 + I] * W[k * 3 + I];
        beta += W[k * 3 + J] * W[k * 3 + J];
        gamma += W[k * 3 + I] * W[k * 3 + J];
    }
    if (fabs(gamma) <= 1e-300 || gamma * gamma <= (2.3e-16 * 2.3e-16) * (alpha * beta)) return 0;
    d = beta - alpha;
    g = 2.0 * gamma;
    t = ((d >= 0) == (g >= 0) ? fabs(g) : -fabs(g)) / (fabs(d) + sqrt(d * d + g * g));
    c = 1.0 / sqrt(1.0 + t * t);
    s = c * t;
    for (k = 0; k < 3; ++k) {
        const double wi = W[k * 3 + I], wj = W[k * 3 + J];
        const double vi = V[k * 3 + I], vj = V[k * 3 + J];
        W[k * 3 + I] = c * wi - s * wj;
        W[k * 3 + J] = s * wi + c * wj;
        V[k * 3 + I] = c * vi - s * vj;
        V[k * 3 + J] = s * vi + c * vj;
    }
    return 1;
}

static void swap_cols3(double *sv, double *W, double *V, int I, int J) { /* ensure sv[I] >= sv[J] */
    int k;
    if (sv[J] > sv[I]) {
        double t = sv[I];
        sv[I] = sv[J];
        sv[J] = t;
        for (k = 0; k < 3; ++k) {
            t = W[k * 3 + I];
            W[k * 3 + I] = W[k * 3 + J];
            W[k * 3 + J] = t;
            t = V[k * 3 + I];
            V[k * 3 + I] = V[k * 3 + J];
            V[k * 3 + J] = t;
        }
    }
}

void wmo_svd3_jacobi(const double *A, double *U, double *S, double *V) {
    double W[9], sv[3], smax;
    int i, j, k, sweep, h0, h1, h2;
    for (i = 0; i < 9; ++i) {
        W[i] = A[i];
        V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    for (sweep = 0; sweep < 60; ++sweep) {
        const int r0 = jacobi_pair3(W, V, 0, 1);
        const int r1 = jacobi_pair3(W, V, 0, 2);
        const int r2 = jacobi_pair3(W, V, 1, 2);
        if (!(r0 || r1 || r2)) break;
    }
    for (j = 0; j < 3; ++j) sv[j] = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
    swap_cols3(sv, W, V, 0, 1);
    swap_cols3(sv, W, V, 0, 2);
    swap_cols3(sv, W, V, 1, 2);
    smax = sv[0];
    h0 = (sv[0] > 1e-300);
    h1 = h0 && (sv[1] > 1e-300 && sv[1] > 1e-14 * smax);
    h2 = h1 && (sv[2] > 1e-300 && sv[2] > 1e-14 * smax);
    for (k = 0; k < 3; ++k) {
        S[k] = sv[k];
        U[k * 3 + 0] = h0 ? W[k * 3 + 0] / sv[0] : 0.0;
        U[k * 3 + 1] = h1 ? W[k * 3 + 1] / sv[1] : 0.0;
        U[k * 3 + 2] = h2 ? W[k * 3 + 2] / sv[2] : 0.0;
    }
    if (!h0) {
        for (i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    if (!h1) { /* any unit vector orthogonal to u0: drop the smallest component */
        const double a0 = U[0], a1 = U[3], a2 = U[6];
        double e0 = 0, e1 = 0, e2 = 0, d, v0, v1, v2, n;
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
        v0 = e0 - d * a0;
        v1 = e1 - d * a1;
        v2 = e2 - d * a2;
        n = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
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
