/*
 * gicp.c -- ORACLE (test infrastructure): CPU restatement of
 *   pcl::GeneralizedIterativeClosestPoint<PointXYZ,PointXYZ>::align   (PCL 1.8)
 * as configured and driven by libwave's GICPMatcher:
 *   wave_matching/src/gicp.cpp:31-34  setCorrespondenceRandomness(corr_rand),
 *        setMaximumIterations, setRotationEpsilon(r_eps), setEuclideanFitnessEpsilon
 *        (max correspondence distance stays at PCL-GICP's 5 m, transformation_epsilon at
 *        5e-4, gicp_epsilon 1e-3, 20 inner iterations)
 *   wave_matching/src/gicp.cpp:37-64  setInputSource / setInputTarget / align
 * PCL is an un-vendored dependency; the algorithm follows its published sources
 *   registration/impl/gicp.hpp  (computeCovariances, computeTransformation,
 *        estimateRigidTransformationBFGS, OptimizationFunctorWithIndices::{(),df,fdf},
 *        computeRDerivative, applyState)
 *   registration/bfgs.h         (a port of GSL's vector_bfgs2 + Fletcher line search)
 * THIRD-PARTY PROVENANCE: PCL 1.8 (BSD-3) is restated from its published algorithm, not copied; its BFGS is
 * in turn GSL's vector_bfgs2 + Fletcher's line search (GNU Scientific Library).  Nothing here comes from
 * /root/reference, which only calls into PCL.
 * PARITY: unpinned (no PCL to run); pinned to the reference tests' assertions
 * (wave_matching/tests/gicp_tests.cpp:43-100) and a finite-difference gradient check.
 */
#include "wm_oracle.h"
#include "wmo_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

void wmo_gicp_default_params(wmo_gicp_params *p) {
    p->corr_rand = 10;      /* gicp.hpp:34 */
    p->max_iter = 100;      /* gicp.hpp:35 */
    p->r_eps = 1e-8;        /* gicp.hpp:36 */
    p->t_eps = 5e-4;        /* PCL GICP transformation_epsilon_ */
    p->max_corr = 5.0;      /* PCL GICP corr_dist_threshold_ */
    p->gicp_epsilon = 1e-3; /* PCL GICP gicp_epsilon_ */
    p->max_inner = 20;      /* max_inner_iterations_ */
    p->force_iterations = 0;
}

/* computeCovariances: k-NN in the cloud itself (self included), float products into
 * double sums, SVD, singular values replaced by (1, 1, eps) */
int wmo_gicp_covariances(const float *xyz, int n, int k, double eps, double *cov_out) {
    wmo_kdtree *tree;
    int *idx;
    float *d2;
    int i, j, a, b;
    if (k > n) return 1; /* "Number of points in cloud is less than k_correspondences_" */
    tree = wmo_kdtree_build(xyz, n);
    idx = (int *) malloc(sizeof(int) * k);
    d2 = (float *) malloc(sizeof(float) * k);
    for (i = 0; i < n; ++i) {
        double mean[3] = {0, 0, 0}, cov[9] = {0}, U[9], S[3], V[9];
        double *out = cov_out + 9 * i;
        wmo_kdtree_knn(tree, xyz + 3 * i, k, idx, d2);
        for (j = 0; j < k; ++j) {
            const float *pt = xyz + 3 * idx[j];
            mean[0] += pt[0];
            mean[1] += pt[1];
            mean[2] += pt[2];
            cov[0] += pt[0] * pt[0]; /* float products, as `cov(0,0) += pt.x*pt.x` */
            cov[3] += pt[1] * pt[0];
            cov[4] += pt[1] * pt[1];
            cov[6] += pt[2] * pt[0];
            cov[7] += pt[2] * pt[1];
            cov[8] += pt[2] * pt[2];
        }
        for (a = 0; a < 3; ++a) mean[a] /= (double) k;
        for (a = 0; a < 3; ++a)
            for (b = 0; b <= a; ++b) {
                cov[a * 3 + b] /= (double) k;
                cov[a * 3 + b] -= mean[a] * mean[b];
                cov[b * 3 + a] = cov[a * 3 + b];
            }
        wmo_svd3_jacobi(cov, U, S, V);
        for (a = 0; a < 9; ++a) out[a] = 0;
        for (j = 0; j < 3; ++j) {
            double v = (j == 2) ? eps : 1.0;
            for (a = 0; a < 3; ++a)
                for (b = 0; b < 3; ++b) out[a * 3 + b] += v * U[a * 3 + j] * U[b * 3 + j];
        }
    }
    free(idx);
    free(d2);
    wmo_kdtree_free(tree);
    return 0;
}

/* applyState on an identity base: T = [Rz(x5) Ry(x4) Rx(x3) | x0..2], float like PCL's
 * Eigen::Matrix4f transformation_matrix */
static void state_to_matrix_f(const double base[16], const double x[6], float T[16]) {
    float cphi = cosf((float) x[3]), sphi = sinf((float) x[3]);
    float cth = cosf((float) x[4]), sth = sinf((float) x[4]);
    float cpsi = cosf((float) x[5]), spsi = sinf((float) x[5]);
    float R[9], B[16], o[16];
    int i, j, k;
    /* Rz * Ry * Rx */
    R[0] = cpsi * cth;
    R[1] = cpsi * sth * sphi - spsi * cphi;
    R[2] = cpsi * sth * cphi + spsi * sphi;
    R[3] = spsi * cth;
    R[4] = spsi * sth * sphi + cpsi * cphi;
    R[5] = spsi * sth * cphi - cpsi * sphi;
    R[6] = -sth;
    R[7] = cth * sphi;
    R[8] = cth * cphi;
    for (i = 0; i < 16; ++i) B[i] = (float) base[i];
    memcpy(o, B, sizeof(o));
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) {
            float s = 0;
            for (k = 0; k < 3; ++k) s += R[i * 3 + k] * B[k * 4 + j];
            o[i * 4 + j] = s;
        }
    o[3] = B[3] + (float) x[0];
    o[7] = B[7] + (float) x[1];
    o[11] = B[11] + (float) x[2];
    memcpy(T, o, sizeof(o));
}

static void mul_pt_f(const float *T, const float *p, float *o) {
    o[0] = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3];
    o[1] = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7];
    o[2] = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11];
}

/* computeRDerivative: g[3..5] = tr(dR/dangle * R_acc) */
static void r_derivative(const double x[6], const double Racc[9], double g[6]) {
    double phi = x[3], theta = x[4], psi = x[5];
    double cphi = cos(phi), sphi = sin(phi), ctheta = cos(theta), stheta = sin(theta),
           cpsi = cos(psi), spsi = sin(psi);
    double dPhi[9], dTheta[9], dPsi[9];
    int i, j;
    dPhi[0] = 0;
    dPhi[3] = 0;
    dPhi[6] = 0;
    dPhi[1] = sphi * spsi + cphi * cpsi * stheta;
    dPhi[4] = -cpsi * sphi + cphi * spsi * stheta;
    dPhi[7] = cphi * ctheta;
    dPhi[2] = cphi * spsi - cpsi * sphi * stheta;
    dPhi[5] = -cphi * cpsi - sphi * spsi * stheta;
    dPhi[8] = -ctheta * sphi;
    dTheta[0] = -cpsi * stheta;
    dTheta[3] = -spsi * stheta;
    dTheta[6] = -ctheta;
    dTheta[1] = cpsi * ctheta * sphi;
    dTheta[4] = ctheta * sphi * spsi;
    dTheta[7] = -sphi * stheta;
    dTheta[2] = cphi * cpsi * ctheta;
    dTheta[5] = cphi * ctheta * spsi;
    dTheta[8] = -cphi * stheta;
    dPsi[0] = -ctheta * spsi;
    dPsi[3] = cpsi * ctheta;
    dPsi[6] = 0;
    dPsi[1] = -cphi * cpsi - sphi * spsi * stheta;
    dPsi[4] = -cphi * spsi + cpsi * sphi * stheta;
    dPsi[7] = 0;
    dPsi[2] = cpsi * sphi - cphi * spsi * stheta;
    dPsi[5] = sphi * spsi + cphi * cpsi * stheta;
    dPsi[8] = 0;
    g[3] = g[4] = g[5] = 0;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) { /* matricesInnerProd: sum mat1(j,i) * mat2(i,j) */
            g[3] += dPhi[j * 3 + i] * Racc[i * 3 + j];
            g[4] += dTheta[j * 3 + i] * Racc[i * 3 + j];
            g[5] += dPsi[j * 3 + i] * Racc[i * 3 + j];
        }
}

/* The sums of the objective are formed in double-double (error-free TwoSum of every term into a
 * (hi, lo) pair, one final rounding): the correctly rounded value of the exact sum of the terms,
 * whatever the order.  Eigen / PCL add plain doubles in index order; the difference is the last
 * bit -- but PCL's BFGS stops at a gradient tolerance of 1e-2, and a last-bit difference can flip a
 * line-search branch and move the stopping point by millimetres.  An order-independent sum is what
 * lets a parallel implementation (the HIP kernel adds in a strided tree order) make exactly the
 * decisions this sequential loop makes. */
/* Summation of the objective's 13 sums: 0 = double-double (default: what the HIP path is held to bit
 * for bit), 1 = PCL-LITERAL -- plain doubles added in index order, as Eigen does inside
 * OptimizationFunctorWithIndices::fdf (`f += double(res.transpose() * temp)`, `g.head<3>() += temp`,
 * `R += p_base_src * temp.transpose()`; PCL registration/impl/gicp.hpp).  The two differ in the last
 * bit of f and the gradient; with PCL's BFGS tolerance of 1e-2 that can move the stopping point: the
 * spread between the modes is what separates ANY faithful implementation from PCL itself
 * (tests/test_gicp_gpu.py::test_gicp_spread_against_pcl_literal_summation). */
static int g_gicp_summation = 0;
void wmo_gicp_set_summation(int mode) { g_gicp_summation = mode ? 1 : 0; }
int wmo_gicp_get_summation(void) { return g_gicp_summation; }

static void dd_add(double *hi, double *lo, double x) {
    if (g_gicp_summation) {  /* PCL-literal: one rounding per addition, in index order */
        *hi += x;
        return;
    }
    const double s = *hi + x;
    const double bb = s - *hi;
    *lo += (*hi - (s - bb)) + (x - bb);
    *hi = s;
}

double wmo_gicp_fdf(const float *src, const float *tgt, const int *src_idx, const int *tgt_idx,
                    const double *mahal, int m, const double base[16], const double x[6],
                    double g[6]) {
    float T[16], Bf[16];
    double hi[13] = {0}, lo[13] = {0};
    double f, gt[3], Racc[9];
    int i, a, b;
    state_to_matrix_f(base, x, T);
    for (i = 0; i < 16; ++i) Bf[i] = (float) base[i];
    for (i = 0; i < m; ++i) {
        const float *ps = src + 3 * src_idx[i], *pt = tgt + 3 * tgt_idx[i];
        const double *M = mahal + 9 * src_idx[i];
        float pp[3], pb[3];
        double res[3], temp[3];
        mul_pt_f(T, ps, pp);
        res[0] = pp[0] - pt[0]; /* float subtraction, then widened */
        res[1] = pp[1] - pt[1];
        res[2] = pp[2] - pt[2];
        for (a = 0; a < 3; ++a) temp[a] = M[a * 3] * res[0] + M[a * 3 + 1] * res[1] + M[a * 3 + 2] * res[2];
        dd_add(&hi[0], &lo[0], res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2]);
        mul_pt_f(Bf, ps, pb);
        for (a = 0; a < 3; ++a) {
            dd_add(&hi[1 + a], &lo[1 + a], temp[a]);
            for (b = 0; b < 3; ++b) dd_add(&hi[4 + a * 3 + b], &lo[4 + a * 3 + b], (double) pb[a] * temp[b]);
        }
    }
    f = hi[0] + lo[0];
    for (a = 0; a < 3; ++a) gt[a] = hi[1 + a] + lo[1 + a];
    for (a = 0; a < 9; ++a) Racc[a] = hi[4 + a] + lo[4 + a];
    if (g) {
        for (a = 0; a < 3; ++a) g[a] = gt[a] * 2.0 / m;
        for (a = 0; a < 9; ++a) Racc[a] = Racc[a] * 2.0 / m;
        r_derivative(x, Racc, g);
    }
    return f / m;
}

/* ------------------------------------------------------------------ the objective as sufficient statistics
 * Objective mode 1 (wmo_gicp_set_objective): NOT PCL's arithmetic but the HIP path's default
 * (libwave_amd/csrc/wm_gicp_quad.hpp -- read its header for the algebra): between two correspondence searches the
 * pairs and their Mahalanobis matrices are fixed and the residual is affine in the transform's entries, so
 *     sum r^T M r = C0 + sum_aj D_aj (B0_aj + G_aj),   G_aj = B0_aj + sum_ck D_ck A_(ac)(jk),   D = T(x) - T0,
 * with 74 sums over the pairs formed once per outer iteration (A: 60, B0: 12, C0, the count) around T0, the
 * transform the pairs were found under.  r0 = T0 p - q is PCL's float residual; away from T0 the per-point float
 * rounding of PCL's transform (a relative ~4e-7 of f) is absent.  The oracle restates it so that the HIP path can
 * be held to it bit for bit (same terms, double-double sums, same evaluation order), while mode 0 -- PCL's per-pair
 * sums -- says how far this objective's registrations are from PCL's (tests/test_gicp_quad_gpu.py, test_oracle_cpu.py). */
static int g_gicp_objective = 0;
void wmo_gicp_set_objective(int mode) { g_gicp_objective = mode ? 1 : 0; }
int wmo_gicp_get_objective(void) { return g_gicp_objective; }

enum { QUAD_N = 74, QUAD_OFF_B = 60, QUAD_OFF_C = 72, QUAD_OFF_COUNT = 73 };
static int quad_s6(int a, int c) {
    const int lo = a < c ? a : c, hi = a < c ? c : a;
    return lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
}
static int quad_s10(int j, int k) {
    const int lo = j < k ? j : k, hi = j < k ? k : j;
    return lo == 0 ? hi : (lo == 1 ? 3 + hi : (lo == 2 ? 5 + hi : 9));
}
/* the 74 sums of the pairs (si, ti, mahal) found under the float transform T0 (always double-double: one rounding) */
static void quad_statistics(const float *src, const float *tgt, const int *src_idx, const int *tgt_idx,
                            const double *mahal, int m, const float T0[16], double Q[QUAD_N]) {
    double hi[QUAD_N] = {0}, lo[QUAD_N] = {0};
    const int keep = g_gicp_summation;
    int i, a, j, k, s, t;
    g_gicp_summation = 0;
    for (i = 0; i < m; ++i) {
        const float *ps = src + 3 * src_idx[i], *pt = tgt + 3 * tgt_idx[i];
        const double *M = mahal + 9 * src_idx[i];
        float pp[3];
        double r0[3], Ms[6], z[4], zz[10], t0[3];
        mul_pt_f(T0, ps, pp);
        r0[0] = pp[0] - pt[0]; /* float subtraction, then widened */
        r0[1] = pp[1] - pt[1];
        r0[2] = pp[2] - pt[2];
        Ms[0] = M[0];
        Ms[1] = 0.5 * (M[1] + M[3]);
        Ms[2] = 0.5 * (M[2] + M[6]);
        Ms[3] = M[4];
        Ms[4] = 0.5 * (M[5] + M[7]);
        Ms[5] = M[8];
        z[0] = ps[0];
        z[1] = ps[1];
        z[2] = ps[2];
        z[3] = 1.0;
        for (j = 0; j < 4; ++j)
            for (k = j; k < 4; ++k) zz[quad_s10(j, k)] = z[j] * z[k];
        for (s = 0; s < 6; ++s)
            for (t = 0; t < 10; ++t) dd_add(&hi[s * 10 + t], &lo[s * 10 + t], Ms[s] * zz[t]);
        for (a = 0; a < 3; ++a) t0[a] = (Ms[quad_s6(a, 0)] * r0[0] + Ms[quad_s6(a, 1)] * r0[1]) + Ms[quad_s6(a, 2)] * r0[2];
        for (a = 0; a < 3; ++a)
            for (j = 0; j < 4; ++j) dd_add(&hi[QUAD_OFF_B + a * 4 + j], &lo[QUAD_OFF_B + a * 4 + j], t0[a] * z[j]);
        dd_add(&hi[QUAD_OFF_C], &lo[QUAD_OFF_C], (r0[0] * t0[0] + r0[1] * t0[1]) + r0[2] * t0[2]);
        dd_add(&hi[QUAD_OFF_COUNT], &lo[QUAD_OFF_COUNT], 1.0);
    }
    g_gicp_summation = keep;
    for (i = 0; i < QUAD_N; ++i) Q[i] = hi[i] + lo[i];
}
/* f and gradient at x from the statistics (gicp_quad_eval of wm_gicp_quad.hpp, operation for operation) */
static double quad_eval(const double Q[QUAD_N], const float T0[16], const double base[16], const double x[6], double g[6]) {
    float T[16];
    double D[12], G[12], fm, m;
    int a, b, c, j, k;
    state_to_matrix_f(base, x, T);
    for (k = 0; k < 12; ++k) D[k] = (double) T[k] - (double) T0[k];
    for (a = 0; a < 3; ++a)
        for (j = 0; j < 4; ++j) {
            double s = Q[QUAD_OFF_B + a * 4 + j];
            for (c = 0; c < 3; ++c)
                for (k = 0; k < 4; ++k) s += D[c * 4 + k] * Q[quad_s6(a, c) * 10 + quad_s10(j, k)];
            G[a * 4 + j] = s;
        }
    fm = Q[QUAD_OFF_C];
    for (k = 0; k < 12; ++k) fm += D[k] * (Q[QUAD_OFF_B + k] + G[k]);
    m = Q[QUAD_OFF_COUNT];
    if (g) {
        double Racc[9];
        for (a = 0; a < 3; ++a) g[a] = G[a * 4 + 3] * 2.0 / m;
        for (a = 0; a < 3; ++a)
            for (b = 0; b < 3; ++b) {
                double s = 0.0;
                for (j = 0; j < 4; ++j) s += base[a * 4 + j] * G[b * 4 + j];
                Racc[a * 3 + b] = s * 2.0 / m;
            }
        r_derivative(x, Racc, g);
    }
    return fm / m;
}
/* test hook: the statistics objective evaluated once (pairs found under T0, state x) */
double wmo_gicp_fdf_statistics(const float *src, const float *tgt, const int *src_idx, const int *tgt_idx,
                               const double *mahal, int m, const double base[16], const float T0[16],
                               const double x[6], double g[6], double Q_out[74]) {
    double Q[QUAD_N];
    quad_statistics(src, tgt, src_idx, tgt_idx, mahal, m, T0, Q);
    if (Q_out) memcpy(Q_out, Q, sizeof(Q));
    return quad_eval(Q, T0, base, x, g);
}

/* ------------------------------------------------------------------ BFGS
 * pcl::BFGS (registration/bfgs.h), itself GSL's vector_bfgs2: a memoryless BFGS
 * direction update with Fletcher's bracketing / sectioning line search. */
typedef struct {
    const float *src, *tgt;
    const int *si, *ti;
    const double *mahal;
    int m;
    const double *base;
    int evals;
    int quad;            /* objective mode 1: evaluate from Q */
    double Q[QUAD_N];
    float T0[16];
} gicp_fn;

static double fn_fdf(gicp_fn *F, const double x[6], double g[6]) {
    double f;
    const char *path = getenv("WMO_GICP_TRACE"); /* developer: every evaluation, in hex floats */
    F->evals++;
    f = F->quad ? quad_eval(F->Q, F->T0, F->base, x, g)
                : wmo_gicp_fdf(F->src, F->tgt, F->si, F->ti, F->mahal, F->m, F->base, x, g);
    if (path) {
        FILE *fp = fopen(path, "a");
        if (fp) {
            int k;
            fprintf(fp, "%d", F->m);
            for (k = 0; k < 6; ++k) fprintf(fp, " %a", x[k]);
            fprintf(fp, " | %a |", f);
            if (g) for (k = 0; k < 6; ++k) fprintf(fp, " %a", g[k]);
            fprintf(fp, "\n");
            fclose(fp);
        }
    }
    return f;
}

typedef struct {
    gicp_fn *F;
    double x0[6], g0[6], p[6];
    double f0, df0;          /* at alpha = 0 */
    double x_a[6], g_a[6];   /* cache at alpha */
    double alpha_c, f_c, df_c;
    int have_c;
} line_fn;

static void line_eval(line_fn *L, double alpha) {
    int i;
    if (L->have_c && alpha == L->alpha_c) return;
    for (i = 0; i < 6; ++i) L->x_a[i] = L->x0[i] + alpha * L->p[i];
    L->f_c = fn_fdf(L->F, L->x_a, L->g_a);
    L->df_c = 0;
    for (i = 0; i < 6; ++i) L->df_c += L->g_a[i] * L->p[i];
    L->alpha_c = alpha;
    L->have_c = 1;
}

static double cubic(double c0, double c1, double c2, double c3, double z) {
    return c0 + z * (c1 + z * (c2 + z * c3));
}
static void check_extremum(double c0, double c1, double c2, double c3, double z, double *zmin,
                           double *fmin) {
    double y = cubic(c0, c1, c2, c3, z);
    if (y < *fmin) {
        *zmin = z;
        *fmin = y;
    }
}
static int solve_quadratic(double a, double b, double c, double *x0, double *x1) {
    double disc;
    if (a == 0) {
        if (b == 0) return 0;
        *x0 = -c / b;
        return 1;
    }
    disc = b * b - 4 * a * c;
    if (disc > 0) {
        if (b == 0) {
            double r = sqrt(-c / a);
            *x0 = -r;
            *x1 = r;
        } else {
            double sgnb = (b > 0 ? 1 : -1);
            double temp = -0.5 * (b + sgnb * sqrt(disc));
            double r1 = temp / a, r2 = c / temp;
            if (r1 < r2) {
                *x0 = r1;
                *x1 = r2;
            } else {
                *x0 = r2;
                *x1 = r1;
            }
        }
        return 2;
    } else if (disc == 0) {
        *x0 = -0.5 * b / a;
        *x1 = -0.5 * b / a;
        return 2;
    }
    return 0;
}
static double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
    double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
    double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
    double c = 2 * (f1 - f0 - fp0);
    double zmin = zl, fmin = fl;
    if (fh < fmin) {
        zmin = zh;
        fmin = fh;
    }
    if (c > 0) {
        double z = -fp0 / c;
        if (z > zl && z < zh) {
            double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
            if (f < fmin) {
                zmin = z;
                fmin = f;
            }
        }
    }
    return zmin;
}
static double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
    double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
    double xi = fp0 + fp1 - 2 * (f1 - f0);
    double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
    double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0, z1;
    int n;
    check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
    n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
    if (n == 2) {
        if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
        if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
    } else if (n == 1) {
        if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
    }
    return zmin;
}
static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb,
                          double xmin, double xmax, int order) {
    double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
    if (ymin > ymax) {
        double t = ymin;
        ymin = ymax;
        ymax = t;
    }
    if (order > 2 && fpb == fpb) /* !isnan */
        y = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax);
    else
        y = interp_quad(fa, fpa * (b - a), fb, ymin, ymax);
    return a + y * (b - a);
}

enum { LS_SUCCESS = 0, LS_NOPROG = 1 };

static int line_search(line_fn *L, double rho, double sigma, double tau1, double tau2, double tau3,
                       int order, double alpha1, double *alpha_new) {
    double f0 = L->f0, fp0 = L->df0, falpha, falpha_prev = f0, fpalpha, fpalpha_prev = fp0, delta,
           alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a = 0.0, b = alpha, fa = f0, fb = 0.0, fpa = fp0, fpb = 0.0;
    const int bracket_iters = 100, section_iters = 100;
    int i = 0;
    while (i++ < bracket_iters) {
        line_eval(L, alpha);
        falpha = L->f_c;
        if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
            a = alpha_prev;
            fa = falpha_prev;
            fpa = fpalpha_prev;
            b = alpha;
            fb = falpha;
            fpb = NAN;
            break;
        }
        fpalpha = L->df_c;
        if (fabs(fpalpha) <= -sigma * fp0) {
            *alpha_new = alpha;
            return LS_SUCCESS;
        }
        if (fpalpha >= 0) {
            a = alpha;
            fa = falpha;
            fpa = fpalpha;
            b = alpha_prev;
            fb = falpha_prev;
            fpb = fpalpha_prev;
            break;
        }
        delta = alpha - alpha_prev;
        alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha,
                                 alpha + delta, alpha + tau1 * delta, order);
        alpha_prev = alpha;
        falpha_prev = falpha;
        fpalpha_prev = fpalpha;
        alpha = alpha_next;
    }
    while (i++ < section_iters) {
        delta = b - a;
        alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
        line_eval(L, alpha);
        falpha = L->f_c;
        if ((a - alpha) * fpa <= DBL_EPSILON) return LS_NOPROG; /* roundoff prevents progress */
        if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
            b = alpha;
            fb = falpha;
            fpb = NAN;
        } else {
            fpalpha = L->df_c;
            if (fabs(fpalpha) <= -sigma * fp0) {
                *alpha_new = alpha;
                return LS_SUCCESS;
            }
            if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
                b = a;
                fb = fa;
                fpb = fpa;
                a = alpha;
                fa = falpha;
                fpa = fpalpha;
            } else {
                a = alpha;
                fa = falpha;
                fpa = fpalpha;
            }
        }
    }
    *alpha_new = alpha;
    return LS_SUCCESS;
}

static double norm6(const double *v) {
    double s = 0;
    int i;
    for (i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}

/* estimateRigidTransformationBFGS.  x in/out; returns number of inner iterations, or -1
 * when fewer than 4 pairs (NotEnoughPointsException) */
static int bfgs_minimize(gicp_fn *F, double x[6], int max_inner, double *f_out) {
    const double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, gradient_tol = 1e-2;
    const int order = 3;
    double f, g[6], x0[6], g0[6], p[6], dx0[6], dg0[6];
    double g0norm, pnorm, fp0, delta_f = 0, step = 1.0; /* parameters.step_size = 1 */
    int inner = 0, i;
    if (F->m < 4) return -1;
    /* minimizeInit */
    f = fn_fdf(F, x, g);
    memcpy(x0, x, sizeof(x0));
    memcpy(g0, g, sizeof(g0));
    g0norm = norm6(g0);
    /* Objective mode 1 only (the statistics objective, wm_bfgs.hpp does the same for it): the minimiser's own
     * success test, |g| < gradient_tol, is made at the starting point too.  pcl::BFGS (GSL's vector_bfgs2) tests
     * only AFTER a step; with PCL's per-pair objective a step from an already converged point dies in the float
     * dust of the objective (NoProgress: x unchanged, the outer loop sees no change and stops).  The statistics
     * objective has no dust: the step succeeds, moves x by ~1e-6, the float transform changes in its last bit, and
     * the outer loop (r_eps = 1e-8) re-pairs and crawls on for dozens of iterations that change nothing. */
    if (F->quad && g0norm < gradient_tol) {
        if (f_out) *f_out = f;
        return 0;
    }
    for (i = 0; i < 6; ++i) p[i] = -g0[i] / g0norm;
    pnorm = norm6(p);
    fp0 = -g0norm;
    do {
        line_fn L;
        double alpha = 0, alpha1, f_prev = f;
        int status;
        ++inner;
        /* minimizeOneStep */
        if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0 || pnorm != pnorm || g0norm != g0norm) break; /* NoProgress */
        if (delta_f < 0) {
            double del = fmax(-delta_f, 10 * DBL_EPSILON * fabs(f_prev));
            alpha1 = fmin(1.0, 2.0 * del / (-fp0));
        } else {
            alpha1 = fabs(step);
        }
        L.F = F;
        memcpy(L.x0, x0, sizeof(x0));
        memcpy(L.g0, g0, sizeof(g0));
        memcpy(L.p, p, sizeof(p));
        L.f0 = f_prev;
        L.df0 = fp0;
        L.have_c = 0;
        status = line_search(&L, rho, sigma, tau1, tau2, tau3, order, alpha1, &alpha);
        if (status != LS_SUCCESS) break; /* NoProgress: x keeps the last accepted step */
        line_eval(&L, alpha);            /* update_position */
        memcpy(x, L.x_a, sizeof(L.x_a));
        memcpy(g, L.g_a, sizeof(L.g_a));
        f = L.f_c;
        delta_f = f - f_prev;
        {
            double dxg = 0, dgg = 0, dxdg = 0, dgnorm, A, B, pg = 0, dir;
            for (i = 0; i < 6; ++i) {
                dx0[i] = x[i] - x0[i];
                dg0[i] = g[i] - g0[i];
            }
            for (i = 0; i < 6; ++i) {
                dxg += dx0[i] * g[i];
                dgg += dg0[i] * g[i];
                dxdg += dx0[i] * dg0[i];
            }
            dgnorm = norm6(dg0);
            if (dxdg != 0) {
                B = dxg / dxdg;
                A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
            } else {
                B = 0;
                A = 0;
            }
            for (i = 0; i < 6; ++i) p[i] = g[i] - A * dx0[i] - B * dg0[i];
            memcpy(g0, g, sizeof(g0));
            memcpy(x0, x, sizeof(x0));
            g0norm = norm6(g0);
            pnorm = norm6(p);
            for (i = 0; i < 6; ++i) pg += p[i] * g0[i];
            dir = (pg >= 0) ? -1.0 : +1.0;
            for (i = 0; i < 6; ++i) p[i] *= dir / pnorm;
            pnorm = norm6(p);
            fp0 = 0;
            for (i = 0; i < 6; ++i) fp0 += p[i] * g0[i];
        }
        if (norm6(g) < gradient_tol) break; /* testGradient -> Success */
    } while (inner < max_inner);
    if (f_out) *f_out = f;
    return inner;
}

/* Eigen's fixed-size 3x3 inverse (Eigen/src/LU/InverseImpl.h: cofactors times 1 / determinant),
 * which PCL's `(C2 + R C1 R^T).inverse()` resolves to */
static void inv3_cofactor(const double *m, double *o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
                 c02 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

int wmo_gicp_align(const float *src, int n, const float *tgt, int m, const wmo_gicp_params *prm,
                   double T_out[16], wmo_gicp_result *res) {
    double *C1 = (double *) malloc(sizeof(double) * 9 * (n > 0 ? n : 1));
    double *C2 = (double *) malloc(sizeof(double) * 9 * (m > 0 ? m : 1));
    double *mahal = (double *) malloc(sizeof(double) * 9 * (n > 0 ? n : 1));
    int *si = (int *) malloc(sizeof(int) * (n > 0 ? n : 1));
    int *ti = (int *) malloc(sizeof(int) * (n > 0 ? n : 1));
    wmo_kdtree *tree;
    float T[16], prevT[16];
    double base[16], f_last = 0;
    int iter = 0, converged = 0, i, a, b, c, cnt = 0, inner_total = 0, rc = 1, evals_total = 0;
    const double dist_thr = prm->max_corr * prm->max_corr;
    const int max_it = prm->force_iterations > 0 ? prm->force_iterations : prm->max_iter;
    for (i = 0; i < 16; ++i) T[i] = (i % 5 == 0);
    memcpy(prevT, T, sizeof(T));
    wmo_mat4_identity(base);
    if (wmo_gicp_covariances(tgt, m, prm->corr_rand, prm->gicp_epsilon, C2) ||
        wmo_gicp_covariances(src, n, prm->corr_rand, prm->gicp_epsilon, C1)) {
        goto done; /* PCL logs an error; align leaves converged_ = false */
    }
    tree = wmo_kdtree_build(tgt, m);
    while (!converged) {
        double R[9], x[6], delta = 0;
        gicp_fn F;
        int inner;
        cnt = 0;
        for (a = 0; a < 3; ++a)
            for (b = 0; b < 3; ++b) R[a * 3 + b] = (double) T[a * 4 + b]; /* transformation_ * guess(I) */
        for (i = 0; i < n; ++i) {
            float q[3], d2;
            int j;
            mul_pt_f(T, src + 3 * i, q);
            j = wmo_kdtree_nn(tree, q, &d2);
            if (j >= 0 && (double) d2 < dist_thr) {
                double Mx[9], tmp[9];
                const double *c1 = C1 + 9 * i, *c2 = C2 + 9 * j;
                for (a = 0; a < 3; ++a)
                    for (b = 0; b < 3; ++b) {
                        double s = 0;
                        for (c = 0; c < 3; ++c) s += R[a * 3 + c] * c1[c * 3 + b];
                        Mx[a * 3 + b] = s; /* M = R*C1 */
                    }
                for (a = 0; a < 3; ++a)
                    for (b = 0; b < 3; ++b) {
                        double s = 0;
                        for (c = 0; c < 3; ++c) s += Mx[a * 3 + c] * R[b * 3 + c];
                        tmp[a * 3 + b] = s + c2[a * 3 + b]; /* R*C1*R' + C2 */
                    }
                inv3_cofactor(tmp, mahal + 9 * i);
                si[cnt] = i;
                ti[cnt] = j;
                ++cnt;
            }
        }
        memcpy(prevT, T, sizeof(T));
        /* estimateRigidTransformationBFGS(output, source_indices, target, ...) */
        x[0] = T[3];
        x[1] = T[7];
        x[2] = T[11];
        /* PCL (C++, <cmath>): unqualified atan2 / asin on Eigen::Matrix4f entries resolve to the
         * FLOAT overloads; the results are then widened into the Vector6d */
        x[3] = (double) atan2f(T[9], T[10]);
        x[4] = (double) asinf(-T[8]);
        x[5] = (double) atan2f(T[4], T[0]);
        F.src = src;
        F.tgt = tgt;
        F.si = si;
        F.ti = ti;
        F.mahal = mahal;
        F.m = cnt;
        F.base = base;
        F.evals = 0;
        F.quad = g_gicp_objective;
        if (F.quad && cnt > 0) {
            memcpy(F.T0, T, sizeof(T));
            quad_statistics(src, tgt, si, ti, mahal, cnt, F.T0, F.Q);
        }
        inner = bfgs_minimize(&F, x, prm->max_inner, &f_last);
        evals_total += F.evals;
        if (inner < 0) break; /* exception -> the loop breaks, converged_ stays false */
        inner_total += inner;
        state_to_matrix_f(base, x, T); /* transformation_matrix.setIdentity(); applyState */
        for (a = 0; a < 4; ++a)
            for (b = 0; b < 4; ++b) {
                double ratio = (a < 3 && b < 3) ? 1.0 / prm->r_eps : 1.0 / prm->t_eps;
                double cd = ratio * fabs((double) prevT[a * 4 + b] - (double) T[a * 4 + b]);
                if (cd > delta) delta = cd;
            }
        ++iter;
        if (prm->force_iterations > 0) {
            if (iter >= max_it) {
                converged = 1;
                memcpy(prevT, T, sizeof(T));
            }
        } else if (iter >= max_it || delta < 1) {
            converged = 1;
            memcpy(prevT, T, sizeof(T));
        }
    }
    wmo_kdtree_free(tree);
    rc = converged ? 0 : 1;
done:
    /* final = previous_transformation_ (guess = identity) */
    for (i = 0; i < 16; ++i) T_out[i] = (double) prevT[i];
    if (res) {
        res->converged = converged;
        res->iterations = iter;
        res->n_corr = cnt;
        res->inner_total = inner_total;
        res->f_final = f_last;
        res->evaluations = evals_total;
        res->reserved = 0;
    }
    free(C1);
    free(C2);
    free(mahal);
    free(si);
    free(ti);
    return rc;
}
