/*
 * ndt.c -- ORACLE (test infrastructure): CPU restatement of
 *   pcl::NormalDistributionsTransform<PointXYZ,PointXYZ>::align   (PCL 1.8)
 *   pcl::VoxelGridCovariance<PointXYZ>::filter (the target voxel statistics)
 * as configured and driven by libwave's NDTMatcher:
 *   wave_matching/src/ndt.cpp:18-34  (setTransformationEpsilon / setStepSize /
 *                                     setResolution / setMaximumIterations)
 *   wave_matching/src/ndt.cpp:48-65  (setInputSource / setInputTarget / align)
 * PCL is an un-vendored dependency; the algorithm follows its published sources
 *   registration/impl/ndt.hpp (computeTransformation, computeDerivatives,
 *     updateDerivatives, computeAngleDerivatives, computePointDerivatives,
 *     computeHessian, computeStepLengthMT, trialValueSelectionMT, updateIntervalMT)
 *   filters/impl/voxel_grid_covariance.hpp (applyFilter)
 * and Magnusson 2009 / More-Thuente 1994 which they implement.
 * Version switches:
 *   skip_line_search  PCL 1.8.x initialises `interval_converged` to true so the
 *                     More-Thuente loop never runs; default here = run it.
 *                     With the reference's own test parameters (tests/config/ndt.yaml: step_size 3,
 *                     max_iter 100, t_eps 1e-8) the 1.8-literal mode DOES reproduce the outcome of
 *                     wave_matching/tests/ndt_tests.cpp:85-102 (green on the reference's CI with
 *                     libpcl1.8): 102 iterations -- PCL's `nr_iterations_ > max_iterations_` rule,
 *                     which is also what makes hasConverged() true --, final |T - T_gt|_F = 0.0099
 *                     < 0.12.  The undamped Newton iteration leaves the basin after three steps
 *                     (0.197 m, then 0.4-1.0 m for some eighty iterations) and returns to it near
 *                     iteration 90; the trajectory is chaotic (pcl_d1_sign = 0 alone ends it 4.4 m
 *                     away), so real PCL's digits cannot be promised, only the test's outcome
 *                     (tests/test_oracle_cpu.py::test_ndt_pcl18_literal_mode_meets_the_reference_test).
 *   pcl_d1_sign       PCL's h_ang_d1 third component is +sy (thesis typo; the true
 *                     derivative is -sy); default = PCL's.
 * PARITY: unpinned (no PCL to run); pinned to the reference tests' assertions
 * (wave_matching/tests/ndt_tests.cpp:37,58-60) and finite-difference checks.
 */
#include "wm_oracle.h"
#include "wmo_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ voxel grid */
typedef struct {
    int i, j, k;
    int n;
    double mean[3];
    double cov[9];
    double icov[9];
    int valid;
} ndt_voxel;

struct wmo_ndt_grid {
    double res;
    int nvox;       /* all occupied voxels, sorted by (k, j, i) */
    int nvalid;
    ndt_voxel *vox;
    /* hash of ijk -> voxel index */
    int hcap;
    int *htab;
};

static unsigned long long ijk_key(int i, int j, int k) {
    return ((unsigned long long) (unsigned) (i + (1 << 20)) << 42) |
           ((unsigned long long) (unsigned) (j + (1 << 20)) << 21) |
           (unsigned long long) (unsigned) (k + (1 << 20));
}

static unsigned hash64(unsigned long long x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (unsigned) x;
}

static int grid_find(const wmo_ndt_grid *g, int i, int j, int k) {
    unsigned h = hash64(ijk_key(i, j, k)) & (unsigned) (g->hcap - 1);
    for (;;) {
        int v = g->htab[h];
        if (v < 0) return -1;
        if (g->vox[v].i == i && g->vox[v].j == j && g->vox[v].k == k) return v;
        h = (h + 1) & (unsigned) (g->hcap - 1);
    }
}

typedef struct {
    int i, j, k, pt;
} vk;

static int vk_cmp(const void *a, const void *b) {
    const vk *x = (const vk *) a, *y = (const vk *) b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    if (x->j != y->j) return x->j < y->j ? -1 : 1;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt);
}

wmo_ndt_grid *wmo_ndt_grid_build(const float *tgt, int m, double res) {
    wmo_ndt_grid *g = (wmo_ndt_grid *) calloc(1, sizeof(wmo_ndt_grid));
    vk *keys = (vk *) malloc(sizeof(vk) * (m > 0 ? m : 1));
    const float inv = 1.0f / (float) res; /* inverse_leaf_size_ is float in PCL */
    int n = 0, a, i, b;
    g->res = res;
    for (i = 0; i < m; ++i) {
        const float *p = tgt + 3 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        /* floor(p * inverse_leaf) - min_b: the absolute leaf coordinate decides
         * membership, min_b only shifts the linear index */
        keys[n].i = (int) floorf(p[0] * inv);
        keys[n].j = (int) floorf(p[1] * inv);
        keys[n].k = (int) floorf(p[2] * inv);
        keys[n].pt = i;
        ++n;
    }
    qsort(keys, n, sizeof(vk), vk_cmp);
    g->vox = (ndt_voxel *) calloc(n > 0 ? n : 1, sizeof(ndt_voxel));
    i = 0;
    while (i < n) {
        int j = i;
        ndt_voxel *v = &g->vox[g->nvox];
        double sum[3] = {0, 0, 0}, spp[9] = {0};
        v->i = keys[i].i;
        v->j = keys[i].j;
        v->k = keys[i].k;
        while (j < n && keys[j].i == v->i && keys[j].j == v->j && keys[j].k == v->k) {
            const float *p = tgt + 3 * keys[j].pt;
            double d[3] = {p[0], p[1], p[2]};
            for (a = 0; a < 3; ++a) {
                sum[a] += d[a];
                for (b = 0; b < 3; ++b) spp[a * 3 + b] += d[a] * d[b];
            }
            ++j;
        }
        v->n = j - i;
        for (a = 0; a < 3; ++a) v->mean[a] = sum[a] / v->n;
        v->valid = 0;
        if (v->n >= 6) { /* min_points_per_voxel_ = 6 */
            double evals[3], evecs[9], nn = (double) v->n;
            for (a = 0; a < 3; ++a)
                for (b = 0; b < 3; ++b)
                    v->cov[a * 3 + b] = (spp[a * 3 + b] - 2.0 * (sum[a] * v->mean[b])) / nn +
                                        v->mean[a] * v->mean[b];
            for (a = 0; a < 9; ++a) v->cov[a] *= (nn - 1.0) / nn;
            wmo_sym_eig(3, v->cov, evals, evecs); /* ascending, like SelfAdjointEigenSolver */
            if (!(evals[0] < 0 || evals[1] < 0 || evals[2] <= 0)) {
                const double minv = 0.01 * evals[2]; /* min_covar_eigvalue_mult_ */
                int ok = 1;
                if (evals[0] < minv) {
                    double D[9] = {0}, t[9], einv[9];
                    evals[0] = minv;
                    if (evals[1] < minv) evals[1] = minv;
                    D[0] = evals[0];
                    D[4] = evals[1];
                    D[8] = evals[2];
                    wmo_inverse(3, evecs, einv);
                    wmo_mat_mul(3, evecs, D, t);
                    wmo_mat_mul(3, t, einv, v->cov);
                }
                wmo_inverse(3, v->cov, v->icov);
                for (a = 0; a < 9; ++a)
                    if (!isfinite(v->icov[a])) ok = 0;
                v->valid = ok;
            }
        }
        g->nvalid += v->valid;
        ++g->nvox;
        i = j;
    }
    free(keys);
    g->hcap = 16;
    while (g->hcap < 2 * g->nvox + 2) g->hcap <<= 1;
    g->htab = (int *) malloc(sizeof(int) * g->hcap);
    for (i = 0; i < g->hcap; ++i) g->htab[i] = -1;
    for (i = 0; i < g->nvox; ++i) {
        unsigned h = hash64(ijk_key(g->vox[i].i, g->vox[i].j, g->vox[i].k)) & (unsigned) (g->hcap - 1);
        while (g->htab[h] >= 0) h = (h + 1) & (unsigned) (g->hcap - 1);
        g->htab[h] = i;
    }
    return g;
}

void wmo_ndt_grid_free(wmo_ndt_grid *g) {
    if (!g) return;
    free(g->vox);
    free(g->htab);
    free(g);
}

int wmo_ndt_grid_size(const wmo_ndt_grid *g) { return g->nvalid; }

void wmo_ndt_grid_export(const wmo_ndt_grid *g, int *ijk, double *mean, double *icov, int *count) {
    int i, o = 0;
    for (i = 0; i < g->nvox; ++i) {
        const ndt_voxel *v = &g->vox[i];
        if (!v->valid) continue;
        ijk[3 * o] = v->i;
        ijk[3 * o + 1] = v->j;
        ijk[3 * o + 2] = v->k;
        memcpy(mean + 3 * o, v->mean, sizeof(v->mean));
        memcpy(icov + 9 * o, v->icov, sizeof(v->icov));
        count[o] = v->n;
        ++o;
    }
}

/* ------------------------------------------------------- pose <-> matrix */
/* Translation(p0..2) * Rx(p3) * Ry(p4) * Rz(p5), stored float like PCL's
 * final_transformation_ (Eigen::Matrix4f) */
static void pose_to_matrix_f(const double p[6], float T[16]) {
    float cx = cosf((float) p[3]), sx = sinf((float) p[3]);
    float cy = cosf((float) p[4]), sy = sinf((float) p[4]);
    float cz = cosf((float) p[5]), sz = sinf((float) p[5]);
    /* Rx*Ry*Rz */
    T[0] = cy * cz;
    T[1] = -cy * sz;
    T[2] = sy;
    T[4] = cx * sz + sx * sy * cz;
    T[5] = cx * cz - sx * sy * sz;
    T[6] = -sx * cy;
    T[8] = sx * sz - cx * sy * cz;
    T[9] = sx * cz + cx * sy * sz;
    T[10] = cx * cy;
    T[3] = (float) p[0];
    T[7] = (float) p[1];
    T[11] = (float) p[2];
    T[12] = T[13] = T[14] = 0;
    T[15] = 1;
}

typedef struct {
    double ja[3], jb[3], jc[3], jd[3], je[3], jf[3], jg[3], jh[3];
    double ha2[3], ha3[3], hb2[3], hb3[3], hc2[3], hc3[3], hd1[3], hd2[3], hd3[3], he1[3], he2[3],
        he3[3], hf1[3], hf2[3], hf3[3];
} angle_derivs;

static void compute_angle_derivatives(const double p[6], int pcl_d1_sign, angle_derivs *A) {
    double cx, cy, cz, sx, sy, sz;
    if (fabs(p[3]) < 10e-5) {
        cx = 1.0;
        sx = 0.0;
    } else {
        cx = cos(p[3]);
        sx = sin(p[3]);
    }
    if (fabs(p[4]) < 10e-5) {
        cy = 1.0;
        sy = 0.0;
    } else {
        cy = cos(p[4]);
        sy = sin(p[4]);
    }
    if (fabs(p[5]) < 10e-5) {
        cz = 1.0;
        sz = 0.0;
    } else {
        cz = cos(p[5]);
        sz = sin(p[5]);
    }
#define SET3(v, a, b, c) \
    do {                 \
        (v)[0] = (a);    \
        (v)[1] = (b);    \
        (v)[2] = (c);    \
    } while (0)
    SET3(A->ja, -sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy);
    SET3(A->jb, cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy);
    SET3(A->jc, -sy * cz, sy * sz, cy);
    SET3(A->jd, sx * cy * cz, -sx * cy * sz, sx * sy);
    SET3(A->je, -cx * cy * cz, cx * cy * sz, -cx * sy);
    SET3(A->jf, -cy * sz, -cy * cz, 0);
    SET3(A->jg, cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0);
    SET3(A->jh, sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0);
    SET3(A->ha2, -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy);
    SET3(A->ha3, -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy);
    SET3(A->hb2, cx * cy * cz, -cx * cy * sz, cx * sy);
    SET3(A->hb3, sx * cy * cz, -sx * cy * sz, sx * sy);
    SET3(A->hc2, -sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0);
    SET3(A->hc3, cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0);
    SET3(A->hd1, -cy * cz, cy * sz, pcl_d1_sign ? sy : -sy);
    SET3(A->hd2, -sx * sy * cz, sx * sy * sz, sx * cy);
    SET3(A->hd3, cx * sy * cz, -cx * sy * sz, -cx * cy);
    SET3(A->he1, sy * sz, sy * cz, 0);
    SET3(A->he2, -sx * cy * sz, -sx * cy * cz, 0);
    SET3(A->he3, cx * cy * sz, cx * cy * cz, 0);
    SET3(A->hf1, -cy * cz, cy * sz, 0);
    SET3(A->hf2, -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0);
    SET3(A->hf3, -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0);
#undef SET3
}

static double dot3(const double *a, const double *b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* J (3x6, row-major) and the second-derivative 3-vectors PH[i][j] (i,j in 3..5) */
static void compute_point_derivatives(const angle_derivs *A, const double x[3], double J[18],
                                      double PH[3][3][3]) {
    int a;
    memset(J, 0, sizeof(double) * 18);
    J[0 * 6 + 0] = J[1 * 6 + 1] = J[2 * 6 + 2] = 1.0;
    J[1 * 6 + 3] = dot3(x, A->ja);
    J[2 * 6 + 3] = dot3(x, A->jb);
    J[0 * 6 + 4] = dot3(x, A->jc);
    J[1 * 6 + 4] = dot3(x, A->jd);
    J[2 * 6 + 4] = dot3(x, A->je);
    J[0 * 6 + 5] = dot3(x, A->jf);
    J[1 * 6 + 5] = dot3(x, A->jg);
    J[2 * 6 + 5] = dot3(x, A->jh);
    {
        double va[3] = {0, dot3(x, A->ha2), dot3(x, A->ha3)};
        double vb[3] = {0, dot3(x, A->hb2), dot3(x, A->hb3)};
        double vc[3] = {0, dot3(x, A->hc2), dot3(x, A->hc3)};
        double vd[3] = {dot3(x, A->hd1), dot3(x, A->hd2), dot3(x, A->hd3)};
        double ve[3] = {dot3(x, A->he1), dot3(x, A->he2), dot3(x, A->he3)};
        double vf[3] = {dot3(x, A->hf1), dot3(x, A->hf2), dot3(x, A->hf3)};
        for (a = 0; a < 3; ++a) {
            PH[0][0][a] = va[a];
            PH[1][0][a] = PH[0][1][a] = vb[a];
            PH[2][0][a] = PH[0][2][a] = vc[a];
            PH[1][1][a] = vd[a];
            PH[2][1][a] = PH[1][2][a] = ve[a];
            PH[2][2][a] = vf[a];
        }
    }
}

static void gauss_constants(const wmo_ndt_params *prm, double *d1, double *d2) {
    double c1 = 10.0 * (1.0 - prm->outlier_ratio);
    double c2 = prm->outlier_ratio / pow(prm->res, 3);
    double d3 = -log(c2);
    *d1 = -log(c1 + c2) - d3;
    *d2 = -2.0 * log((-log(c1 * exp(-0.5) + c2) - d3) / *d1);
}

/* computeDerivatives (score + gradient, optionally the Hessian) or computeHessian
 * (hessian only), over the float-transformed source cloud */
static double derivatives(const wmo_ndt_grid *g, const float *src, int n, const wmo_ndt_params *prm,
                          const double p[6], double grad[6], double hess[36], int want_grad,
                          int want_hess) {
    float Tf[16];
    angle_derivs A;
    double d1, d2, score = 0;
    const float invf = 1.0f / (float) g->res;
    int idx, i, j, a, b;
    pose_to_matrix_f(p, Tf);
    compute_angle_derivatives(p, prm->pcl_d1_sign, &A);
    gauss_constants(prm, &d1, &d2);
    if (want_grad) memset(grad, 0, sizeof(double) * 6);
    if (want_hess) memset(hess, 0, sizeof(double) * 36);
    for (idx = 0; idx < n; ++idx) {
        const float *sp = src + 3 * idx;
        float xt[3];
        double x[3] = {sp[0], sp[1], sp[2]}, J[18], PH[3][3][3];
        int ci, cj, ck, di, dj, dk, derivs_ready = 0;
        xt[0] = ((Tf[0] * sp[0] + Tf[1] * sp[1]) + Tf[2] * sp[2]) + Tf[3];
        xt[1] = ((Tf[4] * sp[0] + Tf[5] * sp[1]) + Tf[6] * sp[2]) + Tf[7];
        xt[2] = ((Tf[8] * sp[0] + Tf[9] * sp[1]) + Tf[10] * sp[2]) + Tf[11];
        ci = (int) floorf(xt[0] * invf);
        cj = (int) floorf(xt[1] * invf);
        ck = (int) floorf(xt[2] * invf);
        /* radiusSearch(x', resolution) over the voxel MEANS: a mean within `res` of x'
         * lies in one of the 27 adjacent voxels.  Visit order: ascending (k, j, i). */
        for (dk = -1; dk <= 1; ++dk)
            for (dj = -1; dj <= 1; ++dj)
                for (di = -1; di <= 1; ++di) {
                    int vi = grid_find(g, ci + di, cj + dj, ck + dk);
                    const ndt_voxel *v;
                    double xx[3], cx[3], q, e, w;
                    if (vi < 0) continue;
                    v = &g->vox[vi];
                    if (!v->valid) continue;
                    for (a = 0; a < 3; ++a) xx[a] = (double) xt[a] - v->mean[a];
                    {
                        /* kd-tree distances are float: (float) mean vs float query */
                        float fx = xt[0] - (float) v->mean[0], fy = xt[1] - (float) v->mean[1],
                              fz = xt[2] - (float) v->mean[2];
                        float dd = (fx * fx + fy * fy) + fz * fz;
                        if (!((double) dd < g->res * g->res)) continue;
                    }
                    if (!derivs_ready) {
                        compute_point_derivatives(&A, x, J, PH);
                        derivs_ready = 1;
                    }
                    for (a = 0; a < 3; ++a)
                        cx[a] = v->icov[a * 3] * xx[0] + v->icov[a * 3 + 1] * xx[1] +
                                v->icov[a * 3 + 2] * xx[2];
                    q = dot3(xx, cx);
                    e = exp(-d2 * q / 2.0);
                    w = d2 * e;
                    if (w > 1 || w < 0 || w != w) continue; /* returns 0: no score either */
                    score += -d1 * e;
                    w *= d1;
                    {
                        double cJ[6][3], xcJ[6];
                        for (i = 0; i < 6; ++i) {
                            for (a = 0; a < 3; ++a)
                                cJ[i][a] = v->icov[a * 3] * J[0 * 6 + i] + v->icov[a * 3 + 1] * J[1 * 6 + i] +
                                           v->icov[a * 3 + 2] * J[2 * 6 + i];
                            xcJ[i] = dot3(xx, cJ[i]);
                            if (want_grad) grad[i] += xcJ[i] * w;
                        }
                        if (want_hess)
                            for (i = 0; i < 6; ++i)
                                for (j = 0; j < 6; ++j) {
                                    double t2 = 0, t3 = 0;
                                    if (i >= 3 && j >= 3) {
                                        double ch[3];
                                        const double *h = PH[i - 3][j - 3];
                                        for (a = 0; a < 3; ++a)
                                            ch[a] = v->icov[a * 3] * h[0] + v->icov[a * 3 + 1] * h[1] +
                                                    v->icov[a * 3 + 2] * h[2];
                                        t2 = dot3(xx, ch);
                                    }
                                    for (b = 0; b < 3; ++b) t3 += J[b * 6 + j] * cJ[i][b];
                                    hess[i * 6 + j] += w * (-d2 * xcJ[i] * xcJ[j] + t2 + t3);
                                }
                    }
                }
    }
    return score;
}

double wmo_ndt_derivatives(const wmo_ndt_grid *g, const float *src, int n, const wmo_ndt_params *prm,
                           const double p[6], double grad[6], double hess[36]) {
    return derivatives(g, src, n, prm, p, grad, hess, 1, 1);
}

/* ------------------------------------------------------ More-Thuente */
static double psi_mt(double a, double f_a, double f_0, double g_0, double mu) {
    return f_a - f_0 - mu * g_0 * a;
}
static double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

static int update_interval_mt(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u,
                              double *g_u, double a_t, double f_t, double g_t) {
    if (f_t > *f_l) {
        *a_u = a_t;
        *f_u = f_t;
        *g_u = g_t;
        return 0;
    } else if (g_t * (*a_l - a_t) > 0) {
        *a_l = a_t;
        *f_l = f_t;
        *g_l = g_t;
        return 0;
    } else if (g_t * (*a_l - a_t) < 0) {
        *a_u = *a_l;
        *f_u = *f_l;
        *g_u = *g_l;
        *a_l = a_t;
        *f_l = f_t;
        *g_l = g_t;
        return 0;
    }
    return 1;
}

static double trial_value_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u,
                             double a_t, double f_t, double g_t) {
    if (f_t > f_l) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
        return 0.5 * (a_q + a_c);
    } else if (g_t * g_l < 0) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
        return a_s;
    } else if (fabs(g_t) <= fabs(g_l)) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        if (a_t > a_l) return fmin(a_t + 0.66 * (a_u - a_t), a_n);
        return fmax(a_t + 0.66 * (a_u - a_t), a_n);
    } else {
        double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
        double w = sqrt(z * z - g_t * g_u);
        return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
    }
}

static double step_length_mt(const wmo_ndt_grid *g, const float *src, int n,
                             const wmo_ndt_params *prm, const double x[6], double step_dir[6],
                             double step_init, double step_max, double step_min, double *score,
                             double grad[6], double hess[36], double x_t_out[6]) {
    double phi_0 = -(*score), d_phi_0 = 0, x_t[6];
    const double mu = 1.e-4, nu = 0.9;
    const int max_step_iterations = 10;
    int step_iterations = 0, a, interval_converged, open_interval = 1;
    double a_l = 0, a_u = 0, f_l, g_l, f_u, g_u, a_t, phi_t, d_phi_t, psi_t, d_psi_t;
    for (a = 0; a < 6; ++a) d_phi_0 -= grad[a] * step_dir[a];
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) {
            memcpy(x_t_out, x, sizeof(double) * 6);
            return 0;
        }
        d_phi_0 *= -1;
        for (a = 0; a < 6; ++a) step_dir[a] *= -1;
    }
    f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu);
    g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
    f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu);
    g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
    /* PCL 1.8.x: `bool interval_converged = (step_max - step_min) > 0` (so the loop
     * below never runs); later releases: `< 0`.  skip_line_search selects 1.8.x. */
    interval_converged = prm->skip_line_search ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
    a_t = step_init;
    a_t = fmin(a_t, step_max);
    a_t = fmax(a_t, step_min);
    for (a = 0; a < 6; ++a) x_t[a] = x[a] + step_dir[a] * a_t;
    *score = derivatives(g, src, n, prm, x_t, grad, hess, 1, 1);
    phi_t = -(*score);
    d_phi_t = 0;
    for (a = 0; a < 6; ++a) d_phi_t -= grad[a] * step_dir[a];
    psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
    d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    while (!interval_converged && step_iterations < max_step_iterations &&
           !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        if (open_interval)
            a_t = trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else
            a_t = trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = fmin(a_t, step_max);
        a_t = fmax(a_t, step_min);
        for (a = 0; a < 6; ++a) x_t[a] = x[a] + step_dir[a] * a_t;
        *score = derivatives(g, src, n, prm, x_t, grad, hess, 1, 0);
        phi_t = -(*score);
        d_phi_t = 0;
        for (a = 0; a < 6; ++a) d_phi_t -= grad[a] * step_dir[a];
        psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = 0;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
            g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
            g_u = g_u + mu * d_phi_0;
        }
        if (open_interval)
            interval_converged = update_interval_mt(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
        else
            interval_converged = update_interval_mt(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
        ++step_iterations;
    }
    if (step_iterations) derivatives(g, src, n, prm, x_t, NULL, hess, 0, 1); /* computeHessian */
    memcpy(x_t_out, x_t, sizeof(double) * 6);
    return a_t;
}

void wmo_ndt_default_params(wmo_ndt_params *p) {
    p->res = 5;        /* ndt.hpp:40 */
    p->step_size = 3;  /* ndt.hpp:37 */
    p->t_eps = 1e-8;   /* ndt.hpp:39 */
    p->max_iter = 100; /* ndt.hpp:38 */
    p->outlier_ratio = 0.55;
    p->skip_line_search = 0;
    p->pcl_d1_sign = 1;
    p->force_iterations = 0;
}

int wmo_ndt_align(const float *src, int n, const float *tgt, int m, const wmo_ndt_params *prm,
                  double T_out[16], wmo_ndt_result *res) {
    wmo_ndt_grid *g = wmo_ndt_grid_build(tgt, m, prm->res);
    double p[6] = {0, 0, 0, 0, 0, 0}, grad[6], hess[36], score, delta[6];
    int iter = 0, converged = 0, a;
    float Tf[16];
    const int max_it = prm->force_iterations > 0 ? prm->force_iterations : prm->max_iter;
    score = derivatives(g, src, n, prm, p, grad, hess, 1, 1);
    while (!converged) {
        double neg[6], norm = 0, alpha, x_t[6];
        for (a = 0; a < 6; ++a) neg[a] = -grad[a];
        wmo_svd_solve(6, hess, neg, delta); /* JacobiSVD(hessian).solve(-score_gradient) */
        for (a = 0; a < 6; ++a) norm += delta[a] * delta[a];
        norm = sqrt(norm);
        if (norm == 0 || norm != norm) {
            converged = (norm == norm);
            break;
        }
        for (a = 0; a < 6; ++a) delta[a] /= norm;
        alpha = step_length_mt(g, src, n, prm, p, delta, norm, prm->step_size, prm->t_eps / 2, &score,
                               grad, hess, x_t);
        for (a = 0; a < 6; ++a) p[a] = p[a] + delta[a] * alpha;
        if (prm->force_iterations > 0) {
            if (iter + 1 >= max_it) converged = 1;
        } else if (iter > max_it || (iter && fabs(alpha) < prm->t_eps)) {
            converged = 1;
        }
        ++iter;
    }
    pose_to_matrix_f(p, Tf); /* final_transformation_ is the float matrix of the last x_t */
    for (a = 0; a < 16; ++a) T_out[a] = (double) Tf[a];
    if (res) {
        res->converged = converged;
        res->iterations = iter;
        res->n_voxels = g->nvalid;
        res->score = n > 0 ? score / n : 0;
    }
    wmo_ndt_grid_free(g);
    return converged ? 0 : 1;
}
