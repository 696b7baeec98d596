/*
 * icp.c -- ORACLE (test infrastructure): CPU restatement of
 *   pcl::IterativeClosestPoint<PointXYZ,PointXYZ>::align   (PCL 1.8, Scalar=float)
 *   pcl::VoxelGrid<PointXYZ>::filter
 *   pcl::transformPointCloud
 * as configured and driven by libwave's ICPMatcher:
 *   wave_matching/src/icp.cpp:32-51   (setters: max_corr, max_iter, t_eps, fit_eps)
 *   wave_matching/src/icp.cpp:75-133  (match(): multiscale / filtered / full-res)
 * PCL is an un-vendored dependency ("PCL 1.8", /root/reference/CMakeLists.txt:48);
 * the algorithm below follows its published sources:
 *   registration/impl/icp.hpp (computeTransformation),
 *   registration/impl/correspondence_estimation.hpp (determineCorrespondences),
 *   registration/impl/transformation_estimation_svd.hpp (pcl::umeyama),
 *   registration/impl/default_convergence_criteria.hpp (hasConverged),
 *   filters/impl/voxel_grid.hpp (applyFilter), common/impl/transforms.hpp.
 * PARITY: unpinned at 1e-4 (no PCL to run); see wm_oracle.h header.
 */
#include "wm_oracle.h"
#include "wmo_internal.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ transforms */
void wmo_transform_cloud_d(const float *in, int n, const double T[16], float *out) {
    int i;
    for (i = 0; i < n; ++i) {
        double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        out[3 * i + 0] = (float) (T[0] * x + T[1] * y + T[2] * z + T[3]);
        out[3 * i + 1] = (float) (T[4] * x + T[5] * y + T[6] * z + T[7]);
        out[3 * i + 2] = (float) (T[8] * x + T[9] * y + T[10] * z + T[11]);
    }
}

void wmo_transform_cloud_f(const float *in, int n, const float T[16], float *out) {
    int i;
    for (i = 0; i < n; ++i) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        out[3 * i + 0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
        out[3 * i + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
        out[3 * i + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    }
}

/* -------------------------------------------------------------- VoxelGrid */
typedef struct {
    unsigned int idx;
    int pt;
} vg_pair;

static int vg_cmp(const void *a, const void *b) {
    const vg_pair *x = (const vg_pair *) a, *y = (const vg_pair *) b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt); /* stable: ascending point index */
}

int wmo_voxel_grid(const float *in, int n, float leaf, float *out) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float inv = 1.0f / leaf;
    int i, d, n_fin = 0, n_out = 0;
    int min_b[3], max_b[3], div_b[3], mul[3];
    vg_pair *pairs;
    for (i = 0; i < n; ++i) {
        const float *p = in + 3 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (d = 0; d < 3; ++d) {
            if (p[d] < mn[d]) mn[d] = p[d];
            if (p[d] > mx[d]) mx[d] = p[d];
        }
        ++n_fin;
    }
    if (n_fin == 0) return 0;
    {
        int64_t dx = (int64_t) ((mx[0] - mn[0]) * inv) + 1;
        int64_t dy = (int64_t) ((mx[1] - mn[1]) * inv) + 1;
        int64_t dz = (int64_t) ((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > (int64_t) INT_MAX) { /* PCL warns and returns the input */
            memcpy(out, in, sizeof(float) * 3 * n);
            return n;
        }
    }
    for (d = 0; d < 3; ++d) {
        min_b[d] = (int) floorf(mn[d] * inv);
        max_b[d] = (int) floorf(mx[d] * inv);
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    mul[0] = 1;
    mul[1] = div_b[0];
    mul[2] = div_b[0] * div_b[1];
    pairs = (vg_pair *) malloc(sizeof(vg_pair) * n_fin);
    n_fin = 0;
    for (i = 0; i < n; ++i) {
        const float *p = in + 3 * i;
        int ijk[3];
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (d = 0; d < 3; ++d) ijk[d] = (int) (floorf(p[d] * inv) - (float) min_b[d]);
        pairs[n_fin].idx = (unsigned int) (ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2]);
        pairs[n_fin].pt = i;
        ++n_fin;
    }
    qsort(pairs, n_fin, sizeof(vg_pair), vg_cmp);
    i = 0;
    while (i < n_fin) {
        int j = i;
        float c[3] = {0, 0, 0};
        while (j < n_fin && pairs[j].idx == pairs[i].idx) {
            const float *p = in + 3 * pairs[j].pt;
            c[0] += p[0];
            c[1] += p[1];
            c[2] += p[2];
            ++j;
        }
        {
            float cnt = (float) (j - i);
            out[3 * n_out + 0] = c[0] / cnt;
            out[3 * n_out + 1] = c[1] / cnt;
            out[3 * n_out + 2] = c[2] / cnt;
        }
        ++n_out;
        i = j;
    }
    free(pairs);
    return n_out;
}

/* -------------------------------------------------------------------- ICP */
void wmo_icp_default_params(wmo_icp_params *p) {
    p->max_corr = 3;       /* icp.hpp:35 */
    p->max_iter = 100;     /* icp.hpp:37 */
    p->t_eps = 1e-8;       /* icp.hpp:41 */
    p->fit_eps = 1e-2;     /* icp.hpp:43 */
    p->force_iterations = 0;
    p->mode = WMO_ICP_SVD;
    p->float_sums = 0;
    p->incremental_float = 1;
    p->prev_mse_in = -1;
}

static void rodrigues(const double w[3], double R[9]) {
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double a, b;
    double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    int i;
    if (th < 1e-12) {
        a = 1.0;
        b = 0.5;
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / (th * th);
    }
    wmo_mat_mul(3, K, K, K2);
    for (i = 0; i < 9; ++i) R[i] = (i % 4 == 0) + a * K[i] + b * K2[i];
}

/* Gauss-Newton step on sum |p + dt + dw x p - q|^2 (left perturbation):
 * J = [I | -[p]x], (J^T J) delta = -J^T r.  north_star's literal "6x6" solver. */
static void gn6_step(const float *p, const float *q, int n, double T[16]) {
    double H[36] = {0}, g[6] = {0}, Hinv[36], delta[6], R[9];
    int i, a, b;
    for (i = 0; i < n; ++i) {
        double px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
        double r[3] = {px - q[3 * i], py - q[3 * i + 1], pz - q[3 * i + 2]};
        double J[3][6] = {{1, 0, 0, 0, pz, -py}, {0, 1, 0, -pz, 0, px}, {0, 0, 1, py, -px, 0}};
        int c;
        for (a = 0; a < 6; ++a) {
            for (c = 0; c < 3; ++c) g[a] += J[c][a] * r[c];
            for (b = 0; b < 6; ++b)
                for (c = 0; c < 3; ++c) H[a * 6 + b] += J[c][a] * J[c][b];
        }
    }
    wmo_inverse(6, H, Hinv);
    for (a = 0; a < 6; ++a) {
        double s = 0;
        for (b = 0; b < 6; ++b) s += Hinv[a * 6 + b] * g[b];
        delta[a] = -s;
    }
    rodrigues(delta + 3, R);
    wmo_mat4_identity(T);
    for (a = 0; a < 3; ++a) {
        for (b = 0; b < 3; ++b) T[a * 4 + b] = R[a * 3 + b];
        T[a * 4 + 3] = delta[a];
    }
}

int wmo_icp_align(const float *src, int n, const float *tgt, int m, const wmo_icp_params *p,
                  double T_out[16], wmo_icp_result *res, int *corr_idx, float *corr_d2,
                  float *final_xyz, double *trace) {
    wmo_kdtree *tree = wmo_kdtree_build(tgt, m);
    float *cur = (float *) malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    float *pp = (float *) malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    float *qq = (float *) malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    int *cidx = (int *) malloc(sizeof(int) * (n > 0 ? n : 1));
    float *cd2 = (float *) malloc(sizeof(float) * (n > 0 ? n : 1));
    float finalf[16];
    double finald[16];
    double prev_mse = p->prev_mse_in < 0 ? DBL_MAX : p->prev_mse_in;
    const double max_d2 = p->max_corr * p->max_corr;
    const double rot_thr = 1.0 - p->t_eps, trans_thr = p->t_eps;
    const int max_it = p->force_iterations > 0 ? p->force_iterations : p->max_iter;
    int iter = 0, converged = 0, state = WMO_CONV_NOT_CONVERGED, n_corr = 0, i;
    double mse = 0;

    memcpy(cur, src, sizeof(float) * 3 * n);
    wmo_mat4_identity(finald);
    for (i = 0; i < 16; ++i) finalf[i] = (float) finald[i];

    do {
        double Tk[16];
        /* determineCorrespondences(max_corr) */
        n_corr = 0;
        mse = 0;
        for (i = 0; i < n; ++i) {
            float d2;
            int j = wmo_kdtree_nn(tree, cur + 3 * i, &d2);
            cidx[i] = -1;
            cd2[i] = d2;
            if (j < 0 || (double) d2 > max_d2) continue;
            cidx[i] = j;
            memcpy(pp + 3 * n_corr, cur + 3 * i, 3 * sizeof(float));
            memcpy(qq + 3 * n_corr, tgt + 3 * j, 3 * sizeof(float));
            mse += (double) d2;
            ++n_corr;
        }
        if (n_corr > 0) mse /= n_corr;
        if (trace) {
            trace[2 * iter] = n_corr;
            trace[2 * iter + 1] = mse;
        }
        if (n_corr < 3) { /* min_number_correspondences_ */
            state = WMO_CONV_NO_CORRESPONDENCES;
            converged = 0;
            break;
        }
        if (p->mode == WMO_ICP_GN6)
            gn6_step(pp, qq, n_corr, Tk);
        else
            wmo_umeyama(pp, qq, n_corr, p->float_sums, Tk);

        if (p->incremental_float) {
            float Tf[16], nf[16];
            int a, b, c;
            for (i = 0; i < 16; ++i) Tf[i] = (float) Tk[i];
            wmo_transform_cloud_f(cur, n, Tf, cur); /* transformCloud, in place */
            for (a = 0; a < 4; ++a)                 /* final = T_k * final (float) */
                for (b = 0; b < 4; ++b) {
                    float s = 0;
                    for (c = 0; c < 4; ++c) s += Tf[a * 4 + c] * finalf[c * 4 + b];
                    nf[a * 4 + b] = s;
                }
            memcpy(finalf, nf, sizeof(nf));
            for (i = 0; i < 16; ++i) Tk[i] = (double) Tf[i]; /* criteria see the float T */
        } else {
            float Tf[16];
            wmo_mat4_mul(Tk, finald, finald);
            for (i = 0; i < 16; ++i) Tf[i] = (float) finald[i];
            wmo_transform_cloud_f(src, n, Tf, cur);
        }
        ++iter;

        /* DefaultConvergenceCriteria::hasConverged() */
        if (p->force_iterations > 0) {
            if (iter >= max_it) {
                converged = 1;
                state = WMO_CONV_FORCED;
            }
            prev_mse = mse;
            continue;
        }
        if (iter >= max_it) {
            converged = 1;
            state = WMO_CONV_ITERATIONS;
            break;
        }
        {
            double cos_angle = 0.5 * (Tk[0] + Tk[5] + Tk[10] - 1.0);
            double tsq = Tk[3] * Tk[3] + Tk[7] * Tk[7] + Tk[11] * Tk[11];
            if (cos_angle >= rot_thr && tsq <= trans_thr) {
                converged = 1; /* max_iterations_similar_transforms_ = 0 */
                state = WMO_CONV_TRANSFORM;
                break;
            }
            if (fabs(mse - prev_mse) < 1e-12) {
                converged = 1;
                state = WMO_CONV_ABS_MSE;
                break;
            }
            if (fabs(mse - prev_mse) / prev_mse < p->fit_eps) {
                converged = 1;
                state = WMO_CONV_REL_MSE;
                break;
            }
            prev_mse = mse;
        }
    } while (!converged);

    if (p->incremental_float)
        for (i = 0; i < 16; ++i) finald[i] = (double) finalf[i];
    memcpy(T_out, finald, sizeof(finald));
    if (final_xyz) {
        float Tf[16];
        for (i = 0; i < 16; ++i) Tf[i] = (float) finald[i];
        wmo_transform_cloud_f(src, n, Tf, final_xyz); /* output = final * source */
    }
    if (corr_idx) memcpy(corr_idx, cidx, sizeof(int) * n);
    if (corr_d2) memcpy(corr_d2, cd2, sizeof(float) * n);
    if (res) {
        res->converged = converged;
        res->iterations = iter;
        res->state = state;
        res->n_corr = n_corr;
        res->mse = mse;
        res->prev_mse_out = prev_mse;
    }
    free(cur);
    free(pp);
    free(qq);
    free(cidx);
    free(cd2);
    wmo_kdtree_free(tree);
    return converged ? 0 : 1;
}

/* ------------------------------------------------ ICPMatcher::match() */
struct wmo_match_state {
    float *ref;    /* the cloud handed to icp.setInputSource at the last align */
    float *target; /* ... setInputTarget */
    float *final;  /* PCL `output` of the last align */
    int *corr_idx; /* last iteration's correspondences (per ref point, -1 none) */
    int n_ref, n_target, n_corr, converged;
};

void wmo_match_free(wmo_match_state *s) {
    if (!s) return;
    free(s->ref);
    free(s->target);
    free(s->final);
    free(s->corr_idx);
    free(s);
}

int wmo_match_counts(const wmo_match_state *s, int *n_ref, int *n_target, int *n_corr) {
    if (n_ref) *n_ref = s->n_ref;
    if (n_target) *n_target = s->n_target;
    if (n_corr) *n_corr = s->n_corr;
    return s->converged;
}

const float *wmo_match_ref(const wmo_match_state *s) { return s->ref; }
const float *wmo_match_target(const wmo_match_state *s) { return s->target; }
const float *wmo_match_final(const wmo_match_state *s) { return s->final; }
const int *wmo_match_corr(const wmo_match_state *s) { return s->corr_idx; }

static void state_take(wmo_match_state *s, float *r, int nr, float *t, int nt) {
    free(s->ref);
    free(s->target);
    free(s->final);
    free(s->corr_idx);
    s->ref = r;
    s->n_ref = nr;
    s->target = t;
    s->n_target = nt;
    s->final = (float *) malloc(sizeof(float) * 3 * (nr > 0 ? nr : 1));
    s->corr_idx = (int *) malloc(sizeof(int) * (nr > 0 ? nr : 1));
}

static float *dup_cloud(const float *c, int n) {
    float *o = (float *) malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    memcpy(o, c, sizeof(float) * 3 * n);
    return o;
}

int wmo_icp_match(const float *ref, int n, const float *target, int m, const wmo_icp_params *p,
                  float resl, int multiscale_steps, double T_out[16], wmo_icp_result *res_out,
                  wmo_match_state **state_out) {
    wmo_match_state *s = (wmo_match_state *) calloc(1, sizeof(wmo_match_state));
    wmo_icp_params prm = *p;
    wmo_icp_result r;
    int ok = 0, i;
    memset(&r, 0, sizeof(r));
    if (resl > 0 && multiscale_steps > 0) {
        double running[16];
        wmo_mat4_identity(running);
        ok = 1;
        for (i = multiscale_steps; i >= 0; --i) { /* icp.cpp:79-102 */
            float leaf = (float) (pow(2, i) * resl);
            float *dr = (float *) malloc(sizeof(float) * 3 * n);
            float *dt = (float *) malloc(sizeof(float) * 3 * m);
            int nr = wmo_voxel_grid(ref, n, leaf, dr);
            int nt = wmo_voxel_grid(target, m, leaf, dt);
            double Ti[16];
            wmo_transform_cloud_d(dr, nr, running, dr); /* icp.cpp:84-86 */
            prm.max_corr = pow(2, i) * p->max_corr;     /* icp.cpp:93-94 */
            state_take(s, dr, nr, dt, nt);
            wmo_icp_align(dr, nr, dt, nt, &prm, Ti, &r, s->corr_idx, NULL, s->final, NULL);
            prm.prev_mse_in = r.prev_mse_out; /* criteria object persists across align() */
            if (!r.converged) {
                ok = 0;
                break;
            }
            wmo_mat4_mul(Ti, running, running); /* icp.cpp:99-101 */
        }
        if (ok) memcpy(T_out, running, sizeof(running));
    } else if (resl > 0) { /* icp.cpp:105-122 */
        float *dr = (float *) malloc(sizeof(float) * 3 * n);
        float *dt = (float *) malloc(sizeof(float) * 3 * m);
        int nr = wmo_voxel_grid(ref, n, resl, dr);
        int nt = wmo_voxel_grid(target, m, resl, dt);
        double Ti[16];
        state_take(s, dr, nr, dt, nt);
        wmo_icp_align(dr, nr, dt, nt, &prm, Ti, &r, s->corr_idx, NULL, s->final, NULL);
        ok = r.converged;
        if (ok) memcpy(T_out, Ti, sizeof(Ti));
    } else { /* icp.cpp:123-131 */
        double Ti[16];
        state_take(s, dup_cloud(ref, n), n, dup_cloud(target, m), m);
        wmo_icp_align(ref, n, target, m, &prm, Ti, &r, s->corr_idx, NULL, s->final, NULL);
        ok = r.converged;
        if (ok) memcpy(T_out, Ti, sizeof(Ti));
    }
    s->converged = r.converged;
    s->n_corr = 0;
    for (i = 0; i < s->n_ref; ++i) s->n_corr += (s->corr_idx[i] >= 0);
    if (res_out) *res_out = r;
    if (state_out)
        *state_out = s;
    else
        wmo_match_free(s);
    return ok ? 0 : 1;
}
