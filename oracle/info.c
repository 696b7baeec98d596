/*
 * info.c -- ORACLE (test infrastructure): libwave's own post-match information
 * matrix estimators, restated from
 *   wave_matching/src/icp_pcl_functions.cpp:182-289  (estimateLUM)
 *   wave_matching/src/icp_pcl_functions.cpp:51-179   (estimateLUMold)
 *   wave_matching/src/icp.cpp:167-397                (estimateCensi)
 * including their documented quirks (float ss accumulation, MM diagonal cast
 * through float, LUMold falling through its failure branch).
 */
#include "wm_oracle.h"
#include "wmo_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

const float *wmo_match_ref(const wmo_match_state *s);
const float *wmo_match_target(const wmo_match_state *s);
const float *wmo_match_final(const wmo_match_state *s);
const int *wmo_match_corr(const wmo_match_state *s);

/* Lu & Milios normal equations over (aligned-source p, matched-target q) pairs.
 * icp_pcl_functions.cpp:199-278.  aver/diff are float vectors, MM/MZ doubles,
 * ss a float accumulator of double terms cast to float. */
int wmo_lum_from_pairs(const float *p, const float *q, int n_corr, double info[36],
                       double mm_out[36], double mz_out[6], float *ss_out) {
    double MM[36] = {0}, MZ[6] = {0}, MMinv[36], D[6];
    float ss = 0.0f;
    int ci, a, b;
#define M_(r, c) MM[(r) * 6 + (c)]
    for (ci = 0; ci < n_corr; ++ci) {
        float av[3], df[3];
        for (a = 0; a < 3; ++a) {
            av[a] = 0.5f * (p[3 * ci + a] + q[3 * ci + a]);
            df[a] = p[3 * ci + a] - q[3 * ci + a];
        }
        M_(0, 4) -= av[1];
        M_(0, 5) += av[2];
        M_(1, 3) -= av[2];
        M_(1, 4) += av[0];
        M_(2, 3) += av[1];
        M_(2, 5) -= av[0];
        M_(3, 4) -= av[0] * av[2];
        M_(3, 5) -= av[0] * av[1];
        M_(4, 5) -= av[1] * av[2];
        M_(3, 3) += av[1] * av[1] + av[2] * av[2];
        M_(4, 4) += av[0] * av[0] + av[1] * av[1];
        M_(5, 5) += av[0] * av[0] + av[2] * av[2];
        MZ[0] += df[0];
        MZ[1] += df[1];
        MZ[2] += df[2];
        MZ[3] += av[1] * df[2] - av[2] * df[1];
        MZ[4] += av[0] * df[1] - av[1] * df[0];
        MZ[5] += av[2] * df[0] - av[0] * df[2];
    }
    M_(0, 0) = M_(1, 1) = M_(2, 2) = (double) (float) n_corr;
    M_(4, 0) = M_(0, 4);
    M_(5, 0) = M_(0, 5);
    M_(3, 1) = M_(1, 3);
    M_(4, 1) = M_(1, 4);
    M_(3, 2) = M_(2, 3);
    M_(5, 2) = M_(2, 5);
    M_(4, 3) = M_(3, 4);
    M_(5, 3) = M_(3, 5);
    M_(5, 4) = M_(4, 5);
#undef M_
    wmo_inverse(6, MM, MMinv);
    for (a = 0; a < 6; ++a) {
        double s = 0;
        for (b = 0; b < 6; ++b) s += MMinv[a * 6 + b] * MZ[b];
        D[a] = s;
    }
    for (ci = 0; ci < n_corr; ++ci) {
        float av[3], df[3];
        double e0, e1, e2;
        for (a = 0; a < 3; ++a) {
            av[a] = 0.5f * (p[3 * ci + a] + q[3 * ci + a]);
            df[a] = p[3 * ci + a] - q[3 * ci + a];
        }
        e0 = df[0] - (D[0] + av[2] * D[5] - av[1] * D[4]);
        e1 = df[1] - (D[1] + av[0] * D[4] - av[2] * D[3]);
        e2 = df[2] - (D[2] + av[1] * D[3] - av[0] * D[5]);
        ss += (float) (e0 * e0 + e1 * e1 + e2 * e2);
    }
    if (mm_out) memcpy(mm_out, MM, sizeof(MM));
    if (mz_out) memcpy(mz_out, MZ, sizeof(MZ));
    if (ss_out) *ss_out = ss;
    if (ss < 0.0000000000001 || !isfinite(ss)) {
        for (a = 0; a < 36; ++a) info[a] = (a % 7 == 0);
        return 1;
    }
    {
        float inv = 1.0f / ss;
        for (a = 0; a < 36; ++a) info[a] = MM[a] * inv;
    }
    return 0;
}

static int gather_pairs(const wmo_match_state *s, const float *src_cloud, float **p, float **q) {
    int n_ref, n_target, n_corr, i, k = 0;
    const int *corr = wmo_match_corr(s);
    const float *tgt = wmo_match_target(s);
    wmo_match_counts(s, &n_ref, &n_target, &n_corr);
    *p = (float *) malloc(sizeof(float) * 3 * (n_corr > 0 ? n_corr : 1));
    *q = (float *) malloc(sizeof(float) * 3 * (n_corr > 0 ? n_corr : 1));
    for (i = 0; i < n_ref; ++i) {
        if (corr[i] < 0) continue;
        memcpy(*p + 3 * k, src_cloud + 3 * i, 3 * sizeof(float));
        memcpy(*q + 3 * k, tgt + 3 * corr[i], 3 * sizeof(float));
        ++k;
    }
    return k;
}

int wmo_info_lum(const wmo_match_state *s, double info[36]) {
    float *p, *q;
    int n, rc;
    if (!wmo_match_counts(s, NULL, NULL, NULL)) return 2; /* !hasConverged(): untouched */
    n = gather_pairs(s, wmo_match_final(s), &p, &q);
    rc = wmo_lum_from_pairs(p, q, n, info, NULL, NULL, NULL);
    free(p);
    free(q);
    return rc;
}

int wmo_info_lumold(const wmo_match_state *s, double max_corr, double info[36]) {
    int n_ref, n_target, i, k = 0, a;
    const float *fin = wmo_match_final(s), *tgt = wmo_match_target(s);
    float *p, *q, ss;
    double MM[36], dummy[36];
    wmo_kdtree *tree;
    wmo_match_counts(s, &n_ref, &n_target, NULL);
    tree = wmo_kdtree_build(tgt, n_target);
    p = (float *) malloc(sizeof(float) * 3 * (n_ref > 0 ? n_ref : 1));
    q = (float *) malloc(sizeof(float) * 3 * (n_ref > 0 ? n_ref : 1));
    for (i = 0; i < n_ref; ++i) {
        float d2;
        int j = wmo_kdtree_nn(tree, fin + 3 * i, &d2);
        if (j >= 0 && (double) d2 < max_corr * max_corr) { /* icp_pcl_functions.cpp:82 */
            memcpy(p + 3 * k, fin + 3 * i, 3 * sizeof(float));
            memcpy(q + 3 * k, tgt + 3 * j, 3 * sizeof(float));
            ++k;
        }
    }
    wmo_lum_from_pairs(p, q, k, dummy, MM, NULL, &ss);
    /* icp_pcl_functions.cpp:170-178: the failure branch does not return */
    {
        float inv = 1.0f / ss;
        for (a = 0; a < 36; ++a) info[a] = MM[a] * inv;
    }
    free(p);
    free(q);
    wmo_kdtree_free(tree);
    return (ss < 0.0000000000001 || !isfinite(ss)) ? 1 : 0;
}

/* estimateCensi, icp.cpp:176-395.  ref_pts = Z4..Z6 (source cloud points as
 * handed to icp), tgt_pts = Z1..Z3 (matched target points). */
int wmo_censi_from_pairs(const float *ref_pts, const float *tgt_pts, int n_corr,
                         const double T[16], double lin_covar, double ang_covar,
                         double info[36], double d2j_dx2_out[36], double middle_out[36]) {
    double R[9], eul[3];
    double X1 = T[3], X2 = T[7], X3 = T[11];
    double cr, sr, cp, sp, cy, sy;
    double sph[6] = {lin_covar, ang_covar, ang_covar, lin_covar, ang_covar, ang_covar};
    double H[36] = {0}, middle[36] = {0};
    int it, a, b, c;
    for (a = 0; a < 3; ++a)
        for (b = 0; b < 3; ++b) R[a * 3 + b] = T[a * 4 + b];
    wmo_euler_angles_012(R, eul);
    cr = cos(eul[0]);
    sr = sin(eul[0]);
    cp = cos(eul[1]);
    sp = sin(eul[1]);
    cy = cos(eul[2]);
    sy = sin(eul[2]);
#define H_(r, c) H[(r) * 6 + (c)]
#define G_(r, c) G[(r) * 6 + (c)]
    for (it = 0; it < n_corr; ++it) {
        const float Z1 = tgt_pts[3 * it], Z2 = tgt_pts[3 * it + 1], Z3 = tgt_pts[3 * it + 2];
        const float Z4 = ref_pts[3 * it], Z5 = ref_pts[3 * it + 1], Z6 = ref_pts[3 * it + 2];
        double j[36] = {0}, covZ[36], G[36] = {0}, tmp[36];
        double rg, br, az;
        /* float expression under std::sqrt / atan2 -> float overloads, as in the
         * reference where Z* are `const float&` (icp.cpp:217-233) */
        rg = sqrtf(Z1 * Z1 + Z2 * Z2 + Z3 * Z3);
        br = atan2f(Z2, Z1);
        az = atanf(Z3 / sqrtf(Z1 * Z1 + Z2 * Z2));
        j[0 * 6 + 0] = cos(br) * sin(az);
        j[1 * 6 + 0] = sin(br) * sin(az);
        j[2 * 6 + 0] = cos(az);
        j[0 * 6 + 1] = -rg * sin(br) * sin(az);
        j[1 * 6 + 1] = rg * cos(br) * sin(az);
        j[0 * 6 + 2] = rg * cos(br) * cos(az);
        j[1 * 6 + 2] = rg * cos(az) * sin(br);
        j[2 * 6 + 2] = -rg * sin(az);
        rg = sqrtf(Z4 * Z4 + Z5 * Z5 + Z6 * Z6);
        br = atan2f(Z5, Z4);
        az = atanf(Z6 / sqrtf(Z4 * Z4 + Z5 * Z5));
        j[3 * 6 + 3] = cos(br) * sin(az);
        j[4 * 6 + 3] = sin(br) * sin(az);
        j[5 * 6 + 3] = cos(az);
        j[3 * 6 + 4] = -rg * sin(br) * sin(az);
        j[4 * 6 + 4] = rg * cos(br) * sin(az);
        j[3 * 6 + 5] = rg * cos(br) * cos(az);
        j[4 * 6 + 5] = rg * cos(az) * sin(br);
        j[5 * 6 + 5] = -rg * sin(az);
        for (a = 0; a < 6; ++a)
            for (b = 0; b < 6; ++b) {
                double s = 0;
                for (c = 0; c < 6; ++c) s += j[a * 6 + c] * sph[c] * j[b * 6 + c];
                covZ[a * 6 + b] = s;
            }

        H_(0, 0) += 2;
        H_(1, 1) += 2;
        H_(2, 2) += 2;
        H_(0, 3) += 2 * Z2 * (sr * sy + cr * cy * sp) + 2 * Z3 * (cr * sy - cy * sr * sp);
        H_(1, 3) += -2 * Z2 * (cy * sr - cr * sp * sy) - 2 * Z3 * (cr * cy + sr * sp * sy);
        H_(2, 3) += 2 * cp * (Z2 * cr - Z3 * sr);
        H_(3, 3) +=
            (2 * Z2 * (cr * sy - cy * sr * sp) - 2 * Z3 * (sr * sy + cr * cy * sp)) *
                (X1 - Z4 - Z2 * (cr * sy - cy * sr * sp) + Z3 * (sr * sy + cr * cy * sp) +
                 Z1 * cp * cy) -
            (2 * Z2 * (cr * cy + sr * sp * sy) - 2 * Z3 * (cy * sr - cr * sp * sy)) *
                (X2 - Z5 + Z2 * (cr * cy + sr * sp * sy) - Z3 * (cy * sr - cr * sp * sy) +
                 Z1 * cp * sy) -
            (2 * Z3 * cr * cp + 2 * Z2 * cp * sr) *
                (X3 - Z6 - Z1 * sp + Z3 * cr * cp + Z2 * cp * sr) +
            (Z2 * (sr * sy + cr * cy * sp) + Z3 * (cr * sy - cy * sr * sp)) *
                (2 * Z2 * (sr * sy + cr * cy * sp) + 2 * Z3 * (cr * sy - cy * sr * sp)) +
            (Z2 * (cy * sr - cr * sp * sy) + Z3 * (cr * cy + sr * sp * sy)) *
                (2 * Z2 * (cy * sr - cr * sp * sy) + 2 * Z3 * (cr * cy + sr * sp * sy)) +
            (Z2 * cr * cp - Z3 * cp * sr) * (2 * Z2 * cr * cp - 2 * Z3 * cp * sr);
        H_(0, 4) += 2 * cy * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr);
        H_(1, 4) += 2 * sy * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr);
        H_(2, 4) += -2 * Z1 * cp - 2 * Z3 * cr * sp - 2 * Z2 * sr * sp;
        H_(3, 4) += -2 * (Z2 * cr - Z3 * sr) *
                    (X3 * sp - Z6 * sp - X1 * cp * cy + Z4 * cp * cy - X2 * cp * sy + Z5 * cp * sy);
        H_(4, 4) +=
            (Z1 * cp + Z3 * cr * sp + Z2 * sr * sp) * (2 * Z1 * cp + 2 * Z3 * cr * sp + 2 * Z2 * sr * sp) -
            (2 * Z3 * cr * cp - 2 * Z1 * sp + 2 * Z2 * cp * sr) *
                (X3 - Z6 - Z1 * sp + Z3 * cr * cp + Z2 * cp * sr) +
            2 * cy * cy * pow((Z3 * cr * cp - Z1 * sp + Z2 * cp * sr), 2) +
            2 * sy * sy * pow((Z3 * cr * cp - Z1 * sp + Z2 * cp * sr), 2) -
            2 * cy * (Z1 * cp + Z3 * cr * sp + Z2 * sr * sp) *
                (X1 - Z4 + Z1 * cp * cy - Z2 * cr * sy + Z3 * sr * sy + Z2 * cy * sr * sp +
                 Z3 * cr * cy * sp) -
            2 * sy * (Z1 * cp + Z3 * cr * sp + Z2 * sr * sp) *
                (X2 - Z5 + Z2 * cr * cy + Z1 * cp * sy - Z3 * cy * sr + Z3 * cr * sp * sy +
                 Z2 * sr * sp * sy);
        H_(0, 5) += 2 * Z3 * (cy * sr - cr * sp * sy) - 2 * Z2 * (cr * cy + sr * sp * sy) -
                    2 * Z1 * cp * sy;
        H_(1, 5) += 2 * Z3 * (sr * sy + cr * cy * sp) - 2 * Z2 * (cr * sy - cy * sr * sp) +
                    2 * Z1 * cp * cy;
        H_(3, 5) += 2 * X1 * Z3 * cr * cy - 2 * Z3 * Z4 * cr * cy + 2 * X1 * Z2 * cy * sr +
                    2 * X2 * Z3 * cr * sy - 2 * Z2 * Z4 * cy * sr - 2 * Z3 * Z5 * cr * sy +
                    2 * X2 * Z2 * sr * sy - 2 * Z2 * Z5 * sr * sy + 2 * X2 * Z2 * cr * cy * sp -
                    2 * Z2 * Z5 * cr * cy * sp - 2 * X1 * Z2 * cr * sp * sy -
                    2 * X2 * Z3 * cy * sr * sp + 2 * Z2 * Z4 * cr * sp * sy +
                    2 * Z3 * Z5 * cy * sr * sp + 2 * X1 * Z3 * sr * sp * sy -
                    2 * Z3 * Z4 * sr * sp * sy;
        H_(4, 5) += 2 * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr) *
                    (X2 * cy - Z5 * cy - X1 * sy + Z4 * sy);
        H_(5, 5) += 2 * Z1 * Z4 * cp * cy - 2 * X2 * Z2 * cr * cy - 2 * X1 * Z1 * cp * cy +
                    2 * Z2 * Z5 * cr * cy + 2 * X1 * Z2 * cr * sy - 2 * X2 * Z1 * cp * sy +
                    2 * X2 * Z3 * cy * sr - 2 * Z2 * Z4 * cr * sy + 2 * Z1 * Z5 * cp * sy -
                    2 * Z3 * Z5 * cy * sr - 2 * X1 * Z3 * sr * sy + 2 * Z3 * Z4 * sr * sy -
                    2 * X1 * Z3 * cr * cy * sp + 2 * Z3 * Z4 * cr * cy * sp -
                    2 * X1 * Z2 * cy * sr * sp - 2 * X2 * Z3 * cr * sp * sy +
                    2 * Z2 * Z4 * cy * sr * sp + 2 * Z3 * Z5 * cr * sp * sy -
                    2 * X2 * Z2 * sr * sp * sy + 2 * Z2 * Z5 * sr * sp * sy;

        G_(3, 0) = -2;
        G_(4, 1) = -2;
        G_(5, 2) = -2;
        G_(0, 0) = 2 * cp * cy;
        G_(1, 0) = 2 * cy * sr * sp - 2 * cr * sy;
        G_(2, 0) = 2 * sr * sy + 2 * cr * cy * sp;
        G_(0, 1) = 2 * cp * sy;
        G_(1, 1) = 2 * cr * cy + 2 * sr * sp * sy;
        G_(2, 1) = 2 * cr * sp * sy - 2 * cy * sr;
        G_(0, 2) = -2 * sp;
        G_(1, 2) = 2 * cp * sr;
        G_(2, 2) = 2 * cr * cp;
        G_(1, 3) = 2 * X3 * cr * cp - 2 * Z6 * cr * cp - 2 * X2 * cy * sr + 2 * Z5 * cy * sr +
                   2 * X1 * sr * sy - 2 * Z4 * sr * sy + 2 * X2 * cr * sp * sy -
                   2 * Z5 * cr * sp * sy + 2 * X1 * cr * cy * sp - 2 * Z4 * cr * cy * sp;
        G_(2, 3) = 2 * Z5 * cr * cy - 2 * X2 * cr * cy + 2 * X1 * cr * sy - 2 * X3 * cp * sr -
                   2 * Z4 * cr * sy + 2 * Z6 * cp * sr - 2 * X1 * cy * sr * sp +
                   2 * Z4 * cy * sr * sp - 2 * X2 * sr * sp * sy + 2 * Z5 * sr * sp * sy;
        G_(3, 3) = -2 * Z2 * (sr * sy + cr * cy * sp) - 2 * Z3 * (cr * sy - cy * sr * sp);
        G_(4, 3) = 2 * Z2 * (cy * sr - cr * sp * sy) + 2 * Z3 * (cr * cy + sr * sp * sy);
        G_(5, 3) = -2 * cp * (Z2 * cr - Z3 * sr);
        G_(0, 4) = 2 * Z6 * cp - 2 * X3 * cp - 2 * X1 * cy * sp + 2 * Z4 * cy * sp -
                   2 * X2 * sp * sy + 2 * Z5 * sp * sy;
        G_(1, 4) = -2 * sr *
                   (X3 * sp - Z6 * sp - X1 * cp * cy + Z4 * cp * cy - X2 * cp * sy + Z5 * cp * sy);
        G_(2, 4) = -2 * cr *
                   (X3 * sp - Z6 * sp - X1 * cp * cy + Z4 * cp * cy - X2 * cp * sy + Z5 * cp * sy);
        G_(3, 4) = -2 * cy * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr);
        G_(4, 4) = -2 * sy * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr);
        G_(5, 4) = 2 * Z1 * cp + 2 * Z3 * cr * sp + 2 * Z2 * sr * sp;
        G_(0, 5) = 2 * cp * (X2 * cy - Z5 * cy - X1 * sy + Z4 * sy);
        G_(1, 5) = 2 * Z4 * cr * cy - 2 * X1 * cr * cy - 2 * X2 * cr * sy + 2 * Z5 * cr * sy +
                   2 * X2 * cy * sr * sp - 2 * Z5 * cy * sr * sp - 2 * X1 * sr * sp * sy +
                   2 * Z4 * sr * sp * sy;
        G_(2, 5) = 2 * X1 * cy * sr - 2 * Z4 * cy * sr + 2 * X2 * sr * sy - 2 * Z5 * sr * sy -
                   2 * X1 * cr * sp * sy + 2 * Z4 * cr * sp * sy + 2 * X2 * cr * cy * sp -
                   2 * Z5 * cr * cy * sp;
        G_(3, 5) = 2 * Z2 * (cr * cy + sr * sp * sy) - 2 * Z3 * (cy * sr - cr * sp * sy) +
                   2 * Z1 * cp * sy;
        G_(4, 5) = 2 * Z2 * (cr * sy - cy * sr * sp) - 2 * Z3 * (sr * sy + cr * cy * sp) -
                   2 * Z1 * cp * cy;
        /* middle += G covZ G^T */
        wmo_mat_mul(6, G, covZ, tmp);
        for (a = 0; a < 6; ++a)
            for (b = 0; b < 6; ++b) {
                double s = 0;
                for (c = 0; c < 6; ++c) s += tmp[a * 6 + c] * G[b * 6 + c];
                middle[a * 6 + b] += s;
            }
    }
#undef H_
#undef G_
    /* selfadjointView<Upper> */
    for (a = 0; a < 6; ++a)
        for (b = 0; b < a; ++b) H[a * 6 + b] = H[b * 6 + a];
    if (d2j_dx2_out) memcpy(d2j_dx2_out, H, sizeof(H));
    if (middle_out) memcpy(middle_out, middle, sizeof(middle));
    {
        double Hinv[36], t1[36], t2[36];
        wmo_inverse(6, H, Hinv);
        wmo_mat_mul(6, Hinv, middle, t1);
        wmo_mat_mul(6, t1, Hinv, t2);
        wmo_inverse(6, t2, info);
    }
    return 0;
}

int wmo_info_censi(const wmo_match_state *s, const double T[16], double lin_covar,
                   double ang_covar, double info[36]) {
    float *p, *q;
    int n, rc;
    if (!wmo_match_counts(s, NULL, NULL, NULL)) return 2;
    n = gather_pairs(s, wmo_match_ref(s), &p, &q);
    rc = wmo_censi_from_pairs(p, q, n, T, lin_covar, ang_covar, info, NULL, NULL);
    free(p);
    free(q);
    return rc;
}
