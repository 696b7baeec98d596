/* ORACLE internal helpers (test infrastructure, not product code). */
#ifndef WMO_INTERNAL_H
#define WMO_INTERNAL_H

#define WMO_MAXN 6

void wmo_mat_mul(int n, const double *A, const double *B, double *C);
double wmo_det3(const double *m);
void wmo_svd_solve(int n, const double *A, const double *b, double *x);
void wmo_umeyama_from_stats(double n, const double sp[3], const double sq[3],
                            const double sqp[9], double T[16]);

/* canonical float squared distance: (dx*dx + dy*dy) + dz*dz, no contraction */
static inline float wmo_d2(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

static inline void wmo_mat4_mul(const double *A, const double *B, double *C) {
    double T[16];
    int i, j, k;
    for (i = 0; i < 4; ++i)
        for (j = 0; j < 4; ++j) {
            double s = 0;
            for (k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            T[i * 4 + j] = s;
        }
    for (i = 0; i < 16; ++i) C[i] = T[i];
}

static inline void wmo_mat4_identity(double *T) {
    int i;
    for (i = 0; i < 16; ++i) T[i] = (i % 5 == 0);
}

#endif
