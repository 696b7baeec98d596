// wm_ndt_dev.hpp -- what more than one NDT kernel is made of: a voxel's record and how it is finished from its sums
// (pcl::VoxelGridCovariance), the arguments of a derivative pass (pose as PCL's float matrix, computeAngleDerivatives'
// 8 + 15 vectors) and how they follow from a pose.  Used by wm_ndt.hip (one registration on the whole device) and
// wm_ndt_small.hip (one registration per workgroup).  [PCL registration/impl/ndt.hpp, filters/impl/voxel_grid_covariance.hpp]
#pragma once
#include "wm_internal.hpp"
#include "wm_bfgs.hpp"  // libm_sincosf: glibc's sinf / cosf for the device side

#include <math.h>
#include <string.h>

namespace wm {

constexpr unsigned kNdtChunkLog2 = 12;  // sharded NDT: ranks take turns in chunks of 4096 source points
constexpr int kNdtAcc = 28;  // score, 6 gradient entries, the 21 of the Hessian's upper triangle
constexpr int kNdtAccGrad = 7;  // score + gradient (the line search's passes)
__host__ __device__ constexpr int ndt_tri(int i, int j) {  // (i <= j) -> accumulator slot
    return 7 + i * 6 - i * (i - 1) / 2 + (j - i);
}

struct NdtVoxel {
    double mean[3];
    double icov[9];
};

// symmetric 3x3 eigen-decomposition (cyclic Jacobi), eigenvalues ascending
__device__ inline void sym_eig3(const double *Ain, double *evals, double *V) {
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        A[i] = Ain[i];
        V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-32 * diag || off == 0.0) break;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const int p = pr == 2 ? 1 : 0, q = pr == 0 ? 1 : 2;
            const double apq = A[p * 3 + q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double akp = A[k * 3 + p], akq = A[k * 3 + q];
                A[k * 3 + p] = c * akp - s * akq;
                A[k * 3 + q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                A[p * 3 + k] = c * apk - s * aqk;
                A[q * 3 + k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                V[k * 3 + p] = c * vkp - s * vkq;
                V[k * 3 + q] = s * vkp + c * vkq;
            }
        }
    }
    evals[0] = A[0];
    evals[1] = A[4];
    evals[2] = A[8];
    // ascending sort with column swaps (static indices)
#define SWAPCOL(a, b)                                  \
    if (evals[b] < evals[a]) {                         \
        double t = evals[a];                           \
        evals[a] = evals[b];                           \
        evals[b] = t;                                  \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) { \
            t = V[k * 3 + a];                          \
            V[k * 3 + a] = V[k * 3 + b];               \
            V[k * 3 + b] = t;                          \
        }                                              \
    }
    SWAPCOL(0, 1)
    SWAPCOL(0, 2)
    SWAPCOL(1, 2)
#undef SWAPCOL
}

__device__ inline bool inverse3(const double *m, double *o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
                 c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) ok = ok && isfinite(o[k]);
    return ok;
}

// mean, covariance, PCL's eigenvalue conditioning and the inverse of one voxel from its sums (s = sum p, pp = sum p p^T
// over `count` points); false: the voxel takes no part (fewer than min_points_per_voxel_ = 6 points, or a covariance that
// cannot be conditioned)
__device__ inline bool ndt_voxel_record(unsigned count, const double *s, const double *pp, NdtVoxel &v) {
    const double nn = (double) count;
    bool valid = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) v.mean[a] = s[a] / nn;
    if (count >= 6) {  // min_points_per_voxel_
        double cov[9], evals[3], evecs[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                cov[a * 3 + b] = ((pp[a * 3 + b] - 2.0 * (s[a] * v.mean[b])) / nn + v.mean[a] * v.mean[b]) *
                                 ((nn - 1.0) / nn);
        sym_eig3(cov, evals, evecs);
        if (!(evals[0] < 0 || evals[1] < 0 || evals[2] <= 0)) {
            const double minv = 0.01 * evals[2];
            if (evals[0] < minv) {
                evals[0] = minv;
                if (evals[1] < minv) evals[1] = minv;
                double einv[9], t[9];
                inverse3(evecs, einv);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) t[a * 3 + b] = evecs[a * 3 + b] * evals[b];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        double acc = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc += t[a * 3 + k] * einv[k * 3 + b];
                        cov[a * 3 + b] = acc;
                    }
            }
            valid = inverse3(cov, v.icov);
        }
    }
    return valid;
}

struct NdtDense {
    const int *table;  // nullptr: use the hash grid
    int i0, j0, k0;    // lattice origin (cell indices)
    int nx, ny, nz;
};

struct __attribute__((packed, aligned(4))) Int3 {  // three adjacent table cells, one 12-byte load
    int a, b, c;
};

struct NdtArgs {
    float Tf[12];
    float inv_res;
    double res2, d1, d2;
    float res2_f;  // the largest float d2 with (double) d2 < res2: the radius test in one float compare, same decisions
    // computeAngleDerivatives: 8 Jacobian and 15 Hessian 3-vectors
    double j[8][3];
    double h[15][3];
};

WM_HD void pose_to_matrix_f(const double p[6], float T[16]) {
#ifdef __HIP_DEVICE_COMPILE__  // (the device library's cosf / sinf are not glibc's: wm_bfgs.hpp)
    const float cx = libm_sincosf((float) p[3], 1), sx = libm_sincosf((float) p[3], 0);
    const float cy = libm_sincosf((float) p[4], 1), sy = libm_sincosf((float) p[4], 0);
    const float cz = libm_sincosf((float) p[5], 1), sz = libm_sincosf((float) p[5], 0);
#else
    const float cx = cosf((float) p[3]), sx = sinf((float) p[3]);
    const float cy = cosf((float) p[4]), sy = sinf((float) p[4]);
    const float cz = cosf((float) p[5]), sz = sinf((float) p[5]);
#endif
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

WM_HD void angle_derivatives(const double p[6], int pcl_d1_sign, NdtArgs *A) {
    double cx, cy, cz, sx, sy, sz;
    if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
    if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
    if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
    const double j[8][3] = {
        {-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy},  // a
        {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},    // b
        {-sy * cz, sy * sz, cy},                                       // c
        {sx * cy * cz, -sx * cy * sz, sx * sy},                        // d
        {-cx * cy * cz, cx * cy * sz, -cx * sy},                       // e
        {-cy * sz, -cy * cz, 0},                                       // f
        {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0},          // g
        {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};          // h
    const double h[15][3] = {
        {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy},   // a2
        {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},  // a3
        {cx * cy * cz, -cx * cy * sz, cx * sy},                        // b2
        {sx * cy * cz, -sx * cy * sz, sx * sy},                        // b3
        {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0},          // c2
        {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},          // c3
        {-cy * cz, cy * sz, pcl_d1_sign ? sy : -sy},                   // d1
        {-sx * sy * cz, sx * sy * sz, sx * cy},                        // d2
        {cx * sy * cz, -cx * sy * sz, -cx * cy},                       // d3
        {sy * sz, sy * cz, 0},                                         // e1
        {-sx * cy * sz, -sx * cy * cz, 0},                             // e2
        {cx * cy * sz, cx * cy * cz, 0},                               // e3
        {-cy * cz, cy * sz, 0},                                        // f1
        {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0},         // f2
        {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};        // f3
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 3; ++b) A->j[a][b] = j[a][b];
    for (int a = 0; a < 15; ++a)
        for (int b = 0; b < 3; ++b) A->h[a][b] = h[a][b];
}

}  // namespace wm
