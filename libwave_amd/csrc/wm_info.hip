// wm_info.hip -- libwave's post-match information-matrix estimators on device:
//   ICPMatcher::estimateLUM     wave_matching/src/icp_pcl_functions.cpp:182-289
//   ICPMatcher::estimateLUMold  wave_matching/src/icp_pcl_functions.cpp:51-179
//   ICPMatcher::estimateCensi   wave_matching/src/icp.cpp:167-397
// All three are O(n_corr) reductions over the final correspondences: LUM/LUMold to
// the 6x6 M^T M / 6x1 M^T Z normal equations (then a second pass for the residual
// s^2), Censi to d2J/dX2 (21 unique) and sum d2J/dZdX cov(z) d2J/dZdX^T (21 unique).
// The per-pair arithmetic keeps the reference's float/double mix (float pair
// averages / differences and float products feeding double accumulators); the sums
// themselves run in double in a fixed tree order (the reference sums sequentially,
// and its s^2 in float -- a ~1e-5 relative difference, documented in DESIGN.md).
#include "wm_internal.hpp"

#include <float.h>
#include <math.h>
#include <string.h>

namespace wm {

constexpr int kInfoAcc = 48;
constexpr int kInfoBlocks = 512;

__device__ __forceinline__ void xform_f(const float *T, const float4 &p, float &x, float &y,
                                        float &z) {
    x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], p.x), __fmul_rn(T[1], p.y)),
                            __fmul_rn(T[2], p.z)), T[3]);
    y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], p.x), __fmul_rn(T[5], p.y)),
                            __fmul_rn(T[6], p.z)), T[7]);
    z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], p.x), __fmul_rn(T[9], p.y)),
                            __fmul_rn(T[10], p.z)), T[11]);
}

template <int NACC>
__device__ __forceinline__ void block_reduce_store(double *a, double *partials) {
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    __shared__ double lds[kBlock / 64][NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NACC; ++k) lds[wave][k] = a[k];
    __syncthreads();
    if (threadIdx.x < NACC) {
        double s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += lds[w][threadIdx.x];
        partials[(size_t) blockIdx.x * kInfoAcc + threadIdx.x] = s;
    }
}

struct InfoArgs {
    float Tf[12];      // float transform giving PCL's `final` cloud from the source
    double D[6];       // LUM pose-difference estimate (second pass)
    double X[3];       // Censi: translation of the result
    double Rk[3][9];   // Censi: dR/dr, dR/dp, dR/dy of R = Rz(y) Ry(p) Rx(r) at the result's angles (row-major)
    double sd[6];      // Censi: square roots of diag(lin, ang, ang, lin, ang, ang)
    // sharded registration: only the pairs of the queries this rank OWNED when they were searched count
    // (transformed x, under the pose of that search, inside the rank's slab); the ranks' sums are added
    float Tg[12];
    int slab_on;
    float slab_lo, slab_hi;
};

__device__ __forceinline__ bool info_owned(const InfoArgs &A, const float4 &p) {
    if (!A.slab_on) return true;
    float gx, gy, gz;
    xform_f(A.Tg, p, gx, gy, gz);
    return gx >= A.slab_lo && gx < A.slab_hi;
}

// pass 1: a[0]=n, a[1..3]=sum av, a[4..9]=MM(3,4),(3,5),(4,5),(3,3),(4,4),(5,5) terms,
//         a[10..15]=MZ(0..5)
__global__ void __launch_bounds__(kBlock)
    k_lum_sums(const float4 *__restrict__ src, unsigned n,
               const unsigned long long *__restrict__ keys, const float4 *__restrict__ tgt,
               InfoArgs A, double *__restrict__ partials) {
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 0.0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned idx = (unsigned) keys[i];
        if (idx == kNoIdx || !info_owned(A, src[i])) continue;
        float px, py, pz;
        xform_f(A.Tf, src[i], px, py, pz);
        const float4 q = tgt[idx];
        const float av0 = __fmul_rn(0.5f, __fadd_rn(px, q.x)), av1 = __fmul_rn(0.5f, __fadd_rn(py, q.y)),
                    av2 = __fmul_rn(0.5f, __fadd_rn(pz, q.z));
        const float df0 = __fsub_rn(px, q.x), df1 = __fsub_rn(py, q.y), df2 = __fsub_rn(pz, q.z);
        a[0] += 1.0;
        a[1] += av0;
        a[2] += av1;
        a[3] += av2;
        a[4] += __fmul_rn(av0, av2);                                           // -MM(3,4)
        a[5] += __fmul_rn(av0, av1);                                           // -MM(3,5)
        a[6] += __fmul_rn(av1, av2);                                           // -MM(4,5)
        a[7] += __fadd_rn(__fmul_rn(av1, av1), __fmul_rn(av2, av2));           // MM(3,3)
        a[8] += __fadd_rn(__fmul_rn(av0, av0), __fmul_rn(av1, av1));           // MM(4,4)
        a[9] += __fadd_rn(__fmul_rn(av0, av0), __fmul_rn(av2, av2));           // MM(5,5)
        a[10] += df0;
        a[11] += df1;
        a[12] += df2;
        a[13] += __fsub_rn(__fmul_rn(av1, df2), __fmul_rn(av2, df1));
        a[14] += __fsub_rn(__fmul_rn(av0, df1), __fmul_rn(av1, df0));
        a[15] += __fsub_rn(__fmul_rn(av2, df0), __fmul_rn(av0, df2));
    }
    block_reduce_store<16>(a, partials);
}

// pass 2: a[0] = sum |diff - (D_t + av x D_r)|^2
__global__ void __launch_bounds__(kBlock)
    k_lum_ss(const float4 *__restrict__ src, unsigned n, const unsigned long long *__restrict__ keys,
             const float4 *__restrict__ tgt, InfoArgs A, double *__restrict__ partials) {
    double a[1] = {0.0};
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned idx = (unsigned) keys[i];
        if (idx == kNoIdx || !info_owned(A, src[i])) continue;
        float px, py, pz;
        xform_f(A.Tf, src[i], px, py, pz);
        const float4 q = tgt[idx];
        const float av0 = __fmul_rn(0.5f, __fadd_rn(px, q.x)), av1 = __fmul_rn(0.5f, __fadd_rn(py, q.y)),
                    av2 = __fmul_rn(0.5f, __fadd_rn(pz, q.z));
        const float df0 = __fsub_rn(px, q.x), df1 = __fsub_rn(py, q.y), df2 = __fsub_rn(pz, q.z);
        const double e0 = df0 - (A.D[0] + av2 * A.D[5] - av1 * A.D[4]);
        const double e1 = df1 - (A.D[1] + av0 * A.D[4] - av2 * A.D[3]);
        const double e2 = df2 - (A.D[2] + av1 * A.D[3] - av0 * A.D[5]);
        a[0] += (double) (float) (e0 * e0 + e1 * e1 + e2 * e2);
    }
    block_reduce_store<1>(a, partials);
}

// ---- Censi's covariance (estimateCensi, icp.cpp:167-397), from the STRUCTURE of its cost, not from its expanded text.
// The reference's J = sum |e|^2, e = t + R(r, p, y) a - b  over the pairs (a = matched target point Z1..Z3, b = source
// point Z4..Z6, x = (t, r, p, y), R = Rz(y) Ry(p) Rx(r)).  It carries machine-expanded expressions of d2J/dx2 and
// d2J/dz dx (130 lines of products of sines and cosines).  Differentiating e instead, with R_k = dR/d theta_k and
// R_kl = d2R/d theta_k d theta_l (nine + eighteen numbers per launch, formed once on the host):
//     d2J/dx2   = 2 [[ n I,  R_k (sum a) ], [ . ,  Theta ]],   Theta_kl = sum (t - b) . (R_kl a)
//                 (the a-quadratic terms of 2 d_k.d_l + 2 e.(R_kl a) cancel identically: differentiate R^T R = I twice)
//     d2J/dz dx = [ P | Q ],   P = [2 R^T ; -2 I]  (the same for every pair),
//                 Q = [ 2 (R_r^T c, R_p^T c, R_y^T c) ; -2 (R_r a, R_p a, R_y a) ],  c = t - b
//                 (again R^T R_k + R_k^T R = 0 removes every a-dependent term of the upper block)
//     middle    = sum G cov_Z G^T = P (sum C_a) P^T + sum Q C_b Q^T,   cov_Z = diag(C_a, C_b) (spherical Jacobians)
// so the Hessian needs only the sums n, sum a, sum b a^T (thirteen: ICP's own statistics), the constant block only
// sum C_a (six), and the per-pair work is Q C_b Q^T = (Q K_b)(Q K_b)^T with K_b = J_b sqrt(diag): ~250 f64 operations
// on forty accumulators and no 36-element temporaries -- 394 registers (one wave per SIMD) before.  Against the oracle's
// literal restatement of the reference's text: 5e-10 of the largest entry on both matrices (its float sub-products, e.g.
// 2 * Z3 * Z4 formed in float, are double here), 1e-6 relative on the information matrix (tests/test_info_gpu.py).
// sums: [0] n, [1..3] sum a, [4..12] sum b_i a_j, [13..18] sum C_a (upper), [19..39] sum (Q/2) C_b (Q/2)^T (upper)
constexpr int kCensiSums = 40;

__device__ __forceinline__ void censi_sph_jacobian(float x, float y, float z, const double *sd /* sqrt of 3 variances */,
                                                   double (&K)[3][3]) {
    // d(point)/d(range, bearing, azimuth) as the reference forms it (icp.cpp:217-247: float range / angles from the
    // float coordinates, double sines and cosines of them), columns scaled by the standard deviations
    const double rg = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    const double br = atan2f(y, x);
    const double az = atanf(__fdiv_rn(z, sqrtf(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)))));
    const double cb = cos(br), sb = sin(br), ca = cos(az), sa = sin(az);
    K[0][0] = cb * sa * sd[0];
    K[1][0] = sb * sa * sd[0];
    K[2][0] = ca * sd[0];
    K[0][1] = -rg * sb * sa * sd[1];
    K[1][1] = rg * cb * sa * sd[1];
    K[2][1] = 0.0;
    K[0][2] = rg * cb * ca * sd[2];
    K[1][2] = rg * ca * sb * sd[2];
    K[2][2] = -rg * sa * sd[2];
}

// (two halves, blockIdx.y says which: the statistics and sum C_a -- the target points' trigonometry --, or the per-pair
// products -- the source points' --: forty accumulators plus a pair's temporaries in one body took 173 registers)
template <int PART>
__device__ __forceinline__ void censi_part(const float4 *__restrict__ src, unsigned n, const unsigned long long *__restrict__ keys,
                                           const float4 *__restrict__ tgt, const InfoArgs &A, double *__restrict__ partials) {
    constexpr int kN = PART == 0 ? 19 : 21;
    double acc[kN];
#pragma unroll
    for (int k = 0; k < kN; ++k) acc[k] = 0.0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned idx = (unsigned) keys[i];
        if (idx == kNoIdx || !info_owned(A, src[i])) continue;
        const float4 t4 = tgt[idx], s4 = src[i];
        const double a[3] = {t4.x, t4.y, t4.z}, b[3] = {s4.x, s4.y, s4.z};
        double K[3][3];
        if constexpr (PART == 0) {
            acc[0] += 1.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[1 + j] += a[j];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[4 + 3 * r + j] += b[r] * a[j];
            censi_sph_jacobian(t4.x, t4.y, t4.z, A.sd, K);
            int u = 13;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = r; c < 3; ++c) acc[u++] += K[r][0] * K[c][0] + K[r][1] * K[c][1] + K[r][2] * K[c][2];
        } else {
            censi_sph_jacobian(s4.x, s4.y, s4.z, A.sd + 3, K);
            // Q / 2: rows 0..2 = R_k^T c, rows 3..5 = -R_k a (column k), then V = (Q / 2) K_b
            const double c[3] = {A.X[0] - b[0], A.X[1] - b[1], A.X[2] - b[2]};
            double V[6][3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                double q1[3], q2[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double *Rk = A.Rk[k];
                    q1[k] = Rk[0 * 3 + m] * c[0] + Rk[1 * 3 + m] * c[1] + Rk[2 * 3 + m] * c[2];
                    q2[k] = -(Rk[m * 3 + 0] * a[0] + Rk[m * 3 + 1] * a[1] + Rk[m * 3 + 2] * a[2]);
                }
#pragma unroll
                for (int sidx = 0; sidx < 3; ++sidx) {
                    V[m][sidx] = q1[0] * K[0][sidx] + q1[1] * K[1][sidx] + q1[2] * K[2][sidx];
                    V[3 + m][sidx] = q2[0] * K[0][sidx] + q2[1] * K[1][sidx] + q2[2] * K[2][sidx];
                }
            }
            int u = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = r; cc < 6; ++cc) acc[u++] += V[r][0] * V[cc][0] + V[r][1] * V[cc][1] + V[r][2] * V[cc][2];
        }
    }
    block_reduce_store<kN>(acc, partials + (PART == 0 ? 0 : 19));
}

__global__ void __launch_bounds__(kBlock)
    k_censi(const float4 *__restrict__ src, unsigned n, const unsigned long long *__restrict__ keys,
            const float4 *__restrict__ tgt, InfoArgs A, double *__restrict__ partials) {
    if (blockIdx.y == 0) censi_part<0>(src, n, keys, tgt, A, partials);
    else censi_part<1>(src, n, keys, tgt, A, partials);
}

// the rotation of estimateCensi's parametrisation, R = Rz(y) Ry(p) Rx(r), with its first and second derivatives in
// (r, p, y): Rk[k] = dR/d theta_k, Rkl[k][l] (k <= l) = d2R/d theta_k d theta_l, row-major 3 x 3
static void censi_rotation_derivatives(const double e[3], double R[9], double Rk[3][9], double Rkl[3][3][9]) {
    auto mul = [](const double *a, const double *b, double *o) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    };
    // order d of the elementary rotation about axis ax (0: x, 1: y, 2: z): d/dt twice = minus the rotating part
    auto elem = [](int ax, double t, int d, double *o) {
        const double c = cos(t), s = sin(t);
        const double c0 = d == 0 ? c : (d == 1 ? -s : -c), s0 = d == 0 ? s : (d == 1 ? c : -s);  // (cos, sin)^(d)
        const double one = d == 0 ? 1.0 : 0.0;
        const double X[9] = {one, 0, 0, 0, c0, -s0, 0, s0, c0};
        const double Y[9] = {c0, 0, s0, 0, one, 0, -s0, 0, c0};
        const double Z[9] = {c0, -s0, 0, s0, c0, 0, 0, 0, one};
        const double *m = ax == 0 ? X : (ax == 1 ? Y : Z);
        for (int k = 0; k < 9; ++k) o[k] = m[k];
    };
    auto compose = [&](int dx, int dy, int dz, double *o) {  // Rz^(dz)(y) Ry^(dy)(p) Rx^(dx)(r)
        double x[9], y[9], z[9], zy[9];
        elem(0, e[0], dx, x);
        elem(1, e[1], dy, y);
        elem(2, e[2], dz, z);
        mul(z, y, zy);
        mul(zy, x, o);
    };
    compose(0, 0, 0, R);
    for (int k = 0; k < 3; ++k) compose(k == 0, k == 1, k == 2, Rk[k]);
    for (int k = 0; k < 3; ++k)
        for (int l = k; l < 3; ++l) compose((k == 0) + (l == 0), (k == 1) + (l == 1), (k == 2) + (l == 2), Rkl[k][l]);
}

// d2J_dX2 and `middle` (full symmetric 6 x 6) from the kernel's forty sums
static void censi_assemble(const double *s, const double t[3], const double R[9], const double Rk[3][9],
                           const double Rkl[3][3][9], double H[36], double Mid[36]) {
    for (int k = 0; k < 36; ++k) H[k] = Mid[k] = 0.0;
    const double n = s[0], *Sa = s + 1, *Sba = s + 4;
    for (int i = 0; i < 3; ++i) H[i * 6 + i] = 2.0 * n;
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i)
            H[i * 6 + 3 + k] = 2.0 * (Rk[k][i * 3] * Sa[0] + Rk[k][i * 3 + 1] * Sa[1] + Rk[k][i * 3 + 2] * Sa[2]);
    for (int k = 0; k < 3; ++k)
        for (int l = k; l < 3; ++l) {
            const double *M = Rkl[k][l];
            double th = 0.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) th += M[i * 3 + j] * (t[i] * Sa[j] - Sba[i * 3 + j]);
            H[(3 + k) * 6 + 3 + l] = 2.0 * th;
        }
    for (int r = 0; r < 6; ++r)
        for (int c = r + 1; c < 6; ++c) H[c * 6 + r] = H[r * 6 + c];
    // middle = P (sum C_a) P^T + 4 sum (Q/2) C_b (Q/2)^T,   P = [2 R^T ; -2 I]
    double Ca[9], P[18], PC[18];
    {
        int u = 13;
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) Ca[r * 3 + c] = Ca[c * 3 + r] = s[u++];
    }
    for (int m = 0; m < 3; ++m)
        for (int j = 0; j < 3; ++j) {
            P[m * 3 + j] = 2.0 * R[j * 3 + m];
            P[(3 + m) * 3 + j] = m == j ? -2.0 : 0.0;
        }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c) PC[r * 3 + c] = P[r * 3] * Ca[c] + P[r * 3 + 1] * Ca[3 + c] + P[r * 3 + 2] * Ca[6 + c];
    int u = 19;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
            const double pcp = PC[r * 3] * P[c * 3] + PC[r * 3 + 1] * P[c * 3 + 1] + PC[r * 3 + 2] * P[c * 3 + 2];
            Mid[r * 6 + c] = Mid[c * 6 + r] = pcp + 4.0 * s[u++];
        }
}

static int reduce_partials(wm_ctx *ctx, int nblocks, int nacc, double *out, wm_comm *comm = nullptr) {
    std::vector<double> h((size_t) nblocks * kInfoAcc);
    WM_TRY(copy_to_caller(ctx, h.data(), ctx->partials.p, h.size() * sizeof(double)));
    for (int k = 0; k < nacc; ++k) {
        double s = 0;
        for (int b = 0; b < nblocks; ++b) s += h[(size_t) b * kInfoAcc + k];  // fixed order
        out[k] = s;
    }
    if (comm) {  // sharded: the ranks' sums, added by the exchange (every rank gets the same totals)
        WM_HIP(ctx, ctx->shard_stats.reserve(64 * sizeof(double)));
        WM_HIP(ctx, hipMemcpyAsync(ctx->shard_stats.p, out, (size_t) nacc * sizeof(double), hipMemcpyHostToDevice,
                                   ctx->stream));
        WM_TRY(comm_allreduce(ctx, comm, ctx->shard_stats.as<double>(), nacc));
        WM_TRY(copy_to_caller(ctx, out, ctx->shard_stats.p, (size_t) nacc * sizeof(double)));
    }
    return WM_OK;
}

static int info_blocks(size_t n) {
    size_t b = (n + kBlock - 1) / kBlock;
    if (b > kInfoBlocks) b = kInfoBlocks;
    return b < 1 ? 1 : (int) b;
}

// Eigen 3.3 MatrixBase::eulerAngles(0,1,2) (reference call: icp.cpp:175)
static void euler_012(const double *T, double e[3]) {
    auto R = [&](int i, int j) { return T[i * 4 + j]; };
    e[0] = atan2(R(1, 2), R(2, 2));
    const double c2 = sqrt(R(0, 0) * R(0, 0) + R(0, 1) * R(0, 1));
    if (e[0] > 0.0) {
        e[0] -= M_PI;
        e[1] = atan2(-R(0, 2), -c2);
    } else {
        e[1] = atan2(-R(0, 2), c2);
    }
    const double s1 = sin(e[0]), c1 = cos(e[0]);
    e[2] = atan2(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
    e[0] = -e[0];
    e[1] = -e[1];
    e[2] = -e[2];
}

static int lum_from_current_keys(wm_ctx *ctx, const InfoArgs &args0, double info[36],
                                 bool lumold_quirk, wm_comm *comm = nullptr) {
    InfoArgs args = args0;
    const unsigned n = (unsigned) ctx->n_src;
    const int nb = info_blocks(n);
    WM_HIP(ctx, ctx->partials.reserve((size_t) kInfoBlocks * kInfoAcc * sizeof(double)));
    double *partials = ctx->partials.as<double>();
    const float4 *src = ctx->src_sorted.as<float4>(), *tgt = ctx->tgt_orig.as<float4>();
    const unsigned long long *keys = ctx->keys.as<unsigned long long>();
    hipLaunchKernelGGL(k_lum_sums, dim3(nb), dim3(kBlock), 0, ctx->stream, src, n, keys, tgt, args,
                       partials);
    WM_HIP(ctx, hipGetLastError());
    double a[16];
    WM_TRY(reduce_partials(ctx, nb, 16, a, comm));
    double MM[36] = {0}, MZ[6], MMinv[36], D[6];
#define M_(r, c) MM[(r) * 6 + (c)]
    M_(0, 4) = -a[2];
    M_(0, 5) = a[3];
    M_(1, 3) = -a[3];
    M_(1, 4) = a[1];
    M_(2, 3) = a[2];
    M_(2, 5) = -a[1];
    M_(3, 4) = -a[4];
    M_(3, 5) = -a[5];
    M_(4, 5) = -a[6];
    M_(3, 3) = a[7];
    M_(4, 4) = a[8];
    M_(5, 5) = a[9];
    M_(0, 0) = M_(1, 1) = M_(2, 2) = (double) (float) (int) a[0];
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
    for (int k = 0; k < 6; ++k) MZ[k] = a[10 + k];
    inverse<6>(MM, MMinv);
    for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += MMinv[r * 6 + c] * MZ[c];
        D[r] = s;
        args.D[r] = s;
    }
    hipLaunchKernelGGL(k_lum_ss, dim3(nb), dim3(kBlock), 0, ctx->stream, src, n, keys, tgt, args,
                       partials);
    WM_HIP(ctx, hipGetLastError());
    double ssd[1];
    WM_TRY(reduce_partials(ctx, nb, 1, ssd, comm));
    const float ss = (float) ssd[0];
    const bool bad = (ss < 0.0000000000001f || !isfinite(ss));
    if (bad && !lumold_quirk) {  // estimateLUM: identity + return (icp_pcl_functions.cpp:281-285)
        for (int k = 0; k < 36; ++k) info[k] = (k % 7 == 0) ? 1.0 : 0.0;
        return 1;
    }
    // estimateLUMold falls through its failure branch (icp_pcl_functions.cpp:170-178)
    const float inv = 1.0f / ss;
    for (int k = 0; k < 36; ++k) info[k] = MM[k] * inv;
    return bad ? 1 : 0;
}

}  // namespace wm

using namespace wm;

static int icp_info_impl(wm_ctx *ctx, wm_comm *comm, int method, const double T_result[16], double lin_covar,
                         double ang_covar, double max_corr, double info[36], int *degenerate) {
    if (!ctx || !info || method < WM_INFO_LUM || method > WM_INFO_LUMOLD) return WM_ERR_ARG;
    if (degenerate) *degenerate = 0;
    if (!ctx->have_corr || !ctx->last_align_valid) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    InfoArgs args;
    memset(&args, 0, sizeof(args));
    for (int k = 0; k < 12; ++k) args.Tf[k] = (float) ctx->corr_T[k];
    const bool sharded = comm != nullptr && ctx->last_align_sharded;
    if (comm != nullptr && !sharded) return WM_ERR_STATE;
    if (sharded) {  // the align's own correspondences: owned under the pose of its last search
        args.slab_on = 1;
        args.slab_lo = ctx->shard_lo;
        args.slab_hi = ctx->shard_hi;
        for (int k = 0; k < 12; ++k) args.Tg[k] = ctx->h_state->Tf_search[k];
    } else if (ctx->last_align_sharded) {
        return WM_ERR_STATE;  // a rank's correspondences alone say nothing: wm_icp_info_sharded
    }
    if (method == WM_INFO_LUM) {
        if (!ctx->last_align_converged) return WM_NOT_CONVERGED;  // information left untouched
        const int rc = lum_from_current_keys(ctx, args, info, false, sharded ? comm : nullptr);
        if (rc < 0) return rc;
        if (degenerate) *degenerate = rc;
        return WM_OK;
    }
    if (method == WM_INFO_LUMOLD) {
        if (!(max_corr > 0)) return WM_ERR_ARG;
        // fresh exact NN of the aligned cloud against the target, d2 < max_corr^2
        // (icp_pcl_functions.cpp:67-101); the align's own correspondences are kept
        const size_t kb = ctx->n_src * sizeof(unsigned long long);
        WM_HIP(ctx, ctx->keys_bak.reserve(kb > 0 ? kb : 8));
        WM_HIP(ctx, hipMemcpyAsync(ctx->keys_bak.p, ctx->keys.p, kb, hipMemcpyDeviceToDevice,
                                   ctx->stream));
        const size_t mb = ctx->n_src * sizeof(float4);
        WM_HIP(ctx, ctx->match_pt_bak.reserve(mb > 0 ? mb : 16));
        WM_HIP(ctx, hipMemcpyAsync(ctx->match_pt_bak.p, ctx->match_pt.p, mb, hipMemcpyDeviceToDevice,
                                   ctx->stream));
        // (sharded: the fresh search runs under the final pose, and so does its ownership test)
        if (sharded)
            for (int k = 0; k < 12; ++k) args.Tg[k] = args.Tf[k];
        const float Tf_keep_lo = ctx->shard_lo, Tf_keep_hi = ctx->shard_hi;
        int rc = nn_pass(ctx, ctx->corr_T, threshold_d2_strict(max_corr), max_corr, true, sharded, Tf_keep_lo, Tf_keep_hi);
        if (rc == WM_OK) rc = lum_from_current_keys(ctx, args, info, true, sharded ? comm : nullptr);
        WM_HIP(ctx, hipMemcpyAsync(ctx->keys.p, ctx->keys_bak.p, kb, hipMemcpyDeviceToDevice,
                                   ctx->stream));
        WM_HIP(ctx, hipMemcpyAsync(ctx->match_pt.p, ctx->match_pt_bak.p, mb, hipMemcpyDeviceToDevice,
                                   ctx->stream));
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (rc < 0) return rc;
        if (degenerate) *degenerate = rc;
        return WM_OK;
    }
    // Censi
    if (!T_result) return WM_ERR_ARG;
    if (!ctx->last_align_converged) return WM_NOT_CONVERGED;
    double e[3], R[9], Rk[3][9], Rkl[3][3][9];
    euler_012(T_result, e);
    censi_rotation_derivatives(e, R, Rk, Rkl);
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 9; ++j) args.Rk[k][j] = Rk[k][j];
    args.X[0] = T_result[3];
    args.X[1] = T_result[7];
    args.X[2] = T_result[11];
    const double sph[6] = {lin_covar, ang_covar, ang_covar, lin_covar, ang_covar, ang_covar};
    for (int k = 0; k < 6; ++k) args.sd[k] = sqrt(sph[k]);
    const unsigned n = (unsigned) ctx->n_src;
    const int nb = info_blocks(n);
    WM_HIP(ctx, ctx->partials.reserve((size_t) kInfoBlocks * kInfoAcc * sizeof(double)));
    hipLaunchKernelGGL(k_censi, dim3(nb, 2), dim3(kBlock), 0, ctx->stream,
                       ctx->src_sorted.as<float4>(), n, ctx->keys.as<unsigned long long>(),
                       ctx->tgt_orig.as<float4>(), args, ctx->partials.as<double>());
    WM_HIP(ctx, hipGetLastError());
    double a[kCensiSums];
    WM_TRY(reduce_partials(ctx, nb, kCensiSums, a, sharded ? comm : nullptr));
    double H[36], Mid[36], Hinv[36], t1[36], t2[36];
    censi_assemble(a, args.X, R, Rk, Rkl, H, Mid);
    inverse<6>(H, Hinv);
    mat_mul<6>(Hinv, Mid, t1);
    mat_mul<6>(t1, Hinv, t2);
    inverse<6>(t2, info);
    return WM_OK;
}

extern "C" {

int wm_icp_info(wm_ctx *ctx, int method, const double T_result[16], double lin_covar,
                double ang_covar, double max_corr, double info[36], int *degenerate) {
    return icp_info_impl(ctx, nullptr, method, T_result, lin_covar, ang_covar, max_corr, info, degenerate);
}

int wm_icp_info_sharded(wm_ctx *ctx, wm_comm *comm, int method, const double T_result[16], double lin_covar,
                        double ang_covar, double max_corr, double info[36], int *degenerate) {
    if (!comm) return WM_ERR_ARG;
    return icp_info_impl(ctx, comm, method, T_result, lin_covar, ang_covar, max_corr, info, degenerate);
}

}  // extern "C"
