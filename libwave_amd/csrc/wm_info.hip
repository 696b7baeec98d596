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
    double cr, sr, cp, sp, cy, sy;
    double sph[6];     // Censi: diag(lin, ang, ang, lin, ang, ang)
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

// Censi: a[0..20] = upper triangle of d2J_dX2, a[21..41] = upper triangle of `middle`
__global__ void __launch_bounds__(kBlock)
    k_censi(const float4 *__restrict__ src, unsigned n, const unsigned long long *__restrict__ keys,
            const float4 *__restrict__ tgt, InfoArgs A, double *__restrict__ partials) {
    double a[42];
#pragma unroll
    for (int k = 0; k < 42; ++k) a[k] = 0.0;
    const double cr = A.cr, sr = A.sr, cp = A.cp, sp = A.sp, cy = A.cy, sy = A.sy;
    const double X1 = A.X[0], X2 = A.X[1], X3 = A.X[2];
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned idx = (unsigned) keys[i];
        if (idx == kNoIdx || !info_owned(A, src[i])) continue;
        const float4 t4 = tgt[idx], s4 = src[i];
        const float Z1 = t4.x, Z2 = t4.y, Z3 = t4.z, Z4 = s4.x, Z5 = s4.y, Z6 = s4.z;
        // spherical-coordinate Jacobians of both points (icp.cpp:225-247)
        double j[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) j[k] = 0.0;
        {
            double rg = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(Z1, Z1), __fmul_rn(Z2, Z2)), __fmul_rn(Z3, Z3)));
            double br = atan2f(Z2, Z1);
            double az = atanf(__fdiv_rn(Z3, sqrtf(__fadd_rn(__fmul_rn(Z1, Z1), __fmul_rn(Z2, Z2)))));
            const double cb = cos(br), sb = sin(br), ca = cos(az), sa = sin(az);
            j[0 * 6 + 0] = cb * sa;
            j[1 * 6 + 0] = sb * sa;
            j[2 * 6 + 0] = ca;
            j[0 * 6 + 1] = -rg * sb * sa;
            j[1 * 6 + 1] = rg * cb * sa;
            j[0 * 6 + 2] = rg * cb * ca;
            j[1 * 6 + 2] = rg * ca * sb;
            j[2 * 6 + 2] = -rg * sa;
        }
        {
            double rg = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(Z4, Z4), __fmul_rn(Z5, Z5)), __fmul_rn(Z6, Z6)));
            double br = atan2f(Z5, Z4);
            double az = atanf(__fdiv_rn(Z6, sqrtf(__fadd_rn(__fmul_rn(Z4, Z4), __fmul_rn(Z5, Z5)))));
            const double cb = cos(br), sb = sin(br), ca = cos(az), sa = sin(az);
            j[3 * 6 + 3] = cb * sa;
            j[4 * 6 + 3] = sb * sa;
            j[5 * 6 + 3] = ca;
            j[3 * 6 + 4] = -rg * sb * sa;
            j[4 * 6 + 4] = rg * cb * sa;
            j[3 * 6 + 5] = rg * cb * ca;
            j[4 * 6 + 5] = rg * ca * sb;
            j[5 * 6 + 5] = -rg * sa;
        }
        // cov_Z = j * diag(sph) * j^T: block diagonal (two 3x3 blocks)
        double cz[36];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) s += j[r * 6 + k] * A.sph[k] * j[c * 6 + k];
                cz[r * 6 + c] = s;
            }
        // d2J_dX2, upper triangle (icp.cpp:258-312); float operands keep the
        // reference's float sub-products (e.g. 2 * Z3 * Z4 is formed in float)
        const double w1 = sr * sy + cr * cy * sp, w2 = cr * sy - cy * sr * sp;
        const double w3 = cy * sr - cr * sp * sy, w4 = cr * cy + sr * sp * sy;
        int u = 0;
        // row 0
        a[u++] += 2;
        u += 2;  // (0,1), (0,2) stay zero
        a[u++] += 2 * Z2 * w1 + 2 * Z3 * w2;
        a[u++] += 2 * cy * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr);
        a[u++] += 2 * Z3 * w3 - 2 * Z2 * w4 - 2 * Z1 * cp * sy;
        // row 1
        a[u++] += 2;
        u += 1;  // (1,2)
        a[u++] += -2 * Z2 * w3 - 2 * Z3 * w4;
        a[u++] += 2 * sy * (Z3 * cr * cp - Z1 * sp + Z2 * cp * sr);
        a[u++] += 2 * Z3 * w1 - 2 * Z2 * w2 + 2 * Z1 * cp * cy;
        // row 2
        a[u++] += 2;
        a[u++] += 2 * cp * (Z2 * cr - Z3 * sr);
        a[u++] += -2 * Z1 * cp - 2 * Z3 * cr * sp - 2 * Z2 * sr * sp;
        u += 1;  // (2,5) is zero
        // row 3
        a[u++] += (2 * Z2 * w2 - 2 * Z3 * w1) * (X1 - Z4 - Z2 * w2 + Z3 * w1 + Z1 * cp * cy) -
                  (2 * Z2 * w4 - 2 * Z3 * w3) * (X2 - Z5 + Z2 * w4 - Z3 * w3 + Z1 * cp * sy) -
                  (2 * Z3 * cr * cp + 2 * Z2 * cp * sr) * (X3 - Z6 - Z1 * sp + Z3 * cr * cp + Z2 * cp * sr) +
                  (Z2 * w1 + Z3 * w2) * (2 * Z2 * w1 + 2 * Z3 * w2) +
                  (Z2 * w3 + Z3 * w4) * (2 * Z2 * w3 + 2 * Z3 * w4) +
                  (Z2 * cr * cp - Z3 * cp * sr) * (2 * Z2 * cr * cp - 2 * Z3 * cp * sr);
        a[u++] += -2 * (Z2 * cr - Z3 * sr) *
                  (X3 * sp - Z6 * sp - X1 * cp * cy + Z4 * cp * cy - X2 * cp * sy + Z5 * cp * sy);
        a[u++] += 2 * X1 * Z3 * cr * cy - 2 * Z3 * Z4 * cr * cy + 2 * X1 * Z2 * cy * sr +
                  2 * X2 * Z3 * cr * sy - 2 * Z2 * Z4 * cy * sr - 2 * Z3 * Z5 * cr * sy +
                  2 * X2 * Z2 * sr * sy - 2 * Z2 * Z5 * sr * sy + 2 * X2 * Z2 * cr * cy * sp -
                  2 * Z2 * Z5 * cr * cy * sp - 2 * X1 * Z2 * cr * sp * sy - 2 * X2 * Z3 * cy * sr * sp +
                  2 * Z2 * Z4 * cr * sp * sy + 2 * Z3 * Z5 * cy * sr * sp + 2 * X1 * Z3 * sr * sp * sy -
                  2 * Z3 * Z4 * sr * sp * sy;
        // row 4
        {
            const double k1 = Z3 * cr * cp - Z1 * sp + Z2 * cp * sr;
            const double k2 = Z1 * cp + Z3 * cr * sp + Z2 * sr * sp;
            a[u++] += k2 * (2 * Z1 * cp + 2 * Z3 * cr * sp + 2 * Z2 * sr * sp) -
                      (2 * Z3 * cr * cp - 2 * Z1 * sp + 2 * Z2 * cp * sr) *
                          (X3 - Z6 - Z1 * sp + Z3 * cr * cp + Z2 * cp * sr) +
                      2 * cy * cy * pow(k1, 2) + 2 * sy * sy * pow(k1, 2) -
                      2 * cy * k2 *
                          (X1 - Z4 + Z1 * cp * cy - Z2 * cr * sy + Z3 * sr * sy + Z2 * cy * sr * sp +
                           Z3 * cr * cy * sp) -
                      2 * sy * k2 *
                          (X2 - Z5 + Z2 * cr * cy + Z1 * cp * sy - Z3 * cy * sr + Z3 * cr * sp * sy +
                           Z2 * sr * sp * sy);
            a[u++] += 2 * k1 * (X2 * cy - Z5 * cy - X1 * sy + Z4 * sy);
        }
        // row 5
        a[u++] += 2 * Z1 * Z4 * cp * cy - 2 * X2 * Z2 * cr * cy - 2 * X1 * Z1 * cp * cy +
                  2 * Z2 * Z5 * cr * cy + 2 * X1 * Z2 * cr * sy - 2 * X2 * Z1 * cp * sy +
                  2 * X2 * Z3 * cy * sr - 2 * Z2 * Z4 * cr * sy + 2 * Z1 * Z5 * cp * sy -
                  2 * Z3 * Z5 * cy * sr - 2 * X1 * Z3 * sr * sy + 2 * Z3 * Z4 * sr * sy -
                  2 * X1 * Z3 * cr * cy * sp + 2 * Z3 * Z4 * cr * cy * sp - 2 * X1 * Z2 * cy * sr * sp -
                  2 * X2 * Z3 * cr * sp * sy + 2 * Z2 * Z4 * cy * sr * sp + 2 * Z3 * Z5 * cr * sp * sy -
                  2 * X2 * Z2 * sr * sp * sy + 2 * Z2 * Z5 * sr * sp * sy;

        // d2J_dZdX (icp.cpp:322-386)
        double G[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) G[k] = 0.0;
#define G_(r, c) G[(r) * 6 + (c)]
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
                   2 * X1 * sr * sy - 2 * Z4 * sr * sy + 2 * X2 * cr * sp * sy - 2 * Z5 * cr * sp * sy +
                   2 * X1 * cr * cy * sp - 2 * Z4 * cr * cy * sp;
        G_(2, 3) = 2 * Z5 * cr * cy - 2 * X2 * cr * cy + 2 * X1 * cr * sy - 2 * X3 * cp * sr -
                   2 * Z4 * cr * sy + 2 * Z6 * cp * sr - 2 * X1 * cy * sr * sp + 2 * Z4 * cy * sr * sp -
                   2 * X2 * sr * sp * sy + 2 * Z5 * sr * sp * sy;
        G_(3, 3) = -2 * Z2 * w1 - 2 * Z3 * w2;
        G_(4, 3) = 2 * Z2 * w3 + 2 * Z3 * w4;
        G_(5, 3) = -2 * cp * (Z2 * cr - Z3 * sr);
        G_(0, 4) = 2 * Z6 * cp - 2 * X3 * cp - 2 * X1 * cy * sp + 2 * Z4 * cy * sp - 2 * X2 * sp * sy +
                   2 * Z5 * sp * sy;
        {
            const double k3 = X3 * sp - Z6 * sp - X1 * cp * cy + Z4 * cp * cy - X2 * cp * sy + Z5 * cp * sy;
            const double k1 = Z3 * cr * cp - Z1 * sp + Z2 * cp * sr;
            G_(1, 4) = -2 * sr * k3;
            G_(2, 4) = -2 * cr * k3;
            G_(3, 4) = -2 * cy * k1;
            G_(4, 4) = -2 * sy * k1;
        }
        G_(5, 4) = 2 * Z1 * cp + 2 * Z3 * cr * sp + 2 * Z2 * sr * sp;
        G_(0, 5) = 2 * cp * (X2 * cy - Z5 * cy - X1 * sy + Z4 * sy);
        G_(1, 5) = 2 * Z4 * cr * cy - 2 * X1 * cr * cy - 2 * X2 * cr * sy + 2 * Z5 * cr * sy +
                   2 * X2 * cy * sr * sp - 2 * Z5 * cy * sr * sp - 2 * X1 * sr * sp * sy +
                   2 * Z4 * sr * sp * sy;
        G_(2, 5) = 2 * X1 * cy * sr - 2 * Z4 * cy * sr + 2 * X2 * sr * sy - 2 * Z5 * sr * sy -
                   2 * X1 * cr * sp * sy + 2 * Z4 * cr * sp * sy + 2 * X2 * cr * cy * sp -
                   2 * Z5 * cr * cy * sp;
        G_(3, 5) = 2 * Z2 * w4 - 2 * Z3 * w3 + 2 * Z1 * cp * sy;
        G_(4, 5) = 2 * Z2 * w2 - 2 * Z3 * w1 - 2 * Z1 * cp * cy;
#undef G_
        // middle += G covZ G^T (upper triangle)
        double t[36];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) s += G[r * 6 + k] * cz[k * 6 + c];
                t[r * 6 + c] = s;
            }
        int v = 21;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (c >= r) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) s += t[r * 6 + k] * G[c * 6 + k];
                    a[v++] += s;
                }
    }
    block_reduce_store<42>(a, partials);
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
    double e[3];
    euler_012(T_result, e);
    args.cr = cos(e[0]);
    args.sr = sin(e[0]);
    args.cp = cos(e[1]);
    args.sp = sin(e[1]);
    args.cy = cos(e[2]);
    args.sy = sin(e[2]);
    args.X[0] = T_result[3];
    args.X[1] = T_result[7];
    args.X[2] = T_result[11];
    const double sph[6] = {lin_covar, ang_covar, ang_covar, lin_covar, ang_covar, ang_covar};
    for (int k = 0; k < 6; ++k) args.sph[k] = sph[k];
    const unsigned n = (unsigned) ctx->n_src;
    const int nb = info_blocks(n);
    WM_HIP(ctx, ctx->partials.reserve((size_t) kInfoBlocks * kInfoAcc * sizeof(double)));
    hipLaunchKernelGGL(k_censi, dim3(nb), dim3(kBlock), 0, ctx->stream,
                       ctx->src_sorted.as<float4>(), n, ctx->keys.as<unsigned long long>(),
                       ctx->tgt_orig.as<float4>(), args, ctx->partials.as<double>());
    WM_HIP(ctx, hipGetLastError());
    double a[42];
    WM_TRY(reduce_partials(ctx, nb, 42, a, sharded ? comm : nullptr));
    double H[36], Mid[36], Hinv[36], t1[36], t2[36];
    int u = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
            H[r * 6 + c] = H[c * 6 + r] = a[u];
            Mid[r * 6 + c] = Mid[c * 6 + r] = a[21 + u];
            ++u;
        }
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
