// wm_ndt_small.hip -- NDTMatcher::match() for MANY queued pairs in one launch: one registration per workgroup, the
// whole of pcl::NormalDistributionsTransform -- setInputTarget's voxel model and align -- inside it (reference
// behaviour: wave_matching/src/ndt.cpp:18-34 setters, :48-65 setRef / setTarget / match; the pool that queues such
// pairs: wave_matching/include/wave/matching/multi_matcher.hpp:29-34).
//
// Why: on the whole device (wm_ndt.hip) a 20 000-point registration is ~45 derivative passes of a few microseconds
// each, every one followed by a trip across PCIe for the host's Newton step / line-search decision; a pool of such
// registrations is bound by the four hardware queues (~6 500 pairs/s).  Here a pair owns one compute unit: 512
// threads build the target's voxel model (counting sort by voxel, one thread per voxel for its sums), wave 0 runs the
// control (wm_ndt_ctl.hpp: the same source as the host's), and every derivative pass is a workgroup-wide sum.
//
// Same as the one-pair path: which voxels a point sees (the float radius test on float means), the terms of score,
// gradient and Hessian (the per-point factorisation of k_ndt_derivs), the control's code.  Different: a voxel's sums
// are formed in double-double (exact, whatever the order the counting sort left its points in) instead of one after
// the other in point order, the passes' sums are added in another order, and exp / log / sin / cos are the device
// library's: results agree with wm_ndt_align to ~1e-9, not bit for bit (tests: 1e-6 m / 1e-6 rad).
#include "wm_internal.hpp"
#include "wm_ndt_dev.hpp"
#include "wm_ndt_ctl.hpp"
#include "wm_gicp_dev.hpp"  // dd_add

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

namespace wm {

constexpr int kNsThreads = 512;
constexpr int kNsWaves = kNsThreads / 64;
constexpr unsigned kNsCells = 262144;  // cells of the voxel lattice (two empty cells of margin per side included) at most

struct NsPair {  // one registration of the batch (device table)
    const unsigned char *src, *tgt;  // caller-layout points in device memory
    unsigned n_src, n_tgt;
    // scratch in HBM (this pair's own)
    unsigned *start;   // first slot of every cell of the lattice in `order` (+ end)
    unsigned *run;     // the counting sort's counters
    unsigned *order;   // the target's finite points (caller indices) grouped by cell
    int *table;        // cell -> voxel record (-1: none)
    NdtVoxel *vox;     // records: mean + inverse covariance
    float4 *meanf;     // float means (the radius test)
    float4 *spts;      // the source's finite points in a spatial order
};

struct NsParams {
    unsigned stride;
    double res, step_size, t_eps, outlier_ratio;
    int max_iter, forced, skip_line_search, pcl_d1_sign, spec_hessian;
};

struct NsOut {
    double T[16];
    double score;
    int converged, iterations, evaluations, n_voxels, status;
    int n_src_valid, n_tgt_valid;
    unsigned long long cyc[3];  // developer: shader-clock cycles of the model build / the align / the align's passes (the rest of the align: wave 0's control)
};

struct NsShared {
    double red[kNsWaves][kNdtAcc];
    float boxf[kNsWaves][8];
    unsigned wcnt[kNsWaves];
    unsigned scan[kNsWaves];
    unsigned cmd;  // 0: the align is over; 1: score + gradient + Hessian, 2: score + gradient, 3: score + Hessian
    NdtArgs A;     // the pass's arguments (wave 0 writes, everybody reads)
    NdtDense dense;
    NsPair pr;
    NsParams P;
    unsigned n_vox;  // records handed out
    unsigned n_valid;
    unsigned n_src_valid, n_tgt_valid;
    int unsupported;  // the lattice does not fit kNsCells
    NdtLoopOut out;
    int evals;
    unsigned long long pass_cyc;
};

__device__ __forceinline__ bool ns_load(const unsigned char *base, unsigned i, unsigned stride, float &x, float &y, float &z) {
    const float *p = reinterpret_cast<const float *>(base + (size_t) i * stride);
    x = p[0], y = p[1], z = p[2];
    return x - x == 0.f && y - y == 0.f && z - z == 0.f;  // finite
}

// exclusive scan of the counting sort's counts, four cells per thread at a time: run[c] = start[c] = first slot of cell
// c (the counts were made by atomics, which live in L2: read there).  Every thread calls.
__device__ __forceinline__ void ns_scan_counts(unsigned *run, unsigned *start, unsigned nc, NsShared &S) {
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned carry = 0;
    for (unsigned c0 = 0; c0 < nc; c0 += 4u * kNsThreads) {
        const unsigned c = c0 + 4u * tid;
        unsigned v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = c + (unsigned) u < nc ? __hip_atomic_load(&run[c + (unsigned) u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const unsigned mine = v[0] + v[1] + v[2] + v[3];
        unsigned incl = mine;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned up = __shfl_up(incl, m);
            if ((int) lane >= m) incl += up;
        }
        if (lane == 63) S.scan[wave] = incl;
        __syncthreads();
        unsigned base = carry, all = 0;
        for (unsigned w = 0; w < (unsigned) kNsWaves; ++w) {
            if (w < wave) base += S.scan[w];
            all += S.scan[w];
        }
        unsigned at = base + incl - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c + (unsigned) u < nc) {
                run[c + (unsigned) u] = at;
                start[c + (unsigned) u] = at;
                at += v[u];
            }
        carry += all;
        __syncthreads();
    }
}

// pcl::VoxelGridCovariance::filter of the target (setInputTarget, ndt.cpp:55): S.dense / S.n_valid
__device__ __attribute__((noinline)) void ns_build_model(NsShared &S) {
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned char *raw = S.pr.tgt;
    const unsigned n = S.pr.n_tgt, stride = S.P.stride;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    unsigned cnt = 0;
    for (unsigned i = tid; i < n; i += kNsThreads) {
        float x, y, z;
        if (ns_load(raw, i, stride, x, y, z)) {
            lo[0] = fminf(lo[0], x), lo[1] = fminf(lo[1], y), lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x), hi[1] = fmaxf(hi[1], y), hi[2] = fmaxf(hi[2], z);
            ++cnt;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], m));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m));
        }
        cnt += __shfl_xor(cnt, m);
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) S.boxf[wave][d] = lo[d], S.boxf[wave][3 + d] = hi[d];
        S.wcnt[wave] = cnt;
    }
    __syncthreads();
    cnt = 0;
    for (int w = 0; w < kNsWaves; ++w) {
#pragma unroll
        for (int d = 0; d < 3; ++d) lo[d] = fminf(lo[d], S.boxf[w][d]), hi[d] = fmaxf(hi[d], S.boxf[w][3 + d]);
        cnt += S.wcnt[w];
    }
    __syncthreads();
    // the lattice: cell numbers from the same float product as the one-pair path's keys, two empty cells of margin
    // on every side (a query's 3 x 3 x 3 block is then read without bounds tests: k_ndt_derivs, wm_ndt.hip)
    const float inv = 1.0f / (float) S.P.res;
    NdtDense d{S.pr.table, 0, 0, 0, 1, 1, 1};
    long long cells = 1;
    if (cnt > 0) {
        int l3[3] = {0, 0, 0}, d3[3] = {1, 1, 1};
        bool fits = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // (in double: a finite outlier with |coordinate / res| >= 2^31 -- a garbage lidar return, res down to 0.05 --
            // saturates an int cast, and the - 2 / + 3 then overflow; wm_ndt.hip does this sum in long long)
            const double fl = floor((double) (lo[a] * inv)), fh = floor((double) (hi[a] * inv));
            const double dd = fh - fl + 5.0;
            if (!(fabs(fl) < 1073741824.0) || !(fabs(fh) < 1073741824.0) || !(dd >= 1.0) || dd > 1048576.0) {
                fits = false;
            } else {
                l3[a] = (int) fl - 2;
                d3[a] = (int) dd;
                cells *= d3[a];
            }
        }
        if (!fits) cells = (long long) kNsCells + 1;  // -> unsupported: registered by wm_ndt_align inside the same call
        else d = NdtDense{S.pr.table, l3[0], l3[1], l3[2], d3[0], d3[1], d3[2]};
    }
    if (tid == 0) {
        S.dense = d;
        S.n_tgt_valid = cnt;
        S.n_vox = 0;
        S.n_valid = 0;
        S.unsupported = cells > (long long) kNsCells ? 1 : 0;
    }
    __syncthreads();
    if (cells > (long long) kNsCells) return;
    const unsigned nc = (unsigned) cells;
    unsigned *run = S.pr.run, *start = S.pr.start;
    int *table = S.pr.table;
    for (unsigned c = tid; c < nc; c += kNsThreads) {
        run[c] = 0u;
        table[c] = -1;
    }
    __syncthreads();
    auto cell_of = [&](float x, float y, float z) {
        const int a = (int) floorf(__fmul_rn(x, inv)) - d.i0, b = (int) floorf(__fmul_rn(y, inv)) - d.j0, c = (int) floorf(__fmul_rn(z, inv)) - d.k0;
        return (unsigned) ((c * d.ny + b) * d.nx + a);
    };
    for (unsigned i = tid; i < n; i += kNsThreads) {
        float x, y, z;
        if (ns_load(raw, i, stride, x, y, z)) atomicAdd(&run[cell_of(x, y, z)], 1u);
    }
    __syncthreads();
    ns_scan_counts(run, start, nc, S);
    if (tid == 0) start[nc] = cnt;
    __syncthreads();
    unsigned *order = S.pr.order;
    for (unsigned i = tid; i < n; i += kNsThreads) {
        float x, y, z;
        if (ns_load(raw, i, stride, x, y, z)) order[atomicAdd(&run[cell_of(x, y, z)], 1u)] = i;
    }
    __syncthreads();
    // one thread per occupied cell: n, sum p, sum p p^T.  The counting sort left a cell's points in no particular
    // order, so the sums are formed in double-double: the exact sums rounded once, whatever the order.
    NdtVoxel *vox = S.pr.vox;
    float4 *meanf = S.pr.meanf;
    for (unsigned c = tid; c < nc; c += kNsThreads) {
        const unsigned s0 = start[c], s1 = start[c + 1];
        if (s1 - s0 < 6u) continue;  // min_points_per_voxel_
        double sh[3] = {0, 0, 0}, sl[3] = {0, 0, 0}, ph[6] = {0, 0, 0, 0, 0, 0}, pl[6] = {0, 0, 0, 0, 0, 0};
        for (unsigned t = s0; t < s1; ++t) {
            float x, y, z;
            (void) ns_load(raw, order[t], stride, x, y, z);
            const double q[3] = {(double) x, (double) y, (double) z};
#pragma unroll
            for (int a = 0; a < 3; ++a) dd_add(sh[a], sl[a], q[a]);
            dd_add(ph[0], pl[0], q[0] * q[0]);  // (a product of two floats is exact in double)
            dd_add(ph[1], pl[1], q[0] * q[1]);
            dd_add(ph[2], pl[2], q[0] * q[2]);
            dd_add(ph[3], pl[3], q[1] * q[1]);
            dd_add(ph[4], pl[4], q[1] * q[2]);
            dd_add(ph[5], pl[5], q[2] * q[2]);
        }
        const double s[3] = {sh[0] + sl[0], sh[1] + sl[1], sh[2] + sl[2]};
        const double u[6] = {ph[0] + pl[0], ph[1] + pl[1], ph[2] + pl[2], ph[3] + pl[3], ph[4] + pl[4], ph[5] + pl[5]};
        const double pp[9] = {u[0], u[1], u[2], u[1], u[3], u[4], u[2], u[4], u[5]};
        NdtVoxel v;
        if (!ndt_voxel_record(s1 - s0, s, pp, v)) continue;
        const unsigned slot = atomicAdd(&S.n_vox, 1u);
        vox[slot] = v;
        meanf[slot] = make_float4((float) v.mean[0], (float) v.mean[1], (float) v.mean[2], 0.0f);
        table[c] = (int) slot;
    }
    __syncthreads();
    if (tid == 0) S.n_valid = S.n_vox;
    __syncthreads();
}

// The source in a spatial order (S.pr.spts, S.n_src_valid points): counting sort by the cell of a grid of its own
// (~8 points per cell by volume), every cell's points in ascending caller index -- a fixed order, so that the passes'
// sums are the same from run to run.  In caller order (arbitrary: a shuffled cloud) every lane of a wave looked up
// another corner of the voxel table and another voxel record; 512 neighbours in space share them.
__device__ __attribute__((noinline)) void ns_sort_source(NsShared &S) {
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned char *raw = S.pr.src;
    const unsigned n = S.pr.n_src, stride = S.P.stride;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    unsigned cnt = 0;
    for (unsigned i = tid; i < n; i += kNsThreads) {
        float x, y, z;
        if (ns_load(raw, i, stride, x, y, z)) {
            lo[0] = fminf(lo[0], x), lo[1] = fminf(lo[1], y), lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x), hi[1] = fmaxf(hi[1], y), hi[2] = fmaxf(hi[2], z);
            ++cnt;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], m));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m));
        }
        cnt += __shfl_xor(cnt, m);
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) S.boxf[wave][d] = lo[d], S.boxf[wave][3 + d] = hi[d];
        S.wcnt[wave] = cnt;
    }
    __syncthreads();
    cnt = 0;
    for (int w = 0; w < kNsWaves; ++w) {
#pragma unroll
        for (int d = 0; d < 3; ++d) lo[d] = fminf(lo[d], S.boxf[w][d]), hi[d] = fmaxf(hi[d], S.boxf[w][3 + d]);
        cnt += S.wcnt[w];
    }
    __syncthreads();
    if (tid == 0) S.n_src_valid = cnt;
    if (cnt == 0) {
        __syncthreads();
        return;
    }
    double vol = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) vol *= fmax((double) hi[d] - (double) lo[d], 1e-3);
    float h = (float) fmax(cbrt(vol / (double) cnt) * 2.0, 1e-4);
    int nx, ny, nz;
    for (;;) {
        nx = (int) floorf((hi[0] - lo[0]) / h) + 1, ny = (int) floorf((hi[1] - lo[1]) / h) + 1, nz = (int) floorf((hi[2] - lo[2]) / h) + 1;
        if ((unsigned long long) nx * (unsigned long long) ny * (unsigned long long) nz <= (unsigned long long) kNsCells) break;
        h *= 1.1f;
    }
    const float inv_h = 1.0f / h;
    const unsigned nc = (unsigned) (nx * ny * nz);
    auto cell_of = [&](float x, float y, float z) {
        const int a = min(max((int) floorf((x - lo[0]) * inv_h), 0), nx - 1), b = min(max((int) floorf((y - lo[1]) * inv_h), 0), ny - 1),
                  c = min(max((int) floorf((z - lo[2]) * inv_h), 0), nz - 1);
        return (unsigned) ((c * ny + b) * nx + a);
    };
    unsigned *run = S.pr.run, *start = S.pr.start, *order = S.pr.order;
    for (unsigned c = tid; c < nc; c += kNsThreads) run[c] = 0u;
    __syncthreads();
    for (unsigned i = tid; i < n; i += kNsThreads) {
        float x, y, z;
        if (ns_load(raw, i, stride, x, y, z)) atomicAdd(&run[cell_of(x, y, z)], 1u);
    }
    __syncthreads();
    ns_scan_counts(run, start, nc, S);
    if (tid == 0) start[nc] = cnt;
    __syncthreads();
    for (unsigned i = tid; i < n; i += kNsThreads) {
        float x, y, z;
        if (ns_load(raw, i, stride, x, y, z)) order[atomicAdd(&run[cell_of(x, y, z)], 1u)] = i;
    }
    __syncthreads();
    // a thread per cell puts its points in ascending index (insertion sort: a cell holds a few dozen), then writes them out
    float4 *spts = S.pr.spts;
    for (unsigned c = tid; c < nc; c += kNsThreads) {
        const unsigned s0 = start[c], s1 = start[c + 1];
        for (unsigned a = s0 + 1; a < s1; ++a) {
            const unsigned v = order[a];
            unsigned b = a;
            while (b > s0 && order[b - 1] > v) {
                order[b] = order[b - 1];
                --b;
            }
            order[b] = v;
        }
        for (unsigned a = s0; a < s1; ++a) {
            float x, y, z;
            (void) ns_load(raw, order[a], stride, x, y, z);
            spts[a] = make_float4(x, y, z, 1.f);
        }
    }
    __syncthreads();
}

// this thread's share of one derivative pass (the body of k_ndt_derivs, wm_ndt.hip: see there for why it is laid out
// the way it is), then the workgroup's rows -> S.red
template <bool GRAD, bool HESS>
__device__ __attribute__((noinline)) void ns_share(NsShared &S, unsigned *s_near, double (*s_jh)[3]) {
#pragma clang fp contract(fast)
    const unsigned tid = threadIdx.x;
    const NdtArgs &A = S.A;
    const NdtDense dense = S.dense;
    const NdtVoxel *__restrict__ vox = S.pr.vox;
    const float4 *__restrict__ meanf = S.pr.meanf, *__restrict__ src = S.pr.spts;
    const unsigned n = S.n_src_valid;
    auto jh_dot = [&](const double (&x)[3], int k, int opaque0) -> double {
        const double *v = &s_jh[k + opaque0][0];
        return x[0] * v[0] + x[1] * v[1] + x[2] * v[2];
    };
    if (GRAD || HESS) {
        if (tid < 69u) (&s_jh[0][0])[tid] = tid < 24u ? (&A.j[0][0])[tid] : (&A.h[0][0])[tid - 24u];
        __syncthreads();
    }
    const float T0 = A.Tf[0], T1 = A.Tf[1], T2 = A.Tf[2], T3 = A.Tf[3], T4 = A.Tf[4], T5 = A.Tf[5], T6 = A.Tf[6], T7 = A.Tf[7], T8 = A.Tf[8],
                T9 = A.Tf[9], T10 = A.Tf[10], T11 = A.Tf[11];
    const float inv_res = A.inv_res, res2_f = A.res2_f;
    const double d1 = A.d1, d2 = A.d2;
    constexpr int NA = HESS ? kNdtAcc : kNdtAccGrad;
    double acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0;
    for (unsigned idx = tid; idx < n; idx += kNsThreads) {
        const float4 sp = src[idx];
        const float xt0 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T0, sp.x), __fmul_rn(T1, sp.y)), __fmul_rn(T2, sp.z)), T3);
        const float xt1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T4, sp.x), __fmul_rn(T5, sp.y)), __fmul_rn(T6, sp.z)), T7);
        const float xt2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T8, sp.x), __fmul_rn(T9, sp.y)), __fmul_rn(T10, sp.z)), T11);
        if (!(xt0 - xt0 == 0.f && xt1 - xt1 == 0.f && xt2 - xt2 == 0.f)) continue;
        const int ci = (int) floorf(__fmul_rn(xt0, inv_res));
        const int cj = (int) floorf(__fmul_rn(xt1, inv_res));
        const int ck = (int) floorf(__fmul_rn(xt2, inv_res));
        int n_cand = 0;
        const int ta = ci - dense.i0, tb = cj - dense.j0, tc = ck - dense.k0;
        if (ta >= 1 && tb >= 1 && tc >= 1 && ta <= dense.nx - 2 && tb <= dense.ny - 2 && tc <= dense.nz - 2) {
            Int3 rows[9];
#pragma unroll
            for (int row = 0; row < 9; ++row) {  // (dk, dj); the three di cells are adjacent in x
                const int dj = row % 3 - 1, dk = row / 3 - 1;
                rows[row] = *(const Int3 *) (dense.table + (((size_t) (tc + dk) * dense.ny + (tb + dj)) * dense.nx + (ta - 1)));
            }
#pragma unroll
            for (int row = 0; row < 9; ++row) {
                const int v3[3] = {rows[row].a, rows[row].b, rows[row].c};
#pragma unroll
                for (int dd = 0; dd < 3; ++dd)
                    if (v3[dd] != -1) {
                        s_near[n_cand * kNsThreads + tid] = (unsigned) v3[dd];
                        ++n_cand;
                    }
            }
        }
        // kd-tree radius test in float on the float means; survivors compacted in place, in cell order
        // (four candidates per trip so that four loads are in flight)
        int n_near = 0;
#pragma unroll 1
        for (int r = 0; r < n_cand; r += 4) {
            unsigned cv[4];
            float4 cm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = s_near[min(r + u, n_cand - 1) * kNsThreads + tid];
#pragma unroll
            for (int u = 0; u < 4; ++u) cm[u] = meanf[cv[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float fx = __fsub_rn(xt0, cm[u].x), fy = __fsub_rn(xt1, cm[u].y), fz = __fsub_rn(xt2, cm[u].z);
                const float dd = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));
                if (r + u < n_cand && dd <= res2_f) {  // <=> (double) dd < res^2
                    s_near[n_near * kNsThreads + tid] = cv[u];
                    ++n_near;
                }
            }
        }
        // the next voxel's record (96 B) is requested before the current one is evaluated
        NdtVoxel vn;
        {
            const unsigned v0 = n_near > 0 ? s_near[tid] : 0u;
#pragma unroll
            for (int k = 0; k < 3; ++k) vn.mean[k] = vox[v0].mean[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) vn.icov[k] = vox[v0].icov[k];
        }
        double g3[3] = {0.0, 0.0, 0.0}, P[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double Q[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        bool any = false;
#pragma unroll 1
        for (int t = 0; t < n_near; ++t) {
            const NdtVoxel v = vn;
            {
                const unsigned v1 = t + 1 < n_near ? s_near[(t + 1) * kNsThreads + tid] : 0u;
#pragma unroll
                for (int k = 0; k < 3; ++k) vn.mean[k] = vox[v1].mean[k];
#pragma unroll
                for (int k = 0; k < 9; ++k) vn.icov[k] = vox[v1].icov[k];
            }
            const double xx[3] = {(double) xt0 - v.mean[0], (double) xt1 - v.mean[1], (double) xt2 - v.mean[2]};
            double cx[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) cx[a] = v.icov[a * 3] * xx[0] + v.icov[a * 3 + 1] * xx[1] + v.icov[a * 3 + 2] * xx[2];
            const double q = xx[0] * cx[0] + xx[1] * cx[1] + xx[2] * cx[2];
            const double e = exp(-d2 * q / 2.0);
            double w = d2 * e;
            if (w > 1 || w < 0 || w != w) continue;
            acc[0] += -d1 * e;
            w *= d1;
            if (GRAD || HESS) {
                any = true;
                const double wc[3] = {w * cx[0], w * cx[1], w * cx[2]};
#pragma unroll
                for (int a = 0; a < 3; ++a) g3[a] += wc[a];
                if (HESS) {
                    P[0] += wc[0] * cx[0];
                    P[1] += wc[0] * cx[1];
                    P[2] += wc[0] * cx[2];
                    P[3] += wc[1] * cx[1];
                    P[4] += wc[1] * cx[2];
                    P[5] += wc[2] * cx[2];
#pragma unroll
                    for (int k = 0; k < 9; ++k) Q[k] += w * v.icov[k];
                }
            }
        }
        if ((GRAD || HESS) && any) {
            const double x[3] = {(double) sp.x, (double) sp.y, (double) sp.z};
            int z0 = 0;
            asm volatile("" : "+s"(z0));
            double Jc[3][3];
            Jc[0][0] = 0.0;
            Jc[0][1] = jh_dot(x, 0, z0);
            Jc[0][2] = jh_dot(x, 1, z0);
            Jc[1][0] = jh_dot(x, 2, z0);
            Jc[1][1] = jh_dot(x, 3, z0);
            Jc[1][2] = jh_dot(x, 4, z0);
            Jc[2][0] = jh_dot(x, 5, z0);
            Jc[2][1] = jh_dot(x, 6, z0);
            Jc[2][2] = jh_dot(x, 7, z0);
            if (GRAD) {
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[1 + i] += g3[i];
                acc[4] += Jc[0][1] * g3[1] + Jc[0][2] * g3[2];
                acc[5] += Jc[1][0] * g3[0] + Jc[1][1] * g3[1] + Jc[1][2] * g3[2];
                acc[6] += Jc[2][0] * g3[0] + Jc[2][1] * g3[1] + Jc[2][2] * g3[2];
            }
            if (HESS) {
                const double md2 = -d2;
                double Sm[3][3];
                Sm[0][0] = md2 * P[0] + Q[0];
                Sm[0][1] = md2 * P[1] + Q[3];
                Sm[0][2] = md2 * P[2] + Q[6];
                Sm[1][0] = md2 * P[1] + Q[1];
                Sm[1][1] = md2 * P[3] + Q[4];
                Sm[1][2] = md2 * P[4] + Q[7];
                Sm[2][0] = md2 * P[2] + Q[2];
                Sm[2][1] = md2 * P[4] + Q[5];
                Sm[2][2] = md2 * P[5] + Q[8];
                double SJ[3][3];  // SJ[a][c] = sum_b S[a][b] Jc[c][b]
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    SJ[a][0] = Sm[a][1] * Jc[0][1] + Sm[a][2] * Jc[0][2];
                    SJ[a][1] = Sm[a][0] * Jc[1][0] + Sm[a][1] * Jc[1][1] + Sm[a][2] * Jc[1][2];
                    SJ[a][2] = Sm[a][0] * Jc[2][0] + Sm[a][1] * Jc[2][1] + Sm[a][2] * Jc[2][2];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = i; j < 3; ++j) acc[ndt_tri(i, j)] += Sm[i][j];
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[ndt_tri(i, 3 + c)] += SJ[i][c];
                }
#pragma unroll
                for (int ci2 = 0; ci2 < 3; ++ci2)
#pragma unroll
                    for (int cj2 = ci2; cj2 < 3; ++cj2) {
                        const int sel = (ci2 == 0 && cj2 == 0) ? 0 : ((ci2 + cj2 == 1) ? 1 : ((ci2 + cj2 == 2 && ci2 != cj2) ? 2 : ((ci2 == 1 && cj2 == 1) ? 3 : ((ci2 + cj2 == 3) ? 4 : 5))));
                        double t = Jc[ci2][1] * SJ[1][cj2] + Jc[ci2][2] * SJ[2][cj2];
                        if (ci2 > 0) t += Jc[ci2][0] * SJ[0][cj2];
                        if (sel < 3) {
                            t += g3[1] * jh_dot(x, 8 + 2 * sel, z0) + g3[2] * jh_dot(x, 8 + 2 * sel + 1, z0);
                        } else {
                            const int k0 = 8 + 6 + 3 * (sel - 3);
                            t += g3[0] * jh_dot(x, k0, z0) + g3[1] * jh_dot(x, k0 + 1, z0) + g3[2] * jh_dot(x, k0 + 2, z0);
                        }
                        acc[ndt_tri(3 + ci2, 3 + cj2)] += t;
                    }
            }
        }
    }
    // fixed-order reduction: wave tree, one row per wave
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    if ((tid & 63u) == 0)
#pragma unroll
        for (int k = 0; k < NA; ++k) S.red[tid >> 6][k] = acc[k];
}

__device__ __forceinline__ void ns_share_cmd(NsShared &S, unsigned cmd, unsigned *s_near, double (*s_jh)[3]) {
    if (cmd == 1u) ns_share<true, true>(S, s_near, s_jh);
    else if (cmd == 2u) ns_share<true, false>(S, s_near, s_jh);
    else ns_share<false, true>(S, s_near, s_jh);
}

// the objective as wm_ndt_ctl.hpp asks for it: wave 0 holds it and runs the control; the other waves serve()
struct NsEval {
    NsShared *S;
    unsigned *s_near;
    double (*s_jh)[3];
    double d1, d2;
    int evals;
    unsigned long long pass_cyc;
    __device__ bool failed() const { return false; }
    __device__ bool skip_line_search() const { return S->P.skip_line_search != 0; }
    __device__ bool spec_hessian() const { return S->P.spec_hessian != 0; }
    __device__ void note_line_search(int) {}
    __device__ double eval(const double p[6], double *grad, double *hess) {
        const unsigned lane = threadIdx.x & 63u;
        const unsigned cmd = (grad && hess) ? 1u : (grad ? 2u : 3u);
        if (lane == 0) {
            NdtArgs &A = S->A;
            float Tf[16];
            pose_to_matrix_f(p, Tf);
            for (int k = 0; k < 12; ++k) A.Tf[k] = Tf[k];
            A.inv_res = 1.0f / (float) S->P.res;
            A.res2 = S->P.res * S->P.res;
            A.d1 = d1;
            A.d2 = d2;
            angle_derivatives(p, S->P.pcl_d1_sign, &A);
            S->cmd = cmd;
        }
        const unsigned long long t0 = clock64();
        __syncthreads();
        ns_share_cmd(*S, cmd, s_near, s_jh);
        __syncthreads();
        pass_cyc += clock64() - t0;
        const int n_acc = hess ? kNdtAcc : kNdtAccGrad;
        double v = 0;
        if ((int) lane < n_acc)
            for (int w = 0; w < kNsWaves; ++w) v += S->red[w][lane];
        double a[kNdtAcc];
#pragma unroll
        for (int k = 0; k < kNdtAcc; ++k) a[k] = __shfl(v, k);
        ++evals;
        if (grad)
            for (int k = 0; k < 6; ++k) grad[k] = a[1 + k];
        if (hess)
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) hess[i * 6 + j] = hess[j * 6 + i] = a[ndt_tri(i, j)];
        return a[0];
    }
    __device__ void serve() {  // waves 1 .. 7, for the length of the align
        for (;;) {
            __syncthreads();
            const unsigned cmd = S->cmd;
            if (cmd == 0u) return;
            ns_share_cmd(*S, cmd, s_near, s_jh);
            __syncthreads();
        }
    }
};

__global__ void __launch_bounds__(kNsThreads) k_ndt_small(const NsPair *__restrict__ table, NsParams P, float res2_f, NsOut *__restrict__ outs) {
    __shared__ unsigned s_near[27 * kNsThreads];  // per-lane lists of a pass (54 KB)
    __shared__ double s_jh[23][3];
    __shared__ NsShared S;
    NsOut &out = outs[blockIdx.x];
    const unsigned tid = threadIdx.x;
    if (tid == 0) {
        S.pr = table[blockIdx.x];
        S.P = P;
        S.A.res2_f = res2_f;
        S.cmd = 0u;
        S.evals = 0;
    }
    __syncthreads();
    unsigned long long t_mark = clock64();
    ns_build_model(S);
    ns_sort_source(S);
    if (tid == 0) {
        out.cyc[0] = clock64() - t_mark;
        out.n_src_valid = (int) S.n_src_valid;
        out.n_tgt_valid = (int) S.n_tgt_valid;
        out.n_voxels = (int) S.n_valid;
        out.converged = 0, out.iterations = 0, out.evaluations = 0;
        out.score = 0;
        out.status = WM_NOT_CONVERGED;
    }
    if (S.unsupported) {  // (uniform) the lattice is too big for this kernel: the host registers the pair by itself
        if (tid == 0) out.status = WM_ERR_ARG;
        return;
    }
    t_mark = clock64();
    NsEval E;
    E.S = &S;
    E.s_near = s_near;
    E.s_jh = s_jh;
    E.evals = 0;
    E.pass_cyc = 0;
    {
        const double c1 = 10.0 * (1.0 - P.outlier_ratio), c2 = P.outlier_ratio / pow(P.res, 3);
        const double d3 = -log(c2);
        E.d1 = -log(c1 + c2) - d3;
        E.d2 = -2.0 * log((-log(c1 * exp(-0.5) + c2) - d3) / E.d1);
    }
    if (tid < 64u) {
        NdtLoopOut lo;
        ndt_align_loop(E, P.step_size, P.t_eps, P.max_iter, P.forced, &lo);
        if (tid == 0) {
            S.out = lo;
            S.evals = E.evals;
            S.pass_cyc = E.pass_cyc;
            S.cmd = 0u;
        }
        __syncthreads();  // (the waves in serve() see the end)
    } else {
        E.serve();
    }
    __syncthreads();
    if (tid == 0) {
        float Tf[16];
        pose_to_matrix_f(S.out.p, Tf);
        for (int k = 0; k < 16; ++k) out.T[k] = (double) Tf[k];
        out.score = S.n_src_valid > 0 ? S.out.score / (double) S.pr.n_src : 0.0;
        out.converged = S.out.converged ? 1 : 0;
        out.iterations = S.out.iterations;
        out.evaluations = S.evals;
        out.status = S.out.converged ? WM_OK : WM_NOT_CONVERGED;
        out.cyc[1] = clock64() - t_mark;
        out.cyc[2] = S.pass_cyc;
    }
}

// ---- host
struct NdtSmallBatch {
    DevBuf d_stage;  // [table | clouds]
    DevBuf d_work;
    DevBuf d_out;
    void *h_stage = nullptr;
    size_t h_stage_cap = 0;
    void *h_out = nullptr;
    size_t h_out_cap = 0;
};

static NdtSmallBatch *ns_of(wm_ctx *ctx) {
    if (!ctx->ndt_small_batch) ctx->ndt_small_batch = new (std::nothrow) NdtSmallBatch();
    return static_cast<NdtSmallBatch *>(ctx->ndt_small_batch);
}

void ndt_small_release(wm_ctx *ctx) {
    NdtSmallBatch *b = static_cast<NdtSmallBatch *>(ctx->ndt_small_batch);
    if (!b) return;
    b->d_stage.release();
    b->d_work.release();
    b->d_out.release();
    if (b->h_stage) (void) hipHostFree(b->h_stage);
    if (b->h_out) (void) hipHostFree(b->h_out);
    delete b;
    ctx->ndt_small_batch = nullptr;
}



struct NsJob {
    const void *src;
    size_t n_src;
    const void *tgt;
    size_t n_tgt;
};

static int ndt_small_run(wm_ctx *ctx, const NsJob *jobs, int n, size_t stride, int mem, const wm_ndt_params *prm, NsOut *res,
                         float *kernel_ms) {
    if (n <= 0) return WM_OK;
    NdtSmallBatch *B = ns_of(ctx);
    if (!B) return WM_ERR_NOMEM;
    size_t cloud_bytes = 0, work_bytes = 0;
    auto work_need = [](size_t ns, size_t nt) {
        return 2 * align_up256(((size_t) kNsCells + 8) * 4) + align_up256((nt > ns ? nt : ns) * 4 + 16) + align_up256((size_t) kNsCells * 4) +
               align_up256((nt / 6 + 1) * sizeof(NdtVoxel)) + align_up256((nt / 6 + 1) * sizeof(float4)) + align_up256(ns * 16 + 16);
    };
    for (int k = 0; k < n; ++k) {
        if (jobs[k].n_src == 0 || jobs[k].n_tgt == 0 || jobs[k].n_src > (size_t) WM_NDT_BATCH_MAX_POINTS ||
            jobs[k].n_tgt > (size_t) WM_NDT_BATCH_MAX_POINTS)
            return WM_ERR_ARG;
        cloud_bytes += align_up256(jobs[k].n_src * stride) + align_up256(jobs[k].n_tgt * stride);
        work_bytes += work_need(jobs[k].n_src, jobs[k].n_tgt);
    }
    const size_t table_bytes = align_up256((size_t) n * sizeof(NsPair));
    const size_t up_bytes = table_bytes + (mem == WM_MEM_HOST ? cloud_bytes : 0);
    WM_HIP(ctx, B->d_stage.reserve(up_bytes));
    WM_HIP(ctx, B->d_work.reserve(work_bytes));
    WM_HIP(ctx, B->d_out.reserve((size_t) n * sizeof(NsOut)));
    WM_TRY(pinned_reserve(ctx, &B->h_stage, &B->h_stage_cap, up_bytes));
    WM_TRY(pinned_reserve(ctx, &B->h_out, &B->h_out_cap, (size_t) n * sizeof(NsOut)));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the stream may still be reading the staging buffer for the previous batch)
    unsigned char *h = static_cast<unsigned char *>(B->h_stage), *d = B->d_stage.as<unsigned char>(), *w = B->d_work.as<unsigned char>();
    NsPair *table = reinterpret_cast<NsPair *>(h);
    size_t off = table_bytes, sent = table_bytes;
    for (int k = 0; k < n; ++k) {
        const NsJob &it = jobs[k];
        NsPair &t = table[k];
        t.n_src = (unsigned) it.n_src;
        t.n_tgt = (unsigned) it.n_tgt;
        if (mem == WM_MEM_HOST) {
            memcpy(h + off, it.src, it.n_src * stride);
            t.src = d + off;
            off += align_up256(it.n_src * stride);
            memcpy(h + off, it.tgt, it.n_tgt * stride);
            t.tgt = d + off;
            off += align_up256(it.n_tgt * stride);
            if (off - sent >= ((size_t) 2 << 20)) {
                WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
                sent = off;
            }
        } else {
            t.src = static_cast<const unsigned char *>(it.src);
            t.tgt = static_cast<const unsigned char *>(it.tgt);
        }
        auto take = [&](size_t bytes) {
            unsigned char *p = w;
            w += align_up256(bytes);
            return p;
        };
        t.start = reinterpret_cast<unsigned *>(take(((size_t) kNsCells + 8) * 4));
        t.run = reinterpret_cast<unsigned *>(take(((size_t) kNsCells + 8) * 4));
        t.order = reinterpret_cast<unsigned *>(take((it.n_tgt > it.n_src ? it.n_tgt : it.n_src) * 4 + 16));
        t.table = reinterpret_cast<int *>(take((size_t) kNsCells * 4));
        t.vox = reinterpret_cast<NdtVoxel *>(take((it.n_tgt / 6 + 1) * sizeof(NdtVoxel)));
        t.meanf = reinterpret_cast<float4 *>(take((it.n_tgt / 6 + 1) * sizeof(float4)));
        t.spts = reinterpret_cast<float4 *>(take(it.n_src * 16 + 16));
    }
    if (off > sent) WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(ctx, hipMemcpyAsync(d, h, table_bytes, hipMemcpyHostToDevice, ctx->stream));
    NsParams P;
    memset(&P, 0, sizeof(P));
    P.stride = (unsigned) stride;
    P.res = prm->res;
    P.step_size = prm->step_size;
    P.t_eps = prm->t_eps;
    P.outlier_ratio = prm->outlier_ratio;
    P.max_iter = prm->max_iter;
    P.forced = prm->force_iterations;
    P.skip_line_search = prm->skip_line_search;
    P.pcl_d1_sign = prm->pcl_d1_sign;
    P.spec_hessian = ctx->tune_ndt_spec_hessian;
    WM_HIP(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    hipLaunchKernelGGL(k_ndt_small, dim3((unsigned) n), dim3(kNsThreads), 0, ctx->stream, reinterpret_cast<const NsPair *>(d), P,
                       threshold_d2_strict(prm->res), B->d_out.as<NsOut>());
    WM_HIP(ctx, hipGetLastError());
    WM_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    WM_HIP(ctx, hipMemcpyAsync(B->h_out, B->d_out.p, (size_t) n * sizeof(NsOut), hipMemcpyDeviceToHost, ctx->stream));
    WM_TRY(sync_sleeping(ctx));
    if (kernel_ms) (void) hipEventElapsedTime(kernel_ms, ctx->ev_a, ctx->ev_b);
    memcpy(res, B->h_out, (size_t) n * sizeof(NsOut));
    return WM_OK;
}

}  // namespace wm

using namespace wm;

extern "C" {

int wm_ndt_batch_match(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride, int mem, const wm_ndt_params *p,
                       double *T_out, wm_ndt_stats *stats, int *status, float *kernel_ms) {
    if (!ctx || !p || !status || n_items < 0 || (n_items > 0 && !items) || stride < 12 || (stride & 3)) return WM_ERR_ARG;
    if (!(p->res > 0) || !(p->step_size > 0)) return WM_ERR_ARG;
    if (kernel_ms) *kernel_ms = 0;
    if (n_items == 0) return WM_OK;
    for (int k = 0; k < n_items; ++k) {
        const wm_batch_item &it = items[k];
        if ((it.n_src > 0 && !it.src) || (it.n_target > 0 && !it.target) || it.n_src > (size_t) WM_NDT_BATCH_MAX_POINTS ||
            it.n_target > (size_t) WM_NDT_BATCH_MAX_POINTS)
            return WM_ERR_ARG;
    }
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats) * (size_t) n_items);
    std::vector<NsJob> jobs;
    std::vector<int> item_of, one_by_one;
    for (int k = 0; k < n_items; ++k) {
        const wm_batch_item &it = items[k];
        status[k] = (it.n_src == 0 || it.n_target == 0) ? WM_ERR_STATE : WM_OK;  // (wm_ndt_align on an empty cloud)
        if (status[k] != WM_OK) continue;
        jobs.push_back(NsJob{it.src, it.n_src, it.target, it.n_target});
        item_of.push_back(k);
    }
    if (!jobs.empty()) {
        std::vector<NsOut> got(jobs.size());
        WM_TRY(ndt_small_run(ctx, jobs.data(), (int) jobs.size(), stride, mem, p, got.data(), kernel_ms));
        for (size_t j = 0; j < jobs.size(); ++j) {
            const int k = item_of[j];
            const NsOut &r = got[j];
            if (r.status == WM_ERR_ARG) {  // (a voxel lattice beyond the kernel's table: registered by itself below)
                one_by_one.push_back(k);
                continue;
            }
            status[k] = r.status;
            if (stats) {
                stats[k].converged = r.converged;
                stats[k].iterations = r.iterations;
                stats[k].n_voxels = r.n_voxels;
                stats[k].evaluations = r.evaluations;
                stats[k].score = r.score;
                stats[k].model_builds = 1;
            }
            if (r.status == WM_OK && T_out) memcpy(T_out + 16 * (size_t) k, r.T, sizeof(r.T));
            if (ctx->trace)
                fprintf(stderr, "[wm] ndt batch: pair %d: %d + %d points, %d voxels, status %d, %d iterations, %d passes; kcycles: model %llu, align %llu (of it the passes %llu)\n",
                        k, r.n_src_valid, r.n_tgt_valid, r.n_voxels, r.status, r.iterations, r.evaluations, r.cyc[0] / 1000, r.cyc[1] / 1000, r.cyc[2] / 1000);
        }
    }
    for (int k : one_by_one) {
        double T[16];
        wm_ndt_stats s;
        int rc = wm_set_source(ctx, items[k].src, items[k].n_src, stride, mem);
        if (rc == WM_OK) rc = wm_set_target(ctx, items[k].target, items[k].n_target, stride, mem);
        if (rc == WM_OK) rc = wm_ndt_align(ctx, p, T, &s);
        // (WM_ERR_ARG here is about THIS pair's data -- a lattice beyond 2^20 voxels along an axis: one garbage
        // return must not fail the whole batch; device errors do)
        if (rc < 0 && rc != WM_ERR_STATE && rc != WM_ERR_ARG) return rc;
        status[k] = rc;
        if (stats) stats[k] = s;
        if (rc == WM_OK && T_out) memcpy(T_out + 16 * (size_t) k, T, sizeof(T));
    }
    return WM_OK;
}

}  // extern "C"
