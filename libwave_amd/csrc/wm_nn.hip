// wm_nn.hip -- exact 1-nearest-neighbour correspondence search on gfx950.
//
// Replaces pcl::registration::CorrespondenceEstimation::determineCorrespondences
// (one FLANN kd-tree 1-NN query per source point per ICP iteration), which is
// where ~95 % of the reference's ICPMatcher::match() time goes
// (wave_matching/src/icp.cpp:95,116,126 -> icp.align()).
//
// Contract (what a FLANN exact search + PCL's distance gate give): for every
// source point the target point with the smallest float squared distance
//     d2 = (dx*dx + dy*dy) + dz*dz            (no FMA contraction)
// provided d2 <= max_corr^2; ties resolve to the lowest target index.  A result
// is carried as one 64-bit key (d2 bits << 32 | target index): non-negative
// floats order like unsigned ints, so "min over keys" is the exact arg-min with
// the tie rule built in, and "no match" is the initial key (threshold, ~0).
//
// Three kernels:
//   k_nn_grid_thread  level 0, one lane per query: probes the 3x3x3 cell block as
//                     9 contiguous x-rows of the cell-sorted target, nearest rows
//                     first, pruning rows by their AABB distance.
//   k_nn_grid_wave    levels >= 1, one wavefront per query still unresolved by the
//                     finer level (its ring did not certify the minimum): lanes
//                     stride the candidate rows with coalesced float4 loads.
//   k_nn_brute        LDS-tiled all-pairs search (small clouds / cross-check).
#include "wm_internal.hpp"

namespace wm {

__device__ __forceinline__ unsigned long long make_key(float d2, unsigned idx) {
    return ((unsigned long long) __float_as_uint(d2) << 32) | idx;
}

__device__ __forceinline__ float canon_d2(float qx, float qy, float qz, const float4 &t) {
    const float dx = qx - t.x, dy = qy - t.y, dz = qz - t.z;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// PCL's float transform of a source point: ((m00*x + m01*y) + m02*z) + m03
__device__ __forceinline__ void xform(const float *T, const float4 &p, float &x, float &y,
                                      float &z) {
    x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], p.x), __fmul_rn(T[1], p.y)),
                            __fmul_rn(T[2], p.z)), T[3]);
    y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], p.x), __fmul_rn(T[5], p.y)),
                            __fmul_rn(T[6], p.z)), T[7]);
    z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], p.x), __fmul_rn(T[9], p.y)),
                            __fmul_rn(T[10], p.z)), T[11]);
}

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).
// Remap so each XCD works on one contiguous (Morton-compact) eighth of the
// queries and its private L2 holds one spatial region of the target.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblocks) {
    const unsigned per = (nblocks + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
}

// (dy, dz) of the 9 x-rows of a 3x3x3 block, nearest first
__device__ __constant__ signed char kRowDy[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
__device__ __constant__ signed char kRowDz[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};

struct Cell {
    float fx, fy, fz;  // query position in cell units
    int cx, cy, cz;
};

__device__ __forceinline__ Cell locate(const GridDev &g, float x, float y, float z) {
    Cell c;
    c.fx = (x - g.ox) * g.inv_h;
    c.fy = (y - g.oy) * g.inv_h;
    c.fz = (z - g.oz) * g.inv_h;
    // clamp before the int conversion so far-away queries cannot overflow
    c.cx = (int) floorf(fminf(fmaxf(c.fx, -4.0f), (float) g.nx + 4.0f));
    c.cy = (int) floorf(fminf(fmaxf(c.fy, -4.0f), (float) g.ny + 4.0f));
    c.cz = (int) floorf(fminf(fmaxf(c.fz, -4.0f), (float) g.nz + 4.0f));
    return c;
}

// distance (cell units) from the query to the faces of its 3x3x3 block: every
// point NOT in the block is at least this far away.
__device__ __forceinline__ float block_margin(const Cell &c) {
    float mx = fminf(c.fx - (float) (c.cx - 1), (float) (c.cx + 2) - c.fx);
    float my = fminf(c.fy - (float) (c.cy - 1), (float) (c.cy + 2) - c.fy);
    float mz = fminf(c.fz - (float) (c.cz - 1), (float) (c.cz + 2) - c.fz);
    return fminf(mx, fminf(my, mz));
}

// lower bound (cell units) of the distance from the query to row (cy+dy, cz+dz)
__device__ __forceinline__ float row_bound(const Cell &c, int dy, int dz) {
    float ry = dy == 0 ? 0.f : (dy < 0 ? c.fy - (float) c.cy : (float) (c.cy + 1) - c.fy);
    float rz = dz == 0 ? 0.f : (dz < 0 ? c.fz - (float) c.cz : (float) (c.cz + 1) - c.fz);
    return sqrtf(ry * ry + rz * rz);
}

__device__ __forceinline__ bool certified(const GridDev &g, const Cell &c, float best_d2,
                                          float thr_d2) {
    // resolved when the block provably contains the minimum, or when it covers
    // the whole acceptance radius (nothing outside can be <= threshold)
    const float m = (block_margin(c) - g.slack) * g.h;
    if (m <= 0.f) return false;
    const float m2 = m * m;
    return best_d2 <= m2 || thr_d2 <= m2;
}

// --------------------------------------------------------------- level 0
__global__ void __launch_bounds__(kBlock)
    k_nn_grid_thread(GridDev g, const float4 *__restrict__ src, unsigned n,
                     const IcpDevState *__restrict__ st, float thr_d2,
                     unsigned long long *__restrict__ keys, unsigned *__restrict__ queue,
                     unsigned *__restrict__ queue_count, int last_level) {
    if (st->done) return;
    const unsigned i = xcd_remap(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    float qx, qy, qz;
    xform(st->Tf, p, qx, qy, qz);
    const Cell c = locate(g, qx, qy, qz);
    unsigned long long best = make_key(thr_d2, kNoIdx);
    const int x0 = max(c.cx - 1, 0), x1 = min(c.cx + 1, g.nx - 1);
    if (x0 <= x1) {
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
            const int dy = kRowDy[r], dz = kRowDz[r];
            const int yy = c.cy + dy, zz = c.cz + dz;
            if (yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz) continue;
            if (r > 0) {
                const float lb = (row_bound(c, dy, dz) - g.slack) * g.h;
                if (lb > 0.f && lb * lb > __uint_as_float((unsigned) (best >> 32))) continue;
            }
            const size_t base = ((size_t) zz * g.ny + yy) * g.nx;
            const unsigned s = g.cell_start[base + x0];
            const unsigned e = g.cell_start[base + x1 + 1];
            for (unsigned j = s; j < e; ++j) {
                const float4 t = g.pts[j];
                const unsigned long long k = make_key(canon_d2(qx, qy, qz, t), __float_as_uint(t.w));
                best = k < best ? k : best;
            }
        }
    }
    keys[i] = best;
    if (!last_level &&
        !certified(g, c, __uint_as_float((unsigned) (best >> 32)), thr_d2)) {
        const unsigned slot = atomicAdd(queue_count, 1u);
        queue[slot] = i;
    }
}

// ------------------------------------------------------------- levels >= 1
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

__global__ void __launch_bounds__(kBlock)
    k_nn_grid_wave(GridDev g, const float4 *__restrict__ src, const IcpDevState *__restrict__ st,
                   float thr_d2, unsigned long long *__restrict__ keys,
                   const unsigned *__restrict__ qin, const unsigned *__restrict__ qin_count,
                   unsigned *__restrict__ qout, unsigned *__restrict__ qout_count,
                   int last_level) {
    if (st->done) return;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kBlock) >> 6;
    const unsigned cnt = *qin_count;
    for (unsigned w = wave; w < cnt; w += nwaves) {
        const unsigned i = qin[w];
        const float4 p = src[i];
        float qx, qy, qz;
        xform(st->Tf, p, qx, qy, qz);
        const Cell c = locate(g, qx, qy, qz);
        unsigned long long best = keys[i];  // upper bound from the finer level
        const int x0 = max(c.cx - 1, 0), x1 = min(c.cx + 1, g.nx - 1);
        if (x0 <= x1) {
#pragma unroll 1
            for (int r = 0; r < 9; ++r) {
                const int dy = kRowDy[r], dz = kRowDz[r];
                const int yy = c.cy + dy, zz = c.cz + dz;
                if (yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz) continue;
                best = wave_min_u64(best);  // wave-uniform bound for the prune
                const float lb = (row_bound(c, dy, dz) - g.slack) * g.h;
                if (lb > 0.f && lb * lb > __uint_as_float((unsigned) (best >> 32))) continue;
                const size_t base = ((size_t) zz * g.ny + yy) * g.nx;
                const unsigned s = g.cell_start[base + x0];
                const unsigned e = g.cell_start[base + x1 + 1];
                for (unsigned j = s + lane; j < e; j += 64u) {
                    const float4 t = g.pts[j];
                    const unsigned long long k =
                        make_key(canon_d2(qx, qy, qz, t), __float_as_uint(t.w));
                    best = k < best ? k : best;
                }
            }
        }
        best = wave_min_u64(best);
        if (lane == 0) {
            keys[i] = best;
            if (!last_level &&
                !certified(g, c, __uint_as_float((unsigned) (best >> 32)), thr_d2)) {
                const unsigned slot = atomicAdd(qout_count, 1u);
                qout[slot] = i;
            }
        }
    }
}

// ----------------------------------------------------------- brute force
constexpr int kBruteTile = 1024;

__global__ void __launch_bounds__(kBlock) k_init_keys(unsigned long long *keys, unsigned n,
                                                       const IcpDevState *st, float thr_d2) {
    if (st->done) return;
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) keys[i] = make_key(thr_d2, kNoIdx);
}

// grid = (ceil(n / 256), splits); block y scans target slice y.  Every lane
// holds one (transformed) query; the target streams through a float4 LDS tile
// that all lanes read at the same address (LDS broadcast, no bank conflicts).
__global__ void __launch_bounds__(kBlock)
    k_nn_brute(const float4 *__restrict__ tgt, unsigned m, const float4 *__restrict__ src,
               unsigned n, const IcpDevState *__restrict__ st, float thr_d2,
               unsigned long long *__restrict__ keys) {
    if (st->done) return;
    __shared__ float4 tile[kBruteTile];
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const unsigned per = (m + gridDim.y - 1) / gridDim.y;
    const unsigned m0 = blockIdx.y * per, m1 = min(m0 + per, m);
    float qx = 0, qy = 0, qz = 0;
    if (i < n) xform(st->Tf, src[i], qx, qy, qz);
    unsigned long long best = make_key(thr_d2, kNoIdx);
    for (unsigned t0 = m0; t0 < m1; t0 += kBruteTile) {
        const unsigned cnt = min((unsigned) kBruteTile, m1 - t0);
        __syncthreads();
        for (unsigned k = threadIdx.x; k < cnt; k += kBlock) tile[k] = tgt[t0 + k];
        __syncthreads();
        for (unsigned k = 0; k < cnt; ++k) {
            const float4 t = tile[k];
            // non-finite target points were packed as NaN: their key (0x7FC0....)
            // exceeds every finite threshold and never wins
            const unsigned long long key = make_key(canon_d2(qx, qy, qz, t), __float_as_uint(t.w));
            best = key < best ? key : best;
        }
    }
    if (i < n && best < make_key(thr_d2, kNoIdx)) atomicMin(&keys[i], best);
}

// largest float whose value, compared as PCL does ((double) d2 > max_corr^2 ->
// reject), is still accepted
float threshold_d2(double max_corr) {
    const double m2 = max_corr * max_corr;
    if (!(m2 < 3.0e38)) return 3.0e38f;
    float f = (float) m2;
    if ((double) f > m2) f = nextafterf(f, 0.0f);
    return f;
}

int launch_nn_grid(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2) {
    const unsigned n = (unsigned) ctx->n_src;
    if (n == 0) return WM_OK;
    const IcpDevState *st = ctx->d_state.as<IcpDevState>();
    unsigned long long *keys = ctx->keys.as<unsigned long long>();
    unsigned *qcount = const_cast<unsigned *>(st->queue_count);
    unsigned *qa = ctx->queue_a.as<unsigned>(), *qb = ctx->queue_b.as<unsigned>();
    const int L = ctx->n_levels;
    unsigned blocks = (n + kBlock - 1) / kBlock;
    blocks = (blocks + 7u) & ~7u;  // xcd_remap needs a multiple of 8
    if (ev0) WM_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    hipLaunchKernelGGL(k_nn_grid_thread, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       ctx->levels[0].d, ctx->src_sorted.as<float4>(), n, st, thr_d2, keys, qa,
                       qcount + 1, L == 1 ? 1 : 0);
    if (ev1) WM_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    for (int l = 1; l < L; ++l) {
        unsigned *qin = (l & 1) ? qa : qb, *qout = (l & 1) ? qb : qa;
        hipLaunchKernelGGL(k_nn_grid_wave, dim3(2048), dim3(kBlock), 0, ctx->stream,
                           ctx->levels[l].d, ctx->src_sorted.as<float4>(), st, thr_d2, keys, qin,
                           qcount + l, qout, qcount + l + 1, l == L - 1 ? 1 : 0);
    }
    if (ev2) WM_HIP(ctx, hipEventRecord(ev2, ctx->stream));
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

int launch_nn_brute(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1) {
    const unsigned n = (unsigned) ctx->n_src, m = (unsigned) ctx->n_tgt_input;
    if (n == 0) return WM_OK;
    const IcpDevState *st = ctx->d_state.as<IcpDevState>();
    unsigned long long *keys = ctx->keys.as<unsigned long long>();
    const unsigned bx = (n + kBlock - 1) / kBlock;
    unsigned splits = 1;
    if (m > 0) {
        // enough workgroups to fill 256 CUs several times over
        splits = (2048 + bx - 1) / bx;
        const unsigned max_splits = (m + kBruteTile - 1) / kBruteTile;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    if (ev0) WM_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    hipLaunchKernelGGL(k_init_keys, dim3(bx), dim3(kBlock), 0, ctx->stream, keys, n, st, thr_d2);
    if (m > 0)
        hipLaunchKernelGGL(k_nn_brute, dim3(bx, splits), dim3(kBlock), 0, ctx->stream,
                           ctx->tgt_orig.as<float4>(), m, ctx->src_sorted.as<float4>(), n, st,
                           thr_d2, keys);
    if (ev1) WM_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

}  // namespace wm
