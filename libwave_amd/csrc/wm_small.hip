// wm_small.hip -- whole registrations of SMALL clouds inside one compute unit, many per launch.
//
// wave::MultiMatcher (wave_matching/include/wave/matching/multi_matcher.hpp:29-96, worker loop
// impl/multi_matcher_impl.hpp:45-53) is the reference's way to throughput: one single-threaded PCL
// registration per core.  A 10 000-point pair (BASELINE configs[0]) is 157 wavefronts of work per
// iteration -- 3 % of an MI355X -- and one registration through wm_icp_align is a chain of ~100
// DEPENDENT launches of 3-7 us each: sixteen workers on sixteen streams top out at ~5 000
// registrations/s whatever the number of hardware queues (measured: 4 / 8 / 16 / 24 queues -> 5 200 /
// 3 400 / 2 900 / 2 400), with the chip mostly idle.  Here the unit of parallelism is the pair, not
// the point: ONE workgroup of 1 024 lanes runs one whole registration --
//   ICPMatcher::match()'s full-resolution branch       wave_matching/src/icp.cpp:123-131
//   + the estimator whose result estimateInfo() keeps  icp.cpp:135-142 -> estimateLUMold,
//                                                      icp_pcl_functions.cpp:51-179
// -- and a launch carries one workgroup per queued pair, so 256 pairs fill the 256 CUs and nothing
// crosses a launch boundary:
//   0. the source is put in cell order (bitonic sort of unique keys in LDS; a packed copy in HBM
//      scratch), so that neighbouring lanes search neighbouring cells;
//   1. the target cloud is counting-sorted by cell INTO LDS (x, y, z floats + a 16-bit original
//      index per point: 14 B x 10 000 points = 140 kB of the CU's 160 KB; 8 192 cells, 16-bit starts;
//      the grid covers the CORE of the cloud, outliers are clamped into its border cells) --
//      k_icp_small<true>; targets of up to 65 535 points keep that cell-sorted copy in HBM scratch
//      (L2 / Infinity Cache resident; only the cell starts live in LDS) -- k_icp_small<false>;
//   2. every iteration chunks of 64 queries go to whichever wavefront is free: PCL's float transform,
//      an exact 1-NN search over the cell rows overlapping a certified ball (radius = distance to the
//      previous iteration's match under the new pose, so one scan certifies), the same 64-bit
//      (d2 bits, index) arg-min key as wm_nn.hip -- hence the same correspondences, bit for bit --
//      and the iteration's 17 sums in double, reduced per chunk so that the result does not depend
//      on which wavefront took which chunk;
//   3. a fixed-order sum of the chunk rows, then lane 0 runs the very solve + stopping rules of the
//      big path (icp_apply_stats, wm_icp_step.hpp) on a state that lives in LDS;
//   4. after the last iteration the LUMold normal equations and residual, same arithmetic as
//      wm_info.hip.
// Latency of ONE registration is worse than wm_icp_align's (one CU instead of a hundred); this path
// is for queues of pairs.  The voxel-filtered branches of match() are built on top of it in
// wm_batch.hip (small_run below is what they call, scale by scale).
#include "wm_icp_step.hpp"
#include "wm_internal.hpp"
#include "wm_wave.hpp"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <type_traits>
#include <vector>

namespace wm {

constexpr int kSmThreads = 1024;
constexpr int kSmWaves = kSmThreads / 64;
constexpr int kSmMaxTgt = WM_BATCH_LDS_TARGET_POINTS;  // 10 000: the target lives in LDS
constexpr int kSmMaxTgtHbm = WM_BATCH_MAX_TARGET_POINTS;  // 65 535: ... in the pair's HBM scratch (16-bit slots and indices)
constexpr int kSmPer = (kSmMaxTgt + kSmThreads - 1) / kSmThreads;
constexpr int kSmCells = 8192;       // LDS variant: 16-bit cell starts, 16 KB
constexpr int kSmCellsHbm = 63 * 1024;  // HBM variant: as many as 16-bit starts fit the 128 KB of LDS the sort leaves
                                        // (8 192 / 32 768 / 64 512 cells: 11.2 / 9.1 / 8.6 ms per 256 pairs of 30 000 points)
constexpr int kSmRedW = 18;
// source clouds of up to 2^bits points are put in cell order by a sort in LDS: 64 KB of keys in the space the
// LDS target will occupy (14-bit index in the key), 128 KB when the target stays in HBM (15-bit index)
template <bool INLDS>
constexpr unsigned kSmSortBits = INLDS ? 14u : 15u;
static_assert(kSmCells % kSmThreads == 0 && kSmCellsHbm % kSmThreads == 0, "layout");

struct SmallPair {  // one registration of the batch (device table)
    const unsigned char *src, *tgt;  // caller-layout points in device memory
    unsigned n_src, n_tgt;
    double prev_mse0;      // the stopping criteria's previous MSE this registration starts with (DBL_MAX: none)
    unsigned presorted;    // the source is in a spatial order already (a voxel filter's output: leaf order)
    unsigned pad0;
    unsigned short *seed;  // n_src entries: LDS slot of each source point's last match (0xFFFF none)
    double *csum;          // ceil(n_src / 64) rows of kAcc doubles of scratch: every 64-query chunk's sums
    // HBM variant only: the cell-sorted target (x, y, z per point + 4 sentinels; caller's index of each; first slot of
    // every cell + end) and, during the build, every point's (cell << 16 | rank)
    float *txyz;
    unsigned short *tidx;
    unsigned *tcs;
    unsigned *trank;
    float4 *sorted;        // n_src entries of scratch: the source packed (x, y, z, index bits), in cell order up to kSmSortMax points
};

struct SmallParams {
    unsigned stride;        // bytes between points
    float thr_d2;           // ICP gate: d2 <= max_corr^2      (pcl CorrespondenceEstimation)
    float thr_d2_strict;    // LUMold gate: d2 < max_corr^2    (icp_pcl_functions.cpp:76-80)
    float r0_cells;         // first radius of an unseeded search, in cells
    int with_info;          // 0: none; 1: LUMold
    int iter_cap;           // safety net over the state's own max_iter
};

struct SmallOut {
    double T[16];
    double info[36];
    double mse, prev_mse;
    int iterations, converged, state, n_corr;
    int info_degenerate, n_target_valid;
    float cell;
    int pad;
    unsigned long long cyc[4];  // developer: shader-clock cycles of the LAST iteration's query loop / reduction / solve, and of the set-up
};

#define WM_SMALL_LDS_COMMON                                                                                  \
    double red[kSmWaves][kSmRedW];                                                                             \
    double sum[kSmRedW];                                                                                       \
    float boxf[kSmWaves][8];                                                                                   \
    unsigned wsum[kSmWaves];                                                                                   \
    unsigned ticket; /* next chunk of 64 queries (the iterations hand them to whichever wavefront is free) */ \
    float ox, oy, oz, h, inv_h;                                                                                \
    int nx, ny, nz;                                                                                            \
    double D[6];                                                                                               \
    IcpDevState st;

struct SmallLds {  // the target in LDS
    float xyz[3 * (kSmMaxTgt + 4)];                  // cell-sorted target, (x, y, z) per point (+ 4 far sentinels)
    unsigned short idx[kSmMaxTgt + 4];               // caller's index of each
    unsigned short cstart[kSmCells + 8];             // first slot of every cell (+ end)
    WM_SMALL_LDS_COMMON
};
struct SmallLdsH {  // the target in HBM scratch: LDS only holds the source's sort and the shared state
    unsigned sortbuf[1u << 15];
    WM_SMALL_LDS_COMMON
};
#undef WM_SMALL_LDS_COMMON
__device__ __forceinline__ unsigned *sm_sort_region(SmallLds &L) { return reinterpret_cast<unsigned *>(L.xyz); }  // (120 KB)
__device__ __forceinline__ unsigned *sm_sort_region(SmallLdsH &L) { return L.sortbuf; }

// Where the cell-sorted target is read from.  c(j, k) = coordinate k of slot j, idx(j) = the caller's index of
// the point in slot j, cs(c) = first slot of cell c.
struct SmTgtLds {
    const SmallLds *L;
    __device__ __forceinline__ float c(unsigned j, unsigned k) const { return L->xyz[3u * j + k]; }
    __device__ __forceinline__ unsigned idx(unsigned j) const { return L->idx[j]; }
    __device__ __forceinline__ unsigned cs(unsigned cell) const { return L->cstart[cell]; }
};
struct SmTgtHbm {  // (global, not flat, loads: L2 / MALL resident scratch); the cell starts are copied into LDS
    const __attribute__((address_space(1))) float *xyz;
    const __attribute__((address_space(1))) unsigned short *idxp;
    const unsigned short *csl;  // 16-bit starts in the LDS the source's sort used
    __device__ __forceinline__ float c(unsigned j, unsigned k) const { return xyz[3u * j + k]; }
    __device__ __forceinline__ unsigned idx(unsigned j) const { return idxp[j]; }
    __device__ __forceinline__ unsigned cs(unsigned cell) const { return csl[cell]; }
};
static_assert(sizeof(SmallLds) <= 160 * 1024 && sizeof(SmallLdsH) <= 160 * 1024, "one workgroup's LDS");

__device__ __forceinline__ unsigned long long sm_key(float d2, unsigned idx) {
    return ((unsigned long long) __float_as_uint(d2) << 32) | idx;
}
__device__ __forceinline__ float sm_d2(float qx, float qy, float qz, float tx, float ty, float tz) {
    const float dx = qx - tx, dy = qy - ty, dz = qz - tz;  // (as canon_d2, wm_nn.hip)
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
// PCL's float transform of a source point: ((m00*x + m01*y) + m02*z) + m03
__device__ __forceinline__ void sm_xform(const float (&T)[12], float px, float py, float pz, float &x,
                                         float &y, float &z) {
    x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], px), __fmul_rn(T[1], py)), __fmul_rn(T[2], pz)), T[3]);
    y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], px), __fmul_rn(T[5], py)), __fmul_rn(T[6], pz)), T[7]);
    z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], px), __fmul_rn(T[9], py)), __fmul_rn(T[10], pz)), T[11]);
}

struct SmGrid {  // (wave-uniform: lives in scalar registers)
    float ox, oy, oz, inv_h, h;
    int nx, ny, nz;
};

__device__ __forceinline__ int sm_cell1(float v, float o, float inv_h, int n) {
    const float f = floorf(fminf(fmaxf((v - o) * inv_h, -1.0f), (float) n));
    const int c = (int) f;
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// Exact nearest neighbour of (qx, qy, qz) among the LDS-resident points: every cell row that meets
// the box around ball(q, r) is walked; the result is certified when the best distance is inside r
// (then nothing outside the box can beat it), else r doubles up to rmax.  `best` comes in as the
// gate (or the seed's key); candidates only ever lower it.
template <class TG>
__device__ __forceinline__ void sm_search(const TG &tg, const SmGrid &g, float qx, float qy, float qz,
                                          float r, float rmax, unsigned long long &best, unsigned &bslot) {
    for (;;) {
        // (the box is a little larger than the ball: it has to hold against the rounding of q -+ r
        // at the cloud's coordinates, half an ulp of |q|)
        const float rb = r * 1.001f + 1e-6f + 2.5e-7f * fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz)));
        const int x0 = sm_cell1(qx - rb, g.ox, g.inv_h, g.nx), x1 = sm_cell1(qx + rb, g.ox, g.inv_h, g.nx);
        const int y0 = sm_cell1(qy - rb, g.oy, g.inv_h, g.ny), y1 = sm_cell1(qy + rb, g.oy, g.inv_h, g.ny);
        const int z0 = sm_cell1(qz - rb, g.oz, g.inv_h, g.nz), z1 = sm_cell1(qz + rb, g.oz, g.inv_h, g.nz);
        unsigned bhi = (unsigned) (best >> 32);
        for (int cz = z0; cz <= z1; ++cz)
            for (int cy = y0; cy <= y1; ++cy) {
                const int row = (cz * g.ny + cy) * g.nx;
                const unsigned s = tg.cs((unsigned) (row + x0)), e = tg.cs((unsigned) (row + x1 + 1));
                // four candidates per trip, their twelve LDS reads in flight together.  No clamp at the end of
                // the run: what follows it are real points of the next cells (or the far sentinels
                // after the last point), and an extra real candidate cannot hurt an arg-min over the
                // target.  The update is a rare branch behind one unsigned min of the four d2.
                for (unsigned j = s; j < e; j += 4u) {
                    float d2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) d2[u] = sm_d2(qx, qy, qz, tg.c(j + (unsigned) u, 0), tg.c(j + (unsigned) u, 1), tg.c(j + (unsigned) u, 2));
                    const unsigned m01 = min(__float_as_uint(d2[0]), __float_as_uint(d2[1]));
                    const unsigned m23 = min(__float_as_uint(d2[2]), __float_as_uint(d2[3]));
                    if (min(m01, m23) <= bhi) {  // (d2 >= 0: bit order = numeric order)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const unsigned long long k = sm_key(d2[u], tg.idx(j + (unsigned) u));
                            if (k < best) {
                                best = k;
                                bslot = j + (unsigned) u;
                            }
                        }
                        bhi = (unsigned) (best >> 32);
                    }
                }
            }
        if (r >= rmax || sqrtf(__uint_as_float(bhi)) <= r) return;
        r = fminf(2.0f * r, rmax);
    }
}

template <int N, class LDS>
__device__ __forceinline__ void sm_block_sum(double (&a)[N], LDS &L, unsigned tid) {
    static_assert(N <= kSmRedW, "reduction scratch");
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    const unsigned lane = tid & 63u, wave = tid >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < N; ++k) L.red[wave][k] = a[k];
    __syncthreads();
    if (tid < (unsigned) N) {
        double s = 0;
        for (int w = 0; w < kSmWaves; ++w) s += L.red[w][tid];  // fixed order
        L.sum[tid] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ bool sm_load_point(const unsigned char *base, unsigned i, unsigned stride, float &x,
                                              float &y, float &z) {
    const float *p = reinterpret_cast<const float *>(base + (size_t) i * stride);
    x = p[0];
    y = p[1];
    z = p[2];
    return isfinite(x) && isfinite(y) && isfinite(z);  // (non-finite points are dropped, as k_pack does)
}

// The grid of a cloud: the finest cubic cells, at most `max_cells` of them, over the CORE of its finite
// points -- their bounding box cut back to mean +- 2.5 standard deviations per axis.  Points outside
// land in the border cells (cell coordinates are clamped, for points and for search boxes alike, so
// any box gives exact searches); what the cut buys is cell size: a lidar scan has a few returns at
// 100 m and most of its points within 20 m, and cells sized for the full box hold hundreds of points
// where the points are.  Every lane returns the same grid; `count` = number of finite points.
template <class LDS>
__device__ __forceinline__ SmGrid sm_fit_grid(LDS &L, const unsigned char *pts, unsigned n, unsigned stride,
                                              int max_cells, unsigned tid, unsigned &count) {
    const unsigned lane = tid & 63u, wave = tid >> 6;
    float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    double m1[3] = {0.0, 0.0, 0.0}, m2[3] = {0.0, 0.0, 0.0};
    unsigned cnt = 0;
    for (unsigned i = tid; i < n; i += kSmThreads) {
        float x, y, z;
        if (sm_load_point(pts, i, stride, x, y, z)) {
            lo0 = fminf(lo0, x), lo1 = fminf(lo1, y), lo2 = fminf(lo2, z);
            hi0 = fmaxf(hi0, x), hi1 = fmaxf(hi1, y), hi2 = fmaxf(hi2, z);
            m1[0] += x, m1[1] += y, m1[2] += z;
            m2[0] += (double) x * x, m2[1] += (double) y * y, m2[2] += (double) z * z;
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo0 = fminf(lo0, __shfl_xor(lo0, off)), lo1 = fminf(lo1, __shfl_xor(lo1, off));
        lo2 = fminf(lo2, __shfl_xor(lo2, off)), hi0 = fmaxf(hi0, __shfl_xor(hi0, off));
        hi1 = fmaxf(hi1, __shfl_xor(hi1, off)), hi2 = fmaxf(hi2, __shfl_xor(hi2, off));
        cnt += __shfl_xor(cnt, off);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            m1[k] += __shfl_xor(m1[k], off);
            m2[k] += __shfl_xor(m2[k], off);
        }
    }
    __syncthreads();  // (the scratch may still be read from the previous call)
    if (lane == 0) {
        L.boxf[wave][0] = lo0, L.boxf[wave][1] = lo1, L.boxf[wave][2] = lo2;
        L.boxf[wave][3] = hi0, L.boxf[wave][4] = hi1, L.boxf[wave][5] = hi2;
        L.wsum[wave] = cnt;
#pragma unroll
        for (int k = 0; k < 3; ++k) L.red[wave][k] = m1[k], L.red[wave][3 + k] = m2[k];
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kSmWaves; ++w) {
            lo0 = fminf(lo0, L.boxf[w][0]), lo1 = fminf(lo1, L.boxf[w][1]), lo2 = fminf(lo2, L.boxf[w][2]);
            hi0 = fmaxf(hi0, L.boxf[w][3]), hi1 = fmaxf(hi1, L.boxf[w][4]), hi2 = fmaxf(hi2, L.boxf[w][5]);
            cnt += L.wsum[w];
            for (int k = 0; k < 3; ++k) m1[k] += L.red[w][k], m2[k] += L.red[w][3 + k];
        }
        int nx = 1, ny = 1, nz = 1;
        float h = 1.0f;
        if (lo0 <= hi0) {  // at least one finite point
            float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
            for (int k = 0; k < 3; ++k) {  // the core: mean +- 2.5 sigma, inside the bounding box
                const double mean = m1[k] / cnt, var = m2[k] / cnt - mean * mean;
                const double sd = var > 0.0 ? sqrt(var) : 0.0;
                const float a = fmaxf(lo[k], (float) (mean - 2.5 * sd)), b = fminf(hi[k], (float) (mean + 2.5 * sd));
                if (a <= b) lo[k] = a, hi[k] = b;
            }
            lo0 = lo[0], lo1 = lo[1], lo2 = lo[2], hi0 = hi[0], hi1 = hi[1], hi2 = hi[2];
            const float ex = hi0 - lo0, ey = hi1 - lo1, ez = hi2 - lo2;
            const float big = fmaxf(ex, fmaxf(ey, ez));
            const float tiny = fmaxf(big * 1e-6f, 1e-30f);
            h = cbrtf(fmaxf(ex, tiny) * fmaxf(ey, tiny) * fmaxf(ez, tiny) / (float) max_cells);
            h = fmaxf(h, tiny);
            for (int it = 0; it < 400; ++it) {
                const float fx = floorf(ex / h) + 1.0f, fy = floorf(ey / h) + 1.0f, fz = floorf(ez / h) + 1.0f;
                if (fx * fy * fz <= (float) max_cells) {
                    nx = (int) fx, ny = (int) fy, nz = (int) fz;
                    break;
                }
                h *= 1.06f;
            }
            if (nx * ny * nz > max_cells || !(h > 0.0f) || !isfinite(h)) nx = ny = nz = 1, h = fmaxf(big, 1.0f);
        } else {
            lo0 = lo1 = lo2 = 0.0f;
        }
        L.ox = lo0, L.oy = lo1, L.oz = lo2, L.h = h, L.inv_h = 1.0f / h;
        L.nx = nx, L.ny = ny, L.nz = nz;
        L.wsum[0] = cnt;
    }
    __syncthreads();
    SmGrid g;
    g.ox = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, L.ox)));
    g.oy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, L.oy)));
    g.oz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, L.oz)));
    g.h = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, L.h)));
    g.inv_h = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, L.inv_h)));
    g.nx = __builtin_amdgcn_readfirstlane(L.nx);
    g.ny = __builtin_amdgcn_readfirstlane(L.ny);
    g.nz = __builtin_amdgcn_readfirstlane(L.nz);
    count = (unsigned) __builtin_amdgcn_readfirstlane((int) L.wsum[0]);
    return g;
}

// One query as the iterations read it: the packed source point (x, y, z, index bits) and the LDS slot of
// its last match, both from the pair's HBM scratch (global, not flat, loads: these never point into LDS).
struct SmQuery {
    float x, y, z;
    unsigned seed;
};
typedef float sm_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ SmQuery sm_fetch(const SmallPair &pr, unsigned q) {
    const sm_f4v v = ((const __attribute__((address_space(1))) sm_f4v *) pr.sorted)[q];
    SmQuery c;
    c.x = v.x, c.y = v.y, c.z = v.z;
    c.seed = ((const __attribute__((address_space(1))) unsigned short *) pr.seed)[q];
    return c;
}
// for (q = tid; q < n_q; q += 1 024) body(q, query) -- with the NEXT query's two loads issued before the
// current one is searched: a lane has ten queries per iteration and nothing else to hide their HBM
// round trips behind (16 wavefronts per CU; without this 70 us of a 100 us iteration were those waits)
template <class F>
__device__ __forceinline__ void sm_for_queries(const SmallPair &pr, unsigned n_q, unsigned tid, F &&body) {
    unsigned q = tid;
    bool has = q < n_q;
    SmQuery cur = {0.f, 0.f, 0.f, 0xFFFFu};
    if (has) cur = sm_fetch(pr, q);
    while (has) {
        const unsigned qn = q + kSmThreads;
        const bool hn = qn < n_q;
        SmQuery nxt = cur;
        if (hn) nxt = sm_fetch(pr, qn);
        body(q, cur);
        q = qn;
        cur = nxt;
        has = hn;
    }
}

__device__ __forceinline__ unsigned sm_cell_of(const SmGrid &g, float x, float y, float z) {
    return (unsigned) ((sm_cell1(z, g.oz, g.inv_h, g.nz) * g.ny + sm_cell1(y, g.oy, g.inv_h, g.ny)) * g.nx +
                       sm_cell1(x, g.ox, g.inv_h, g.nx));
}

template <bool INLDS>
__global__ void __launch_bounds__(kSmThreads)
    k_icp_small(const SmallPair *__restrict__ pairs, SmallParams P, IcpDevState st0, SmallOut *__restrict__ out) {
    using LDS = typename std::conditional<INLDS, SmallLds, SmallLdsH>::type;
    using TG = typename std::conditional<INLDS, SmTgtLds, SmTgtHbm>::type;
    constexpr unsigned kSortMax = 1u << kSmSortBits<INLDS>;
    constexpr int kCells = INLDS ? kSmCells : kSmCellsHbm;
    __shared__ LDS L;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const SmallPair pr = pairs[blockIdx.x];
    const unsigned long long cyc_begin = clock64();
    constexpr unsigned kMaxT = INLDS ? (unsigned) kSmMaxTgt : (unsigned) kSmMaxTgtHbm;
    const unsigned n_tgt = pr.n_tgt < kMaxT ? pr.n_tgt : kMaxT;  // (host checked)

    if (tid == 0) {
        L.st = st0;
        L.st.prev_mse = pr.prev_mse0;
        L.ticket = kSmWaves;
    }

    // ---- 0. the source in cell order (of a grid over its own bounding box): neighbouring lanes then
    // search neighbouring cells -- the same LDS rows (broadcast reads instead of bank conflicts) for a
    // similar number of trips.  The order is a bitonic sort, in the LDS the target will occupy, of the
    // unique keys (cell << 14 or 15 | index): deterministic, so the sums of a pair do not depend on timing.
    // The sorted copy (x, y, z, index bits) lives in HBM scratch; clouds beyond 16 384 (32 768) points keep
    // the caller's order, and so do sources that come out of a voxel filter (leaf order IS a cell order)
    // (packed all the same: the iterations read one layout).
    unsigned n_q = pr.n_src;  // queries per iteration
    if (pr.n_src > kSortMax || pr.presorted) {
        for (unsigned i = tid; i < pr.n_src; i += kSmThreads) {
            float x, y, z;
            if (!sm_load_point(pr.src, i, P.stride, x, y, z)) x = y = z = __builtin_nanf("");
            pr.sorted[i] = make_float4(x, y, z, __uint_as_float(i));
        }
        __threadfence_block();
        __syncthreads();
    } else {
        unsigned n_fin;
        const SmGrid gs = sm_fit_grid(L, pr.src, pr.n_src, P.stride, kSmCells, tid, n_fin);
        unsigned N = 2048;
        while (N < pr.n_src) N <<= 1;
        unsigned *K = sm_sort_region(L);
        for (unsigned i = tid; i < N; i += kSmThreads) {
            unsigned key = ~0u;
            float x, y, z;
            if (i < pr.n_src && sm_load_point(pr.src, i, P.stride, x, y, z)) key = (sm_cell_of(gs, x, y, z) << kSmSortBits<INLDS>) | i;
            K[i] = key;
        }
        for (unsigned k = 2; k <= N; k <<= 1)
            for (unsigned j = k >> 1; j > 0; j >>= 1) {
                __syncthreads();
                for (unsigned t = tid; t < N / 2; t += kSmThreads) {
                    const unsigned i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                    const unsigned a = K[i], b = K[i + j];
                    if ((a > b) == ((i & k) == 0u)) {
                        K[i] = b;
                        K[i + j] = a;
                    }
                }
            }
        __syncthreads();
        for (unsigned q = tid; q < n_fin; q += kSmThreads) {
            const unsigned i = K[q] & (kSortMax - 1u);
            float x, y, z;
            (void) sm_load_point(pr.src, i, P.stride, x, y, z);
            pr.sorted[q] = make_float4(x, y, z, __uint_as_float(i));
        }
        n_q = n_fin;
        __threadfence_block();
        __syncthreads();
    }

    // ---- 1. the target cell-sorted -- into LDS, or into the pair's HBM scratch when it is too large for that:
    // bounding box -> grid -> counting sort by cell
    unsigned n_tgt_fin;
    const SmGrid g = sm_fit_grid(L, pr.tgt, n_tgt, P.stride, kCells, tid, n_tgt_fin);
    TG tg;
    if constexpr (INLDS) {
        tg.L = &L;
        for (unsigned c = tid; c < (kSmCells + 8) / 2; c += kSmThreads) reinterpret_cast<unsigned *>(L.cstart)[c] = 0u;
        __syncthreads();
        {
            // count: two 16-bit counters per LDS word; the returned old value is the point's rank in its cell
            unsigned cr[kSmPer];
            unsigned *cw = reinterpret_cast<unsigned *>(L.cstart);
#pragma unroll
            for (int k = 0; k < kSmPer; ++k) {
                const unsigned i = tid + (unsigned) k * kSmThreads;
                cr[k] = ~0u;
                float x, y, z;
                if (i < n_tgt && sm_load_point(pr.tgt, i, P.stride, x, y, z)) {
                    const unsigned c = sm_cell_of(g, x, y, z);
                    const unsigned sh = (c & 1u) * 16u;
                    const unsigned old = atomicAdd(&cw[c >> 1], 1u << sh);
                    cr[k] = (c << 16) | ((old >> sh) & 0xFFFFu);
                }
            }
            __syncthreads();
            // exclusive scan of the 8 192 counts, in place (8 per lane)
            constexpr int E = kSmCells / kSmThreads;
            unsigned v[E], s = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                v[e] = L.cstart[E * tid + e];
                s += v[e];
            }
            unsigned incl = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = __shfl_up(incl, off);
                if (lane >= (unsigned) off) incl += t;
            }
            if (lane == 63) L.wsum[wave] = incl;
            __syncthreads();
            unsigned run = incl - s;
            for (unsigned w = 0; w < wave; ++w) run += L.wsum[w];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                L.cstart[E * tid + e] = (unsigned short) run;
                run += v[e];
            }
            if (tid == kSmThreads - 1) L.cstart[kSmCells] = (unsigned short) run;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kSmPer; ++k) {
                if (cr[k] == ~0u) continue;
                const unsigned i = tid + (unsigned) k * kSmThreads;
                float x, y, z;
                (void) sm_load_point(pr.tgt, i, P.stride, x, y, z);
                const unsigned slot = (unsigned) L.cstart[cr[k] >> 16] + (cr[k] & 0xFFFFu);
                L.xyz[3u * slot] = x, L.xyz[3u * slot + 1u] = y, L.xyz[3u * slot + 2u] = z;
                L.idx[slot] = (unsigned short) i;
            }
            if (tid < 4u) {  // far sentinels behind the last point (see sm_search)
                const unsigned slot = n_tgt_fin + tid;
                L.xyz[3u * slot] = L.xyz[3u * slot + 1u] = L.xyz[3u * slot + 2u] = INFINITY;
                L.idx[slot] = 0xFFFFu;
            }
            __syncthreads();
        }
    } else {
        unsigned *cs = pr.tcs;
        for (unsigned c = tid; c < (unsigned) kCells + 8u; c += kSmThreads) cs[c] = 0u;
        __threadfence_block();
        __syncthreads();
        // count: the atomic's old value is the point's rank in its cell; (cell << 16 | rank) kept in scratch
        for (unsigned i = tid; i < n_tgt; i += kSmThreads) {
            unsigned cr = ~0u;
            float x, y, z;
            if (sm_load_point(pr.tgt, i, P.stride, x, y, z)) {
                const unsigned c = sm_cell_of(g, x, y, z);
                cr = (c << 16) | (atomicAdd(&cs[c], 1u) & 0xFFFFu);
            }
            pr.trank[i] = cr;
        }
        __threadfence_block();
        __syncthreads();
        // exclusive scan of the 64 512 counts, in place (63 consecutive per lane)
        constexpr int E = kSmCellsHbm / kSmThreads;
        unsigned sum = 0;
        for (int e = 0; e < E; ++e) sum += ((const __attribute__((address_space(1))) unsigned *) cs)[E * tid + e];
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off);
            if (lane >= (unsigned) off) incl += t;
        }
        if (lane == 63) L.wsum[wave] = incl;
        __syncthreads();
        unsigned run = incl - sum;
        for (unsigned w = 0; w < wave; ++w) run += L.wsum[w];
        for (int e = 0; e < E; ++e) {
            const unsigned v = cs[E * tid + e];
            cs[E * tid + e] = run;
            run += v;
        }
        if (tid == kSmThreads - 1) cs[kCells] = run;
        __threadfence_block();
        __syncthreads();
        for (unsigned i = tid; i < n_tgt; i += kSmThreads) {
            const unsigned cr = pr.trank[i];
            if (cr == ~0u) continue;
            float x, y, z;
            (void) sm_load_point(pr.tgt, i, P.stride, x, y, z);
            const unsigned slot = cs[cr >> 16] + (cr & 0xFFFFu);
            pr.txyz[3u * slot] = x, pr.txyz[3u * slot + 1u] = y, pr.txyz[3u * slot + 2u] = z;
            pr.tidx[slot] = (unsigned short) i;
        }
        if (tid < 4u) {  // far sentinels behind the last point (see sm_search)
            const unsigned slot = n_tgt_fin + tid;
            pr.txyz[3u * slot] = pr.txyz[3u * slot + 1u] = pr.txyz[3u * slot + 2u] = INFINITY;
            pr.tidx[slot] = 0xFFFFu;
        }
        __threadfence_block();
        __syncthreads();
        tg.xyz = (const __attribute__((address_space(1))) float *) pr.txyz;
        tg.idxp = (const __attribute__((address_space(1))) unsigned short *) pr.tidx;
        // the 64 513 cell starts fit 16 bits (<= 65 535 points): into the LDS the source's sort has left
        unsigned short *csl = reinterpret_cast<unsigned short *>(L.sortbuf);
        for (unsigned c = tid; c <= (unsigned) kCells; c += kSmThreads) csl[c] = (unsigned short) cs[c];
        __syncthreads();
        tg.csl = csl;
    }
    const float rmax = sqrtf(P.thr_d2) * 1.0001f + 1e-6f;
    const unsigned nchunks = (n_q + 63u) / 64u;
    const unsigned long long cyc_setup = clock64() - cyc_begin;
    unsigned long long cyc_loop = 0, cyc_red = 0, cyc_solve = 0;

    // ---- 2./3. the iterations
    for (int guard = 0; guard < P.iter_cap; ++guard) {
        if (L.st.done) break;  // (uniform: written by lane 0 before the barrier that ended the last trip)
        float Tf[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Tf[k] = L.st.Tf[k];
        const bool have_prev = L.st.have_prev != 0;
        const int mode = L.st.mode;
        double a[kAcc];
        const unsigned long long t0 = clock64();
        auto one_query = [&](unsigned q, const SmQuery &c) {
            if (!(c.x == c.x)) return;  // (a non-finite point of a cloud too large to be sorted)
            float qx, qy, qz;
            sm_xform(Tf, c.x, c.y, c.z, qx, qy, qz);
            unsigned long long best = sm_key(P.thr_d2, kNoIdx);
            unsigned bslot = 0xFFFFu;
            float r = P.r0_cells * g.h;
            if (have_prev) {
                // the point matched in the previous iteration is a real candidate: its distance under
                // the NEW pose bounds the new neighbour's, so one scan of that ball certifies
                const unsigned ps = c.seed;
                r = rmax;
                if (ps != 0xFFFFu) {
                    const float d2b = sm_d2(qx, qy, qz, tg.c(ps, 0), tg.c(ps, 1), tg.c(ps, 2));
                    if (d2b <= P.thr_d2) {
                        best = sm_key(d2b, tg.idx(ps));
                        bslot = ps;
                        r = fmaxf(sqrtf(d2b) * 1.0001f + 1e-6f, 0.05f * g.h);
                    }
                }
            }
            sm_search(tg, g, qx, qy, qz, fminf(r, rmax), rmax, best, bslot);
            const bool matched = (unsigned) best != kNoIdx;
            pr.seed[q] = (unsigned short) (matched ? bslot : 0xFFFFu);
            a[17] += 1.0;
            if (!matched) return;
            // the iteration's sums, as k_icp_stats (wm_icp.hip) forms them
            const double px = qx, py = qy, pz = qz, tx = tg.c(bslot, 0), ty = tg.c(bslot, 1), tz = tg.c(bslot, 2);
            a[0] += 1.0;
            a[1] += px;
            a[2] += py;
            a[3] += pz;
            if (mode == WM_ICP_SVD) {
                a[4] += tx;
                a[5] += ty;
                a[6] += tz;
                a[7] += tx * px;
                a[8] += tx * py;
                a[9] += tx * pz;
                a[10] += ty * px;
                a[11] += ty * py;
                a[12] += ty * pz;
                a[13] += tz * px;
                a[14] += tz * py;
                a[15] += tz * pz;
            } else {
                const double rx = px - tx, ry = py - ty, rz = pz - tz;
                a[4] += py * py + pz * pz;
                a[5] += -px * py;
                a[6] += -px * pz;
                a[7] += px * px + pz * pz;
                a[8] += -py * pz;
                a[9] += px * px + py * py;
                a[10] += rx;
                a[11] += ry;
                a[12] += rz;
                a[13] += py * rz - pz * ry;
                a[14] += pz * rx - px * rz;
                a[15] += px * ry - py * rx;
            }
            a[16] += (double) __uint_as_float((unsigned) (best >> 32));
        };
        // Chunks of 64 consecutive (cell-ordered) queries go to whichever wavefront is free: the first
        // sixteen by wave number, the rest by a ticket in LDS -- static shares left the slowest wave
        // 45 % behind the first.  A chunk's sums are reduced inside its wavefront (recursive halving,
        // wm_wave.hpp) into a row of their own, so which wave took which chunk cannot change a bit of
        // the result; the next chunk's ticket is drawn, and its loads issued, before this chunk's search.
        {
            unsigned c = wave;
            bool has = c < nchunks;
            unsigned q = c * 64u + lane;
            SmQuery cur = {0.f, 0.f, 0.f, 0xFFFFu};
            if (has && q < n_q) cur = sm_fetch(pr, q);
            while (has) {
                unsigned cn = 0;
                if (lane == 0) cn = atomicAdd(&L.ticket, 1u);
                cn = (unsigned) __builtin_amdgcn_readfirstlane((int) cn);
                const bool hn = cn < nchunks;
                const unsigned qn = cn * 64u + lane;
                SmQuery nxt = cur;
                if (hn && qn < n_q) nxt = sm_fetch(pr, qn);
#pragma unroll
                for (int k = 0; k < kAcc; ++k) a[k] = 0.0;
                if (q < n_q) one_query(q, cur);
                acc_halve<kAcc, 32>(a, lane);
                const int comp = acc_comp_of_lane(lane);
                if (comp >= 0) pr.csum[(size_t) c * kAcc + comp] = a[0];
                c = cn, q = qn, cur = nxt, has = hn;
            }
        }
        const unsigned long long t1 = clock64();
        __threadfence_block();
        __syncthreads();
        if (tid < (unsigned) (kSmWaves * kAcc)) {  // the rows added in a fixed order: 16 interleaved partial sums, then those
            const unsigned w = tid / (unsigned) kAcc, k = tid % (unsigned) kAcc;
            double sum = 0;
            for (unsigned c = w; c < nchunks; c += kSmWaves) sum += ((const __attribute__((address_space(1))) double *) pr.csum)[(size_t) c * kAcc + k];
            L.red[w][k] = sum;
        }
        __syncthreads();
        if (tid < (unsigned) kAcc) {
            double sum = 0;
            for (int w = 0; w < kSmWaves; ++w) sum += L.red[w][tid];
            L.sum[tid] = sum;
        }
        __syncthreads();
        const unsigned long long t2 = clock64();
        if (tid == 0) {
            double ex[kStatsLen];
            expand_stats(mode, L.sum, ex);
            icp_apply_stats(&L.st, ex);
            L.ticket = kSmWaves;
        }
        __syncthreads();
        cyc_loop = t1 - t0, cyc_red = t2 - t1, cyc_solve = clock64() - t2;
    }

    // ---- 4. estimateLUMold on the aligned cloud (icp_pcl_functions.cpp:51-179; arithmetic of wm_info.hip)
    const bool with_info = P.with_info != 0;
    if (with_info) {
        float Tf[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Tf[k] = (float) L.st.T[k];
        const float rmax_s = sqrtf(P.thr_d2_strict) * 1.0001f + 1e-6f;
        const bool have_prev = L.st.have_prev != 0;
        double a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = 0.0;
        sm_for_queries(pr, n_q, tid, [&](unsigned q, const SmQuery &c) {
            if (!(c.x == c.x)) return;
            float px, py, pz;
            sm_xform(Tf, c.x, c.y, c.z, px, py, pz);
            unsigned long long best = sm_key(P.thr_d2_strict, kNoIdx);
            unsigned bslot = 0xFFFFu;
            float r = rmax_s;
            const unsigned ps = have_prev ? c.seed : 0xFFFFu;
            if (ps != 0xFFFFu) {
                const float d2b = sm_d2(px, py, pz, tg.c(ps, 0), tg.c(ps, 1), tg.c(ps, 2));
                if (d2b <= P.thr_d2_strict) {
                    best = sm_key(d2b, tg.idx(ps));
                    bslot = ps;
                    r = fmaxf(sqrtf(d2b) * 1.0001f + 1e-6f, 0.05f * g.h);
                }
            } else if (!have_prev) {
                r = P.r0_cells * g.h;
            }
            sm_search(tg, g, px, py, pz, fminf(r, rmax_s), rmax_s, best, bslot);
            const bool matched = (unsigned) best != kNoIdx;
            pr.seed[q] = (unsigned short) (matched ? bslot : 0xFFFFu);
            if (!matched) return;
            const float tx = tg.c(bslot, 0), ty = tg.c(bslot, 1), tz = tg.c(bslot, 2);
            const float av0 = __fmul_rn(0.5f, __fadd_rn(px, tx)), av1 = __fmul_rn(0.5f, __fadd_rn(py, ty)),
                        av2 = __fmul_rn(0.5f, __fadd_rn(pz, tz));
            const float df0 = __fsub_rn(px, tx), df1 = __fsub_rn(py, ty), df2 = __fsub_rn(pz, tz);
            a[0] += 1.0;
            a[1] += av0;
            a[2] += av1;
            a[3] += av2;
            a[4] += __fmul_rn(av0, av2);
            a[5] += __fmul_rn(av0, av1);
            a[6] += __fmul_rn(av1, av2);
            a[7] += __fadd_rn(__fmul_rn(av1, av1), __fmul_rn(av2, av2));
            a[8] += __fadd_rn(__fmul_rn(av0, av0), __fmul_rn(av1, av1));
            a[9] += __fadd_rn(__fmul_rn(av0, av0), __fmul_rn(av2, av2));
            a[10] += df0;
            a[11] += df1;
            a[12] += df2;
            a[13] += __fsub_rn(__fmul_rn(av1, df2), __fmul_rn(av2, df1));
            a[14] += __fsub_rn(__fmul_rn(av0, df1), __fmul_rn(av1, df0));
            a[15] += __fsub_rn(__fmul_rn(av2, df0), __fmul_rn(av0, df2));
        });
        sm_block_sum<16>(a, L, tid);
        double MM[36];
        if (tid == 0) {
            const double *s = L.sum;
#pragma unroll
            for (int k = 0; k < 36; ++k) MM[k] = 0.0;
#define M_(r, c) MM[(r) * 6 + (c)]
            M_(0, 4) = -s[2];
            M_(0, 5) = s[3];
            M_(1, 3) = -s[3];
            M_(1, 4) = s[1];
            M_(2, 3) = s[2];
            M_(2, 5) = -s[1];
            M_(3, 4) = -s[4];
            M_(3, 5) = -s[5];
            M_(4, 5) = -s[6];
            M_(3, 3) = s[7];
            M_(4, 4) = s[8];
            M_(5, 5) = s[9];
            M_(0, 0) = M_(1, 1) = M_(2, 2) = (double) (float) (int) s[0];
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
            double MMinv[36];
            inverse<6>(MM, MMinv);
            for (int r = 0; r < 6; ++r) {
                double d = 0;
                for (int c = 0; c < 6; ++c) d += MMinv[r * 6 + c] * s[10 + c];
                L.D[r] = d;
            }
        }
        __syncthreads();
        double D[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) D[k] = L.D[k];
        double ss[1] = {0.0};
        sm_for_queries(pr, n_q, tid, [&](unsigned, const SmQuery &c) {
            const unsigned bslot = c.seed;
            if (!(c.x == c.x) || bslot == 0xFFFFu) return;
            float px, py, pz;
            sm_xform(Tf, c.x, c.y, c.z, px, py, pz);
            const float tx = tg.c(bslot, 0), ty = tg.c(bslot, 1), tz = tg.c(bslot, 2);
            const float av0 = __fmul_rn(0.5f, __fadd_rn(px, tx)), av1 = __fmul_rn(0.5f, __fadd_rn(py, ty)),
                        av2 = __fmul_rn(0.5f, __fadd_rn(pz, tz));
            const float df0 = __fsub_rn(px, tx), df1 = __fsub_rn(py, ty), df2 = __fsub_rn(pz, tz);
            const double e0 = df0 - (D[0] + av2 * D[5] - av1 * D[4]);
            const double e1 = df1 - (D[1] + av0 * D[4] - av2 * D[3]);
            const double e2 = df2 - (D[2] + av1 * D[3] - av0 * D[5]);
            ss[0] += (double) (float) (e0 * e0 + e1 * e1 + e2 * e2);
        });
        sm_block_sum<1>(ss, L, tid);
        if (tid == 0) {
            const float s2 = (float) L.sum[0];
            const bool bad = (s2 < 0.0000000000001f || !isfinite(s2));
            // estimateLUMold falls through its failure branch (icp_pcl_functions.cpp:170-178)
            const float inv = 1.0f / s2;
            SmallOut &o = out[blockIdx.x];
            for (int k = 0; k < 36; ++k) o.info[k] = MM[k] * inv;
            o.info_degenerate = bad ? 1 : 0;
        }
    }
    if (tid == 0) {
        SmallOut &o = out[blockIdx.x];
        const IcpDevState &s = L.st;
        for (int k = 0; k < 16; ++k) o.T[k] = s.T[k];
        o.mse = s.mse;
        o.prev_mse = s.prev_mse;
        o.iterations = s.iter;
        o.converged = s.converged;
        o.state = s.state;
        o.n_corr = s.n_corr;
        o.n_target_valid = (int) n_tgt_fin;
        o.cell = g.h;
        o.cyc[0] = cyc_loop, o.cyc[1] = cyc_red, o.cyc[2] = cyc_solve, o.cyc[3] = cyc_setup;
        if (!with_info) o.info_degenerate = 0;
    }
}

// ------------------------------------------------------------------ host side
struct SmallBatch {
    DevBuf d_stage;     // [table | clouds | seeds]
    DevBuf d_out;
    DevBuf d_big;       // HBM-resident targets: cell-sorted copies, cell starts, ranks
    void *h_stage = nullptr;  // pinned mirror of table + clouds
    size_t h_stage_cap = 0;
    void *h_out = nullptr;
    size_t h_out_cap = 0;
};

static SmallBatch *small_of(wm_ctx *ctx) {
    if (!ctx->small_batch) ctx->small_batch = new (std::nothrow) SmallBatch();
    return static_cast<SmallBatch *>(ctx->small_batch);
}

void small_batch_release(wm_ctx *ctx) {
    SmallBatch *b = static_cast<SmallBatch *>(ctx->small_batch);
    if (!b) return;
    b->d_stage.release();
    b->d_out.release();
    b->d_big.release();
    if (b->h_stage) (void) hipHostFree(b->h_stage);
    if (b->h_out) (void) hipHostFree(b->h_out);
    delete b;
    ctx->small_batch = nullptr;
}


}  // namespace wm

namespace wm {

// The resident registrations of `n` jobs (none empty) in one or two launches: LDS-resident targets,
// then HBM-resident ones.  Clouds in caller memory (host: staged through pinned memory in 2 MB slices,
// each slice's DMA under the next copy; device: read in place).  res[k] = what the kernel left.
int small_run(wm_ctx *ctx, const SmallJob *jobs, int n, size_t stride, int mem, const wm_icp_params *p, int with_info,
              double info_max_corr, SmallResult *res, float *kernel_ms) {
    if (n <= 0) return WM_OK;
    size_t cloud_bytes = 0, seeds = 0, sorted_pts = 0, chunk_rows = 0, big_bytes = 0;
    int n_lds = 0, n_hbm = 0;
    auto big_need = [](size_t n_target) {  // scratch of one HBM-resident target, 16-byte aligned pieces
        const size_t a = ((n_target + 4) * 12 + 15) & ~(size_t) 15, b = ((n_target + 4) * 2 + 15) & ~(size_t) 15;
        const size_t c = ((size_t) kSmCellsHbm + 8) * 4, e = (n_target * 4 + 15) & ~(size_t) 15;
        return a + b + c + e;
    };
    for (int k = 0; k < n; ++k) {
        const SmallJob &it = jobs[k];
        if (it.n_src == 0 || it.n_tgt == 0 || it.n_tgt > (size_t) kSmMaxTgtHbm || it.n_src > 0x7FFFFFF0u) return WM_ERR_ARG;
        cloud_bytes += ((it.n_src * stride + 15) & ~(size_t) 15) + ((it.n_tgt * stride + 15) & ~(size_t) 15);
        seeds += (it.n_src + 7) & ~(size_t) 7;
        sorted_pts += it.n_src;
        chunk_rows += (it.n_src + 63) / 64;
        if (it.n_tgt > (size_t) kSmMaxTgt) {
            big_bytes += big_need(it.n_tgt);
            ++n_hbm;
        } else {
            ++n_lds;
        }
    }
    SmallBatch *B = small_of(ctx);
    if (!B) return WM_ERR_NOMEM;
    const size_t table_bytes = ((size_t) n * sizeof(SmallPair) + 255) & ~(size_t) 255;
    const size_t up_bytes = table_bytes + (mem == WM_MEM_HOST ? cloud_bytes : 0);
    const size_t seed_bytes = (seeds * sizeof(unsigned short) + 15) & ~(size_t) 15;
    const size_t dev_bytes = table_bytes + (mem == WM_MEM_HOST ? cloud_bytes : 0) + seed_bytes + sorted_pts * sizeof(float4) +
                             chunk_rows * kAcc * sizeof(double);
    WM_HIP(ctx, B->d_stage.reserve(dev_bytes));
    WM_HIP(ctx, B->d_out.reserve((size_t) n * sizeof(SmallOut)));
    if (big_bytes) WM_HIP(ctx, B->d_big.reserve(big_bytes));
    WM_TRY(pinned_reserve(ctx, &B->h_stage, &B->h_stage_cap, up_bytes));
    WM_TRY(pinned_reserve(ctx, &B->h_out, &B->h_out_cap, (size_t) n * sizeof(SmallOut)));
    // the stream may still be reading the staging buffer for the previous batch
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));

    unsigned char *h = static_cast<unsigned char *>(B->h_stage);
    unsigned char *d = B->d_stage.as<unsigned char>();
    SmallPair *table = reinterpret_cast<SmallPair *>(h);
    size_t off = table_bytes;
    unsigned short *seed_base = reinterpret_cast<unsigned short *>(d + table_bytes + (mem == WM_MEM_HOST ? cloud_bytes : 0));
    float4 *sorted_base = reinterpret_cast<float4 *>(reinterpret_cast<unsigned char *>(seed_base) + seed_bytes);
    double *csum_base = reinterpret_cast<double *>(sorted_base + sorted_pts);
    size_t seed_off = 0, sorted_off = 0, csum_off = 0, big_off = 0;
    // rows of the table: the LDS-resident jobs first, then the HBM-resident ones (a launch each)
    std::vector<int> row_of((size_t) n, -1);
    int next_lds = 0, next_hbm = n_lds;
    size_t sent = table_bytes;
    for (int k = 0; k < n; ++k) {
        const SmallJob &it = jobs[k];
        const bool big = it.n_tgt > (size_t) kSmMaxTgt;
        row_of[(size_t) k] = big ? next_hbm++ : next_lds++;
        SmallPair &t = table[row_of[(size_t) k]];
        t.n_src = (unsigned) it.n_src;
        t.n_tgt = (unsigned) it.n_tgt;
        t.prev_mse0 = it.prev_mse0;
        t.presorted = it.presorted ? 1u : 0u;
        t.pad0 = 0;
        if (mem == WM_MEM_HOST) {
            memcpy(h + off, it.src, it.n_src * stride);
            t.src = d + off;
            off += (it.n_src * stride + 15) & ~(size_t) 15;
            memcpy(h + off, it.tgt, it.n_tgt * stride);
            t.tgt = d + off;
            off += (it.n_tgt * stride + 15) & ~(size_t) 15;
            if (off - sent >= ((size_t) 2 << 20)) {
                WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
                sent = off;
            }
        } else {
            t.src = static_cast<const unsigned char *>(it.src);
            t.tgt = static_cast<const unsigned char *>(it.tgt);
        }
        t.seed = seed_base + seed_off;
        seed_off += (it.n_src + 7) & ~(size_t) 7;
        t.sorted = sorted_base + sorted_off;
        sorted_off += it.n_src;
        t.csum = csum_base + csum_off;
        csum_off += ((it.n_src + 63) / 64) * kAcc;
        t.txyz = nullptr, t.tidx = nullptr, t.tcs = nullptr, t.trank = nullptr;
        if (big) {
            unsigned char *b = B->d_big.as<unsigned char>() + big_off;
            t.txyz = reinterpret_cast<float *>(b);
            b += ((it.n_tgt + 4) * 12 + 15) & ~(size_t) 15;
            t.tidx = reinterpret_cast<unsigned short *>(b);
            b += ((it.n_tgt + 4) * 2 + 15) & ~(size_t) 15;
            t.tcs = reinterpret_cast<unsigned *>(b);
            b += ((size_t) kSmCellsHbm + 8) * 4;
            t.trank = reinterpret_cast<unsigned *>(b);
            big_off += big_need(it.n_tgt);
        }
    }
    if (off > sent) WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(ctx, hipMemcpyAsync(d, h, table_bytes, hipMemcpyHostToDevice, ctx->stream));

    double I[16];
    mat4_identity(I);
    IcpDevState st0;
    memset(&st0, 0, sizeof(st0));
    for (int k = 0; k < 16; ++k) st0.T[k] = I[k];
    for (int k = 0; k < 12; ++k) st0.Tf[k] = (float) I[k];
    mat4_identity(st0.Tk);
    st0.prev_mse = DBL_MAX;  // (replaced per job by its prev_mse0)
    st0.forced = p->force_iterations > 0;
    st0.max_iter = st0.forced ? p->force_iterations : p->max_iter;
    st0.mode = p->mode;
    st0.rot_thr = 1.0 - p->t_eps;
    st0.trans_thr = p->t_eps;
    st0.fit_eps = p->fit_eps;
    st0.svd_warm = ctx->tune_fast_solve ? 1 : 0;
    SmallParams P;
    memset(&P, 0, sizeof(P));
    P.stride = (unsigned) stride;
    P.thr_d2 = threshold_d2(p->max_corr);
    // (estimateLUMold gates with the matcher's max_corr, whatever the scale of the last align: icp_pcl_functions.cpp:76-80)
    P.thr_d2_strict = threshold_d2_strict(info_max_corr > 0 ? info_max_corr : p->max_corr);
    P.r0_cells = 0.5f;
    P.with_info = with_info;
    P.iter_cap = st0.max_iter + 1;
    WM_HIP(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    if (n_lds)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_icp_small<true>), dim3((unsigned) n_lds), dim3(kSmThreads), 0, ctx->stream,
                           reinterpret_cast<const SmallPair *>(d), P, st0, B->d_out.as<SmallOut>());
    if (n_hbm)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_icp_small<false>), dim3((unsigned) n_hbm), dim3(kSmThreads), 0, ctx->stream,
                           reinterpret_cast<const SmallPair *>(d) + n_lds, P, st0, B->d_out.as<SmallOut>() + n_lds);
    WM_HIP(ctx, hipGetLastError());
    WM_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    WM_HIP(ctx, hipMemcpyAsync(B->h_out, B->d_out.p, (size_t) n * sizeof(SmallOut), hipMemcpyDeviceToHost, ctx->stream));
    WM_TRY(sync_sleeping(ctx));  // (milliseconds: the registrations of the whole batch)
    float ms = 0;
    (void) hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
    if (kernel_ms) *kernel_ms = ms;
    const SmallOut *o = static_cast<const SmallOut *>(B->h_out);
    for (int k = 0; k < n; ++k) {
        const SmallOut &r = o[row_of[(size_t) k]];
        SmallResult &q = res[k];
        memcpy(q.T, r.T, sizeof(q.T));
        memcpy(q.info, r.info, sizeof(q.info));
        q.mse = r.mse;
        q.prev_mse = r.prev_mse;
        q.iterations = r.iterations;
        q.converged = r.converged;
        q.state = r.state;
        q.n_corr = r.n_corr;
        q.info_degenerate = r.info_degenerate;
        q.cell = r.cell;
        for (int c = 0; c < 4; ++c) q.cyc[c] = r.cyc[c];
    }
    return WM_OK;
}

// what wm_icp_align reports for a finished job
void small_fill_stats(const SmallResult &r, float kernel_ms, wm_icp_stats *s) {
    s->converged = r.converged;
    s->iterations = r.iterations;
    s->state = r.state;
    s->n_corr = r.n_corr;
    s->mse = r.mse;
    s->prev_mse = r.prev_mse;
    s->align_ms = kernel_ms;  // (the whole batch's launch)
    s->nn_levels = 1;
    s->grid_cell = r.cell;
    // (developer: kilocycles of the last iteration's query loop / reduction / solve; set-up in coarse_ms)
    s->nn_ms = (float) r.cyc[0] * 1e-3f, s->stats_ms = (float) r.cyc[1] * 1e-3f, s->solve_ms = (float) r.cyc[2] * 1e-3f;
    s->coarse_ms = (float) r.cyc[3] * 1e-3f;
}

}  // namespace wm

using namespace wm;

extern "C" {

int wm_icp_batch_match(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride, int mem,
                       const wm_icp_params *p, float res, int multiscale_steps, int with_info, double *T_out,
                       double *info_out, wm_icp_stats *stats, int *status) {
    if (!ctx || !p || !status || n_items < 0 || (n_items > 0 && !items) || stride < 12 || (stride & 3)) return WM_ERR_ARG;
    if (!(p->max_corr > 0) || (p->mode != WM_ICP_SVD && p->mode != WM_ICP_GN6)) return WM_ERR_ARG;
    if (p->force_iterations <= 0 && p->max_iter <= 0) return WM_ERR_ARG;
    if (with_info != 0 && with_info != 1) return WM_ERR_ARG;
    if (n_items == 0) return WM_OK;
    for (int k = 0; k < n_items; ++k) {
        const wm_batch_item &it = items[k];
        if ((it.n_src > 0 && !it.src) || (it.n_target > 0 && !it.target) || it.n_src > 0x7FFFFFF0u || it.n_target > 0x7FFFFFF0u)
            return WM_ERR_ARG;
        if (!(res > 0) && it.n_target > (size_t) kSmMaxTgtHbm) return WM_ERR_ARG;
    }
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats) * (size_t) n_items);
    if (res > 0)  // the voxel-filtered branches of match(): wm_batch.hip
        return batch_match_scaled(ctx, items, n_items, stride, mem, p, res, multiscale_steps, with_info, T_out, info_out,
                                  stats, status);
    std::vector<SmallJob> jobs;
    std::vector<int> item_of;
    for (int k = 0; k < n_items; ++k) {
        // PCL: an empty input cloud -> "Not enough correspondences"; match() returns false
        const wm_batch_item &it = items[k];
        status[k] = (it.n_src == 0 || it.n_target == 0) ? (it.n_src == 0 && it.n_target == 0 ? WM_ERR_STATE : WM_TOO_FEW_CORRESPONDENCES) : WM_OK;
        if (status[k] != WM_OK) {
            if (stats) stats[k].state = WM_CONV_NO_CORRESPONDENCES;
            continue;
        }
        jobs.push_back(SmallJob{it.src, it.n_src, it.target, it.n_target, DBL_MAX, 0});  // fresh stopping criteria
        item_of.push_back(k);
    }
    if (jobs.empty()) return WM_OK;
    std::vector<SmallResult> got(jobs.size());
    float ms = 0;
    WM_TRY(small_run(ctx, jobs.data(), (int) jobs.size(), stride, mem, p, with_info, p->max_corr, got.data(), &ms));
    for (size_t j = 0; j < jobs.size(); ++j) {
        const int k = item_of[j];
        const SmallResult &r = got[j];
        if (stats) small_fill_stats(r, ms, &stats[k]);
        if (r.state == WM_CONV_NO_CORRESPONDENCES)
            status[k] = WM_TOO_FEW_CORRESPONDENCES;
        else if (!r.converged)
            status[k] = WM_NOT_CONVERGED;
        else if (T_out)
            memcpy(T_out + 16 * (size_t) k, r.T, sizeof(r.T));
        // estimateLUMold runs whatever match() returned (icp.cpp:135-142 has no hasConverged() guard on it)
        if (with_info && info_out) memcpy(info_out + 36 * (size_t) k, r.info, sizeof(r.info));
    }
    return WM_OK;
}

}  // extern "C"
