// wm_gicp_small.hip -- GICPMatcher::match() for MANY queued pairs in one launch: one registration per
// workgroup, the whole of pcl::GeneralizedIterativeClosestPoint::align inside it (reference behaviour:
// wave_matching/src/gicp.cpp:31-34 setters, :58-60 align / hasConverged / getFinalTransformation; the pool that
// queues such pairs: wave_matching/include/wave/matching/multi_matcher.hpp:29-34).
//
// Why: a registration of two 20 000-point clouds is ~170 evaluations of the objective, each a dependent step of
// the optimiser.  On the whole device (wm_gicp.hip) every one of them is a trip across PCIe or a pair of kernel
// launches for a few microseconds of work, and a pool of such registrations is bound by the four hardware queues
// (~2 000 pairs/s).  Here a pair owns ONE compute unit for its whole life: 1024 threads build both clouds' grids
// (counting sort, the histogram in LDS), compute the covariances (the same k-NN search and SVD as k_gicp_cov),
// and run the outer loop -- 1-NN search, Mahalanobis matrices, BFGS -- with the optimiser's scalar code executed
// by every thread (uniformly) and the objective as a workgroup-wide double-double sum.  256 pairs run at once.
//
// What is the same as the one-pair path: the neighbours ((d2, index) order), the covariances and Mahalanobis
// matrices (same device functions), the terms of the objective and their sums (double-double: the value does not
// depend on which thread added what), the optimiser's code (wm_bfgs.hpp, compiled for both sides), and the float
// sinf / cosf / atan2f / asinf of PCL's float-quantised transform (glibc's algorithms restated in wm_bfgs.hpp: the
// device library's differ from glibc's in the last bit for ~1 % of arguments, which once moved a stopping point by
// 1.5 mm).  With that every tested pair comes out EQUAL to the one-pair path and the oracle -- transform, objective,
// iteration and evaluation counts (tests/test_gicp_batch_gpu.py asserts np.array_equal).  The GUARANTEE
// (include/wavematch.h) stays 1e-6 m / 1e-6 rad: the double sin / cos of the gradient's rotation are the device
// library's; a last-bit difference there showed once, after 300 evaluations of a registration that does not converge.
#include "wm_internal.hpp"
#include "wm_gicp_dev.hpp"
#include "wm_bfgs.hpp"
#include "wm_gicp_quad.hpp"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

namespace wm {

constexpr int kGsThreads = 1024;
constexpr int kGsWaves = kGsThreads / 64;
// An evaluation's data: per source slot the point (float4), its match (float4; x = NaN: none) and the pair's
// Mahalanobis matrix (9 doubles) -- 104 bytes, read ~170 times per registration, from HBM every time (256
// registrations at once: ~0.5 GB per round of evaluations).  Laid out in blocks of 64 slots = one wave's trip:
// [64 x point | 64 x match | 9 x 64 x one matrix entry], 6656 contiguous bytes, every load of the wave a full
// line.  (As eleven separate arrays per registration -- 2800 streams over the device -- the same bytes took 5 % longer
// with 256 registrations running, 10 % with 16.)
constexpr unsigned kGsEvBlock = 64u * (16u + 16u + 72u);
__device__ __forceinline__ float4 *gs_ev_point(unsigned char *evb, unsigned i) {
    return reinterpret_cast<float4 *>(evb + (size_t) (i >> 6) * kGsEvBlock) + (i & 63u);
}
__device__ __forceinline__ float4 *gs_ev_match(unsigned char *evb, unsigned i) {
    return reinterpret_cast<float4 *>(evb + (size_t) (i >> 6) * kGsEvBlock + 1024u) + (i & 63u);
}
__device__ __forceinline__ double *gs_ev_mahal(unsigned char *evb, unsigned i) {  // entry c at [c * 64]
    return reinterpret_cast<double *>(evb + (size_t) (i >> 6) * kGsEvBlock + 2048u) + (i & 63u);
}
constexpr unsigned kGsCells = 262144;  // cells of a cloud's grid at most (two arrays of that many words per cloud in the pair's scratch)

struct GsPair {  // one registration of the batch (device table)
    const unsigned char *src, *tgt;  // caller-layout points in device memory
    unsigned n_src, n_tgt;
    // scratch in HBM (this pair's own)
    float4 *s_pts, *t_pts;  // the finite points in cell order, .w = the caller's index
    unsigned *s_cs, *t_cs;  // first slot of every cell (+ end)
    unsigned *run;          // the counting sort's histogram / running offsets
    double *c1, *c2;        // covariances, 9 per point, by the caller's index
    unsigned char *evb;     // what an evaluation reads, in blocks of 64 source slots (kGsEvBlock bytes each, see gs_ev_*)
    unsigned *match;        // by the source's slot: the caller's index of its target match (kNoIdx none)
};

struct GsParams {
    unsigned stride;  // bytes between points
    int k;            // corr_rand
    double eps;       // gicp_epsilon
    float thr_d2;     // d2 <= thr_d2 <=> d2 < max_corr^2
    int max_iter, max_inner, forced;
    double r_eps, t_eps;
    int debug;  // developer (WM_GICP_SMALL_TRACE): pair 0 prints every evaluation, as WM_GICP_TRACE does on the one-pair path
    int objective;  // wm_gicp_params::objective (WM_GICP_OBJECTIVE_STATISTICS: no evaluation blocks, 74 sums per outer iteration)
};

struct GsOut {
    double T[16];
    double f;
    int converged, iterations, n_corr, inner_total, evaluations, status;
    int n_src_valid, n_tgt_valid;
    unsigned long long cyc[4];  // developer: shader-clock cycles of the grids / covariances / searches / minimisations
};

struct GsShared {
    double red[kGsWaves][kGicpAcc][2];
    double sum[16];
    float boxf[kGsWaves][8];
    unsigned wcnt[kGsWaves];
    unsigned scan[kGsWaves];
    unsigned cmd;      // 1 / 2: evaluate at cmdT (2: walking the blocks backwards), 0: the minimisation is over
    float cmdT[12];
    double res[8];     // what wave 0's minimisation left: x (6), f, inner iterations
    int res_evals;
    // the registration's state between the phases (each phase is a function of its own -- not inlined, so that
    // what one phase keeps in registers is not spilled inside another's loops: as ONE function the kernel kept
    // 230 registers in scratch memory and every trip of the search loop went there)
    GsPair pr;
    GsParams P;
    GridDev gs, gt;
    unsigned n_s, n_t;
    float T[16];
    unsigned cnt;  // matched pairs of the last search
    int evals;
    double Q[kQuadN];  // the statistics objective: the 74 sums of the last search's pairs (wm_gicp_quad.hpp)
};

// The statistics objective inside the workgroup.  A trip of gs_correspondences gives every thread one pair; what the
// pair's 74 terms are made of -- the symmetric part of its Mahalanobis matrix, the source point, PCL's float residual
// at the pairing transform -- goes into LDS (structure of arrays: 72 bytes per pair), and after a barrier WAVE w adds
// the terms of components w, w + 16, ... (five at most) of all 1024 pairs, sixteen pairs per lane, in double-double:
// a component is wave-uniform, so its formula is scalar control flow, and a thread keeps five (hi, lo) pairs instead
// of seventy-four.  Same terms, same double-double sums as k_gicp_quad (wm_gicp.hip): the same 74 numbers, bit for bit.
struct GsQuadLds {
    double Ms[6][kGsThreads];
    float z[3][kGsThreads];
    float r0[3][kGsThreads];
    unsigned ok[kGsThreads];
};
constexpr int kGsQuadPerWave = (kQuadN + kGsWaves - 1) / kGsWaves;  // 5
// component c's term of the pair in slot e (the expressions of gicp_quad_terms, operation for operation)
__device__ __forceinline__ double gs_quad_term(const GsQuadLds &G, int c, unsigned e) {
    if (c < kQuadOffB) {
        const int sidx = c / 10, t = c - sidx * 10;
        // t -> (j, k), j <= k, in quad_s10's order
        const int j = t < 4 ? 0 : (t < 7 ? 1 : (t < 9 ? 2 : 3));
        const int k = t < 4 ? t : (t < 7 ? t - 3 : (t < 9 ? t - 5 : 3));
        const double zj = j < 3 ? (double) G.z[j][e] : 1.0, zk = k < 3 ? (double) G.z[k][e] : 1.0;
        return G.Ms[sidx][e] * (zj * zk);
    }
    const double r0[3] = {(double) G.r0[0][e], (double) G.r0[1][e], (double) G.r0[2][e]};
    if (c < kQuadOffC) {
        const int a = (c - kQuadOffB) >> 2, j = (c - kQuadOffB) & 3;
        const double t0 = (G.Ms[quad_s6(a, 0)][e] * r0[0] + G.Ms[quad_s6(a, 1)][e] * r0[1]) + G.Ms[quad_s6(a, 2)][e] * r0[2];
        return t0 * (j < 3 ? (double) G.z[j][e] : 1.0);
    }
    if (c == kQuadOffC) {
        double t0[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            t0[a] = (G.Ms[quad_s6(a, 0)][e] * r0[0] + G.Ms[quad_s6(a, 1)][e] * r0[1]) + G.Ms[quad_s6(a, 2)][e] * r0[2];
        return (r0[0] * t0[0] + r0[1] * t0[1]) + r0[2] * t0[2];
    }
    return 1.0;
}


__device__ __forceinline__ bool gs_load(const unsigned char *base, unsigned i, unsigned stride, float &x, float &y, float &z) {
    const float *p = reinterpret_cast<const float *>(base + (size_t) i * stride);
    x = p[0], y = p[1], z = p[2];
    return x - x == 0.f && y - y == 0.f && z - z == 0.f;  // finite
}

__device__ __forceinline__ unsigned gs_cell(const GridDev &g, float x, float y, float z) {
    int cx = (int) floorf((x - g.ox) * g.inv_h), cy = (int) floorf((y - g.oy) * g.inv_h), cz = (int) floorf((z - g.oz) * g.inv_h);
    cx = min(max(cx, 0), g.nx - 1);
    cy = min(max(cy, 0), g.ny - 1);
    cz = min(max(cz, 0), g.nz - 1);
    return (unsigned) ((cz * g.ny + cy) * g.nx + cx);
}

// One cloud's grid: bounding box of its finite points, cell edge, counting sort (histogram and running offsets in
// the pair's scratch: L2-resident; `cs` = first slot of every cell + end, `run` = scratch of the same size).
// The cell edge: first what the one-pair path takes for the covariance search (choose_cell: 1.5 x the mean spacing
// of n points in the box's VOLUME); a scan is points on surfaces, though, most of the box is empty and the occupied
// cells then hold a dozen points each -- every search would wade through hundreds of candidates.  So the occupied
// cells are counted and, above kGsOccupancy points per occupied cell, the edge is shrunk by sqrt(ratio) (surfaces:
// occupancy goes with the edge squared) and the histogram redone, as long as the cells fit kGsCells.  (Which cell
// edge the grid has changes what a search costs, never what it finds.)
// Returns the number of finite points (uniform); every thread calls.
constexpr float kGsOccupancy = 3.0f;
__device__ __attribute__((noinline)) unsigned gs_build_grid(const unsigned char *raw, unsigned n, unsigned stride, float4 *pts, unsigned *cs, unsigned *run,
                                  GridDev &g, GsShared &S) {
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    unsigned cnt = 0;
    for (unsigned i = tid; i < n; i += kGsThreads) {
        float x, y, z;
        if (gs_load(raw, i, stride, x, y, z)) {
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
    for (int w = 0; w < kGsWaves; ++w) {
#pragma unroll
        for (int d = 0; d < 3; ++d) lo[d] = fminf(lo[d], S.boxf[w][d]), hi[d] = fmaxf(hi[d], S.boxf[w][3 + d]);
        cnt += S.wcnt[w];
    }
    __syncthreads();
    const unsigned n_valid = cnt;
    g.pts = pts;
    g.cell_start = cs;
    if (n_valid == 0) {
        g.ox = g.oy = g.oz = 0.f;
        g.h = g.inv_h = 1.f;
        g.slack = 1e-3f;
        g.nx = g.ny = g.nz = 1;
        if (tid < 2) cs[tid] = 0u;
        __syncthreads();
        return 0;
    }
    // (choose_cell / grid_dims / build_grid_level of the one-pair path, wm_gicp.hip / wm_grid.hip)
    double vol = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) vol *= fmax((double) hi[d] - (double) lo[d], 1e-3);
    float h = (float) fmax(cbrt(vol / (double) n_valid) * 1.5, 1e-4);
    const float extent = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    float amax = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) amax = fmaxf(amax, fmaxf(fabsf(lo[d]), fabsf(hi[d])));
    const float ulp = fmaxf(amax, extent) * 1.2e-7f;
    g.ox = lo[0], g.oy = lo[1], g.oz = lo[2];
    unsigned cells = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int nx, ny, nz;
        for (;;) {
            nx = (int) floorf((hi[0] - lo[0]) / h) + 1, ny = (int) floorf((hi[1] - lo[1]) / h) + 1, nz = (int) floorf((hi[2] - lo[2]) / h) + 1;
            if ((unsigned long long) nx * (unsigned long long) ny * (unsigned long long) nz <= (unsigned long long) kGsCells) break;
            h *= 1.1f;
        }
        g.h = h;
        g.inv_h = 1.0f / h;
        g.slack = fmaxf(1e-3f, 8.0f * ulp / h);
        g.nx = nx, g.ny = ny, g.nz = nz;
        cells = (unsigned) (nx * ny * nz);
        for (unsigned c = tid; c < cells; c += kGsThreads) run[c] = 0u;
        __syncthreads();
        for (unsigned i = tid; i < n; i += kGsThreads) {
            float x, y, z;
            if (gs_load(raw, i, stride, x, y, z)) atomicAdd(&run[gs_cell(g, x, y, z)], 1u);
        }
        __syncthreads();
        if (attempt == 1) break;
        unsigned occ = 0;
        // (the counts were made by atomics, which live in L2: read them there, past whatever this CU's L1 holds of the array)
        for (unsigned c = tid; c < cells; c += kGsThreads) occ += __hip_atomic_load(&run[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) occ += __shfl_xor(occ, m);
        if (lane == 0) S.wcnt[wave] = occ;
        __syncthreads();
        occ = 0;
        for (int w = 0; w < kGsWaves; ++w) occ += S.wcnt[w];
        __syncthreads();
        const float per_cell = (float) n_valid / (float) max(occ, 1u);
        if (!(per_cell > 1.25f * kGsOccupancy)) break;
        // (not below what kGsCells cells can cover: the loop above would only widen it again)
        const float h_floor = (float) cbrt(vol / (double) kGsCells) * 1.05f;
        const float h_new = fmaxf(h * sqrtf(kGsOccupancy / per_cell), h_floor);
        if (!(h_new < 0.9f * h)) break;
        h = h_new;
    }
    // exclusive scan of the counts, 4096 cells (four per thread) at a time: run[c] and cs[c] = the cell's first slot
    unsigned carry = 0;
    for (unsigned c0 = 0; c0 < cells; c0 += 4u * kGsThreads) {
        const unsigned c = c0 + 4u * tid;
        unsigned v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = c + (unsigned) u < cells ? __hip_atomic_load(&run[c + (unsigned) u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
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
        for (unsigned w = 0; w < (unsigned) kGsWaves; ++w) {
            if (w < wave) base += S.scan[w];
            all += S.scan[w];
        }
        unsigned at = base + incl - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c + (unsigned) u < cells) {
                run[c + (unsigned) u] = at;
                cs[c + (unsigned) u] = at;
                at += v[u];
            }
        carry += all;
        __syncthreads();
    }
    if (tid == 0) cs[cells] = n_valid;
    __syncthreads();
    for (unsigned i = tid; i < n; i += kGsThreads) {
        float x, y, z;
        if (gs_load(raw, i, stride, x, y, z)) {
            const unsigned pos = atomicAdd(&run[gs_cell(g, x, y, z)], 1u);
            pts[pos] = make_float4(x, y, z, __uint_as_float(i));
        }
    }
    __syncthreads();
    return n_valid;
}

// The objective of one minimisation, as wm_bfgs.hpp asks for it.  The optimiser is scalar code on six unknowns:
// WAVE 0 runs it (sixteen waves running it side by side, four to a SIMD, took longer than the sums it asks for); the
// other fifteen sit in serve(), take the transform of each evaluation from LDS, add their share of the pairs and go
// back to waiting.  Two barriers per evaluation (transform posted / rows in), one more when wave 0 says it is done.
struct GsFn {
    const GsPair *pr;
    GsShared *S;
    unsigned n;  // the source's finite points
    int m;       // matched pairs
    int evals;
    int debug;
    __device__ int pairs() const { return m; }
    __device__ bool failed() const { return false; }
    __device__ bool test_at_start() const { return false; }
    // this wave's share of one evaluation -> its row of S->red
    __device__ void share(const FdfArgs &A, bool backwards) {
        double hi[kGicpAcc], lo[kGicpAcc];
#pragma unroll
        for (int k = 0; k < kGicpAcc; ++k) hi[k] = lo[k] = 0.0;
        const unsigned tid = threadIdx.x;
        unsigned char *evb = pr->evb;
        // Every other evaluation walks the blocks from the far end: what the last one read last -- about half of a
        // 20 000-point registration's 2 MB, with 256 of them sharing the 256 MB memory-side cache -- is then read
        // first, before the rest of the sweep has pushed it out.  (The sums are double-double: any order, same bits.)
        const unsigned trips = n > tid ? (n - tid + kGsThreads - 1u) / kGsThreads : 0u;
        for (unsigned t = 0; t < trips; ++t) {
            const unsigned i = tid + (backwards ? trips - 1u - t : t) * kGsThreads;
            const float4 p = *gs_ev_point(evb, i), q = *gs_ev_match(evb, i);
            const double *m = gs_ev_mahal(evb, i);
            double M[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) M[c] = m[c * 64];
            if (!(q.x == q.x)) continue;  // (no match)
            gicp_fdf_point(hi, lo, A, p.x, p.y, p.z, q.x, q.y, q.z, M);
        }
        const unsigned lane = tid & 63u, wave = tid >> 6;
        dd_halve<kGicpAcc, 32>(hi, lo, lane);
        const int comp = dd_comp_of_lane(lane);
        if (comp >= 0) {
            S->red[wave][comp][0] = hi[0];
            S->red[wave][comp][1] = lo[0];
        }
    }
    // waves 1 .. 15, for the length of one minimisation
    __device__ void serve() {
        for (;;) {
            __syncthreads();  // (a transform, or the end, has been posted)
            if (S->cmd == 0u) return;
            FdfArgs A;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                A.T[k] = S->cmdT[k];
                A.B[k] = (k % 5 == 0) ? 1.f : 0.f;
            }
            share(A, S->cmd == 2u);
            __syncthreads();  // (the rows are in)
        }
    }
    __device__ void done() {  // wave 0, after the minimisation
        if ((threadIdx.x & 63u) == 0) S->cmd = 0u;
        __syncthreads();
    }
    // wave 0
    __device__ double fdf(const double x[6], double g[6]) {
        double I[16];
        mat4_identity(I);
        float T[16];
        state_to_matrix_f(I, x, T);
        FdfArgs A;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            A.T[k] = T[k];
            A.B[k] = (float) I[k];
        }
        const unsigned lane = threadIdx.x & 63u;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) S->cmdT[k] = T[k];
            S->cmd = 1u + ((unsigned) evals & 1u);
        }
        __syncthreads();
        share(A, (evals & 1) != 0);
        __syncthreads();
        double v = 0;
        if (lane < (unsigned) kGicpAcc) {
            double h = 0, l = 0;
            for (int w = 0; w < kGsWaves; ++w) {
                dd_add(h, l, S->red[w][lane][0]);
                l += S->red[w][lane][1];
            }
            v = h + l;
        }
        double a[kGicpAcc];
#pragma unroll
        for (int k = 0; k < kGicpAcc; ++k) a[k] = __shfl(v, k);
        ++evals;
        const double mm = (double) m;
        if (g) {
            double Racc[9];
            for (int k = 0; k < 3; ++k) g[k] = a[1 + k] * 2.0 / mm;
            for (int k = 0; k < 9; ++k) Racc[k] = a[4 + k] * 2.0 / mm;
            r_derivative(x, Racc, g);
        }
        if ((debug & 1) && blockIdx.x == 0 && lane == 0)
            printf("%d %.17g %.17g %.17g %.17g %.17g %.17g | %.17g | %.17g %.17g %.17g %.17g %.17g %.17g\n", m, x[0], x[1], x[2], x[3], x[4], x[5], a[0] / mm,
                   g ? g[0] : 0.0, g ? g[1] : 0.0, g ? g[2] : 0.0, g ? g[3] : 0.0, g ? g[4] : 0.0, g ? g[5] : 0.0);
        return a[0] / mm;
    }
};

// computeCovariances of both clouds
template <int K>
__device__ __attribute__((noinline)) void gs_covariances(GsShared &S, uint2 *runs) {
    const unsigned tid = threadIdx.x;
    const int k = S.P.k;
    const double eps = S.P.eps;
    const unsigned stride = S.P.stride;
    for (int which = 0; which < 2; ++which) {
        const GridDev g = which ? S.gs : S.gt;
        const unsigned n = which ? S.n_s : S.n_t;
        const unsigned char *raw = which ? S.pr.src : S.pr.tgt;
        double *cov = which ? S.pr.c1 : S.pr.c2;
        for (unsigned i = tid; i < n; i += kGsThreads) {
            const float4 q = g.pts[i];
            unsigned long long best[K];
            knn_search<K>(g, q.x, q.y, q.z, k, k <= 12 ? 1.0f : 1.5f, best, runs, tid, kGsThreads);
            gicp_cov_of_list<K>(best, k, eps, [&](unsigned idx) {
                float x, y, z;
                (void) gs_load(raw, idx, stride, x, y, z);
                return make_float4(x, y, z, 0.f);
            }, cov + (size_t) __float_as_uint(q.w) * 9);
        }
    }
    __syncthreads();
}

// one outer iteration's correspondences: 1-NN of every transformed source point, d2 < max_corr^2, and the
// Mahalanobis matrix of every pair; S.cnt = the number of pairs
// one trip's pairs -> this wave's components (a function of its own: its registers are not the search loop's)
__device__ __attribute__((noinline)) void gs_quad_accumulate(const GsQuadLds &G, unsigned wave, unsigned lane, double *qh, double *ql) {
#pragma unroll
    for (int u = 0; u < kGsQuadPerWave; ++u) {
        const int c = (int) wave + u * kGsWaves;  // (wave-uniform)
        if (c < kQuadN) {
            double h = qh[u], l = ql[u];
            for (unsigned e = lane; e < (unsigned) kGsThreads; e += 64u)
                if (G.ok[e]) ddn_add(h, l, gs_quad_term(G, c, e));
            qh[u] = h;
            ql[u] = l;
        }
    }
}

template <bool STATS>
__device__ __attribute__((noinline)) void gs_correspondences(GsShared &S, uint2 *runs, GsQuadLds *G) {
    const unsigned tid = threadIdx.x;
    const GridDev gt = S.gt;
    const unsigned n_s = S.n_s, n_t = S.n_t, stride = S.P.stride;
    const float thr_d2 = S.P.thr_d2;
    const int debug = S.P.debug;
    const float r_stop = sqrtf(thr_d2) * 1.0001f + 1e-6f;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = S.T[i];
    const float4 *s_pts = S.pr.s_pts;
    unsigned char *evb = S.pr.evb;
    unsigned *match = S.pr.match;
    const double *c1 = S.pr.c1, *c2 = S.pr.c2;
    const unsigned char *tgt = S.pr.tgt;
    unsigned mine = 0;
    // (statistics: this wave's components of the 74 sums, sixteen pairs per lane and trip)
    const unsigned lane = tid & 63u, wave = tid >> 6;
    double qh[kGsQuadPerWave], ql[kGsQuadPerWave];
#pragma unroll
    for (int u = 0; u < kGsQuadPerWave; ++u) qh[u] = ql[u] = 0.0;
    for (unsigned i0 = 0; i0 < n_s; i0 += kGsThreads) {
        const unsigned i = i0 + tid;
        if constexpr (STATS) G->ok[tid] = 0u;
        if (i < n_s) {
        const float4 p = s_pts[i];
        // PCL's float transform of a source point: ((m00*x + m01*y) + m02*z) + m03
        const float qx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], p.x), __fmul_rn(T[1], p.y)), __fmul_rn(T[2], p.z)), T[3]);
        const float qy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], p.x), __fmul_rn(T[5], p.y)), __fmul_rn(T[6], p.z)), T[7]);
        const float qz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], p.x), __fmul_rn(T[9], p.y)), __fmul_rn(T[10], p.z)), T[11]);
        unsigned long long best[1];
        if (debug & 4) best[0] = i < n_t ? (unsigned long long) __float_as_uint(gt.pts[i].w) : ~0ull;  // (developer timing experiment: no search)
        else knn_search<1>(gt, qx, qy, qz, 1, 0.5f, best, runs, tid, kGsThreads, r_stop);
        unsigned j = kNoIdx;
        if (best[0] != ~0ull && __uint_as_float((unsigned) (best[0] >> 32)) <= thr_d2) j = (unsigned) best[0];
        match[i] = j;
        if (j != kNoIdx) {
            float x, y, z;
            (void) gs_load(tgt, j, stride, x, y, z);
            double R[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) R[a * 3 + b] = (double) T[a * 4 + b];
            double o[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (!(debug & 2)) gicp_mahal_of(c1 + (size_t) __float_as_uint(p.w) * 9, c2 + (size_t) j * 9, R, o);
            if constexpr (STATS) {
                // what the pair's 74 terms are made of (gicp_quad_terms, wm_gicp_quad.hpp): the symmetric part of M,
                // the point, PCL's float residual under the pairing transform
                G->Ms[0][tid] = o[0];
                G->Ms[1][tid] = 0.5 * (o[1] + o[3]);
                G->Ms[2][tid] = 0.5 * (o[2] + o[6]);
                G->Ms[3][tid] = o[4];
                G->Ms[4][tid] = 0.5 * (o[5] + o[7]);
                G->Ms[5][tid] = o[8];
                G->z[0][tid] = p.x, G->z[1][tid] = p.y, G->z[2][tid] = p.z;
                G->r0[0][tid] = __fsub_rn(qx, x), G->r0[1][tid] = __fsub_rn(qy, y), G->r0[2][tid] = __fsub_rn(qz, z);
                G->ok[tid] = 1u;
            } else {
                *gs_ev_point(evb, i) = p;
                *gs_ev_match(evb, i) = make_float4(x, y, z, 0.f);
#pragma unroll
                for (int a = 0; a < 9; ++a) gs_ev_mahal(evb, i)[a * 64] = o[a];
            }
            ++mine;
        } else if constexpr (!STATS) {
            *gs_ev_match(evb, i) = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
        }
        }  // (i < n_s)
        if constexpr (STATS) {
            __syncthreads();  // (the trip's pairs are in LDS)
            gs_quad_accumulate(*G, wave, lane, qh, ql);
            __syncthreads();  // (... and read, before the next trip overwrites them)
        }
    }
    if constexpr (STATS) {
        // the lanes' double-double sums of a component -> one (hi, lo) -> rounded once (any order: the same value)
#pragma unroll
        for (int u = 0; u < kGsQuadPerWave; ++u) {
            const int c = (int) wave + u * kGsWaves;
            double h = qh[u], l = ql[u];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                const double oh = __shfl_xor(h, m), ol = __shfl_xor(l, m);
                ddn_add(h, l, oh);
                l += ol;
            }
            if (lane == 0 && c < kQuadN) S.Q[c] = h + l;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) mine += __shfl_xor(mine, m);
    if ((tid & 63u) == 0) S.wcnt[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        unsigned cnt = 0;
        for (int w = 0; w < kGsWaves; ++w) cnt += S.wcnt[w];
        S.cnt = cnt;
    }
    __syncthreads();
}

// one outer iteration's minimisation (estimateRigidTransformationBFGS), from S.T: S.res = x, f, inner iterations
__device__ __attribute__((noinline)) void gs_minimise(GsShared &S) {
    const unsigned tid = threadIdx.x;
    GsFn F;
    F.pr = &S.pr;
    F.S = &S;
    F.n = S.n_s;
    F.m = (int) S.cnt;
    F.evals = 0;
    F.debug = S.P.debug;
    if (tid < 64u) {
        // (float arguments: PCL's atan2 / asin on Matrix4f entries are the float functions)
        double x[6] = {(double) S.T[3], (double) S.T[7], (double) S.T[11], (double) libm_atan2f(S.T[9], S.T[10]),
                       (double) libm_asinf(-S.T[8]), (double) libm_atan2f(S.T[4], S.T[0])};
        double fl = 0;
        const int in0 = bfgs_minimize(F, x, S.P.max_inner, &fl);
        F.done();
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) S.res[k] = x[k];
            S.res[6] = fl;
            S.res[7] = (double) in0;
            S.evals += F.evals;
        }
    } else {
        F.serve();
    }
    __syncthreads();
}

// ... with the statistics objective: the 74 sums are in S.Q (gs_correspondences<true>), wave 0 runs the optimiser
// and every evaluation it asks for is scalar work (gicp_quad_eval); the other waves have nothing to serve
struct GsQuadFn {
    const GsShared *S;  // (the 74 sums are read where they are: in LDS)
    float T0[12];
    double base[16];
    int m, evals, debug;
    __device__ int pairs() const { return m; }
    __device__ bool failed() const { return false; }
    __device__ bool test_at_start() const { return true; }
    __device__ double fdf(const double x[6], double g[6]) {
        ++evals;
        const double f = gicp_quad_eval(S->Q, T0, base, x, g);
        if ((debug & 1) && blockIdx.x == 0 && (threadIdx.x & 63u) == 0)
            printf("%d %.17g %.17g %.17g %.17g %.17g %.17g | %.17g | %.17g %.17g %.17g %.17g %.17g %.17g\n", m, x[0], x[1], x[2], x[3], x[4], x[5], f,
                   g ? g[0] : 0.0, g ? g[1] : 0.0, g ? g[2] : 0.0, g ? g[3] : 0.0, g ? g[4] : 0.0, g ? g[5] : 0.0);
        return f;
    }
};
__device__ __attribute__((noinline)) void gs_minimise_stats(GsShared &S) {
    const unsigned tid = threadIdx.x;
    if (tid < 64u) {
        GsQuadFn F;
        F.S = &S;
        for (int k = 0; k < 12; ++k) F.T0[k] = S.T[k];
        mat4_identity(F.base);
        F.m = (int) S.cnt;
        F.evals = 0;
        F.debug = S.P.debug;
        double x[6] = {(double) S.T[3], (double) S.T[7], (double) S.T[11], (double) libm_atan2f(S.T[9], S.T[10]),
                       (double) libm_asinf(-S.T[8]), (double) libm_atan2f(S.T[4], S.T[0])};
        double fl = 0;
        const int in0 = bfgs_minimize(F, x, S.P.max_inner, &fl);
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) S.res[k] = x[k];
            S.res[6] = fl;
            S.res[7] = (double) in0;
            S.evals += F.evals;
        }
    }
    __syncthreads();
}

template <int K>
__global__ void __launch_bounds__(kGsThreads) k_gicp_small(const GsPair *__restrict__ table, GsParams P, GsOut *__restrict__ outs) {
    __shared__ uint2 runs[kKnnRows * kGsThreads];  // the k-NN search's run lists (64 KB)
    __shared__ GsQuadLds quad;                     // the statistics objective's per-trip pair data (76 KB)
    __shared__ GsShared S;
    GsOut &out = outs[blockIdx.x];
    const unsigned tid = threadIdx.x;
    if (tid == 0) {
        S.pr = table[blockIdx.x];
        S.P = P;
        S.evals = 0;
        S.cnt = 0;
    }
    __syncthreads();
    unsigned long long t_mark = clock64();
    {
        GridDev g;
        const unsigned n_t = gs_build_grid(S.pr.tgt, S.pr.n_tgt, P.stride, S.pr.t_pts, S.pr.t_cs, S.pr.run, g, S);
        if (tid == 0) S.gt = g, S.n_t = n_t;
        const unsigned n_s = gs_build_grid(S.pr.src, S.pr.n_src, P.stride, S.pr.s_pts, S.pr.s_cs, S.pr.run, g, S);
        if (tid == 0) S.gs = g, S.n_s = n_s;
        __syncthreads();
    }
    if (tid == 0) {
        out.n_src_valid = (int) S.n_s, out.n_tgt_valid = (int) S.n_t;
        out.converged = 0, out.iterations = 0, out.n_corr = 0, out.inner_total = 0, out.evaluations = 0;
        out.f = 0;
        out.cyc[0] = clock64() - t_mark;
        out.cyc[1] = out.cyc[2] = out.cyc[3] = 0;
    }
    // PCL: "Number of points in cloud is less than k_correspondences_" -> no alignment
    if ((unsigned) P.k > S.n_s || (unsigned) P.k > S.n_t) {
        if (tid == 0) out.status = WM_NOT_CONVERGED;
        return;
    }
    t_mark = clock64();
    gs_covariances<K>(S, runs);
    if (tid == 0) out.cyc[1] = clock64() - t_mark;

    // the outer loop of align (computeTransformation)
    float prevT[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) prevT[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (tid < 16) S.T[tid] = (tid % 5 == 0) ? 1.f : 0.f;
    __syncthreads();
    const int max_it = P.forced > 0 ? P.forced : P.max_iter;
    const bool statistics = P.objective != WM_GICP_OBJECTIVE_PCL_SUMS;
    int iter = 0, inner_total = 0, status = WM_NOT_CONVERGED;
    bool converged = false;
    double f_last = 0;
    unsigned cnt = 0;
    unsigned long long cyc_search = 0, cyc_min = 0;
    while (!converged) {
        t_mark = clock64();
        if (statistics) gs_correspondences<true>(S, runs, &quad);
        else gs_correspondences<false>(S, runs, nullptr);
        cnt = S.cnt;
        cyc_search += clock64() - t_mark;
        t_mark = clock64();
#pragma unroll
        for (int i = 0; i < 16; ++i) prevT[i] = S.T[i];
        if (statistics) gs_minimise_stats(S);
        else gs_minimise(S);
        cyc_min += clock64() - t_mark;
        const int inner = (int) S.res[7];
        if (inner < 0) break;  // NotEnoughPointsException: the loop breaks, converged_ stays false
        f_last = S.res[6];
        inner_total += inner;
        double x[6], I[16];
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = S.res[k];
        mat4_identity(I);
        float T[16];
        state_to_matrix_f(I, x, T);
        __syncthreads();  // (everybody has read S.res and S.T)
        if (tid < 16) S.T[tid] = T[tid];
        __syncthreads();
        double delta = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double ratio = (a < 3 && b < 3) ? 1.0 / P.r_eps : 1.0 / P.t_eps;
                const double cd = ratio * fabs((double) prevT[a * 4 + b] - (double) T[a * 4 + b]);
                if (cd > delta) delta = cd;
            }
        ++iter;
        if (P.forced > 0 ? (iter >= max_it) : (iter >= max_it || delta < 1)) {
            converged = true;
#pragma unroll
            for (int i = 0; i < 16; ++i) prevT[i] = T[i];
        }
    }
    if (converged) status = WM_OK;
    else if (cnt < 4) status = WM_TOO_FEW_CORRESPONDENCES;
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) out.T[i] = (double) prevT[i];
        out.f = f_last;
        out.converged = converged ? 1 : 0;
        out.iterations = iter;
        out.n_corr = (int) cnt;
        out.inner_total = inner_total;
        out.evaluations = S.evals;
        out.status = status;
        out.cyc[2] = cyc_search;
        out.cyc[3] = cyc_min;
    }
}

// ---- host
struct GicpSmallBatch {
    DevBuf d_stage;  // [table | clouds]
    DevBuf d_work;   // the pairs' scratch
    DevBuf d_out;
    void *h_stage = nullptr;
    size_t h_stage_cap = 0;
    void *h_out = nullptr;
    size_t h_out_cap = 0;
};

static GicpSmallBatch *gs_of(wm_ctx *ctx) {
    if (!ctx->gicp_small_batch) ctx->gicp_small_batch = new (std::nothrow) GicpSmallBatch();
    return static_cast<GicpSmallBatch *>(ctx->gicp_small_batch);
}

void gicp_small_release(wm_ctx *ctx) {
    GicpSmallBatch *b = static_cast<GicpSmallBatch *>(ctx->gicp_small_batch);
    if (!b) return;
    b->d_stage.release();
    b->d_work.release();
    b->d_out.release();
    if (b->h_stage) (void) hipHostFree(b->h_stage);
    if (b->h_out) (void) hipHostFree(b->h_out);
    delete b;
    ctx->gicp_small_batch = nullptr;
}



struct GsJob {
    const void *src;
    size_t n_src;
    const void *tgt;
    size_t n_tgt;
};

// `n` registrations (no empty cloud among them) in one launch; out[k] = what the kernel left for job k
static int gicp_small_run(wm_ctx *ctx, const GsJob *jobs, int n, size_t stride, int mem, const wm_gicp_params *prm, GsOut *res,
                          float *kernel_ms) {
    if (n <= 0) return WM_OK;
    GicpSmallBatch *B = gs_of(ctx);
    if (!B) return WM_ERR_NOMEM;
    size_t cloud_bytes = 0, work_bytes = 0;
    auto work_need = [](size_t ns, size_t nt) {
        return align_up256((ns + 4) * 16) + align_up256((nt + 4) * 16) + 3 * align_up256(((size_t) kGsCells + 8) * 4) + align_up256(ns * 72) + align_up256(nt * 72) +
               align_up256(((ns + 63) / 64) * (size_t) kGsEvBlock) + align_up256(ns * 4);
    };
    for (int k = 0; k < n; ++k) {
        if (jobs[k].n_src == 0 || jobs[k].n_tgt == 0 || jobs[k].n_src > (size_t) WM_GICP_BATCH_MAX_POINTS ||
            jobs[k].n_tgt > (size_t) WM_GICP_BATCH_MAX_POINTS)
            return WM_ERR_ARG;
        cloud_bytes += align_up256(jobs[k].n_src * stride) + align_up256(jobs[k].n_tgt * stride);
        work_bytes += work_need(jobs[k].n_src, jobs[k].n_tgt);
    }
    const size_t table_bytes = align_up256((size_t) n * sizeof(GsPair));
    const size_t up_bytes = table_bytes + (mem == WM_MEM_HOST ? cloud_bytes : 0);
    WM_HIP(ctx, B->d_stage.reserve(up_bytes));
    WM_HIP(ctx, B->d_work.reserve(work_bytes));
    WM_HIP(ctx, B->d_out.reserve((size_t) n * sizeof(GsOut)));
    WM_TRY(pinned_reserve(ctx, &B->h_stage, &B->h_stage_cap, up_bytes));
    WM_TRY(pinned_reserve(ctx, &B->h_out, &B->h_out_cap, (size_t) n * sizeof(GsOut)));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the stream may still be reading the staging buffer for the previous batch)
    unsigned char *h = static_cast<unsigned char *>(B->h_stage), *d = B->d_stage.as<unsigned char>(), *w = B->d_work.as<unsigned char>();
    GsPair *table = reinterpret_cast<GsPair *>(h);
    size_t off = table_bytes, sent = table_bytes;
    for (int k = 0; k < n; ++k) {
        const GsJob &it = jobs[k];
        GsPair &t = table[k];
        t.n_src = (unsigned) it.n_src;
        t.n_tgt = (unsigned) it.n_tgt;
        if (mem == WM_MEM_HOST) {
            memcpy(h + off, it.src, it.n_src * stride);
            t.src = d + off;
            off += align_up256(it.n_src * stride);
            memcpy(h + off, it.tgt, it.n_tgt * stride);
            t.tgt = d + off;
            off += align_up256(it.n_tgt * stride);
            if (off - sent >= ((size_t) 2 << 20)) {  // (each slice's DMA runs under the next slices' copies)
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
        t.s_pts = reinterpret_cast<float4 *>(take((it.n_src + 4) * 16));
        t.t_pts = reinterpret_cast<float4 *>(take((it.n_tgt + 4) * 16));
        t.s_cs = reinterpret_cast<unsigned *>(take(((size_t) kGsCells + 8) * 4));
        t.t_cs = reinterpret_cast<unsigned *>(take(((size_t) kGsCells + 8) * 4));
        t.run = reinterpret_cast<unsigned *>(take(((size_t) kGsCells + 8) * 4));
        t.c1 = reinterpret_cast<double *>(take(it.n_src * 72));
        t.c2 = reinterpret_cast<double *>(take(it.n_tgt * 72));
        t.evb = take(((it.n_src + 63) / 64) * (size_t) kGsEvBlock);
        t.match = reinterpret_cast<unsigned *>(take(it.n_src * 4));
    }
    if (off > sent) WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(ctx, hipMemcpyAsync(d, h, table_bytes, hipMemcpyHostToDevice, ctx->stream));
    GsParams P;
    memset(&P, 0, sizeof(P));
    P.stride = (unsigned) stride;
    P.k = prm->corr_rand;
    P.eps = prm->gicp_epsilon;
    P.thr_d2 = threshold_d2_strict(prm->max_corr);
    P.max_iter = prm->max_iter;
    P.max_inner = prm->max_inner;
    P.forced = prm->force_iterations;
    P.r_eps = prm->r_eps;
    P.t_eps = prm->t_eps;
    P.objective = prm->objective;
    // 1: trace.  Bits 2 / 4 are timing experiments that give WRONG registrations (no Mahalanobis matrices / no
    // search): they exist only in a developer build (-DWM_GICP_SMALL_EXPERIMENTS); a stray environment variable
    // must not be able to switch them on in the production library
    P.debug = getenv("WM_GICP_SMALL_TRACE") ? atoi(getenv("WM_GICP_SMALL_TRACE")) : 0;
#ifndef WM_GICP_SMALL_EXPERIMENTS
    P.debug &= 1;
#endif
    WM_HIP(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    const GsPair *dt = reinterpret_cast<const GsPair *>(d);
    if (P.k <= 10)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gicp_small<10>), dim3((unsigned) n), dim3(kGsThreads), 0, ctx->stream, dt, P, B->d_out.as<GsOut>());
    else if (P.k <= 20)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gicp_small<20>), dim3((unsigned) n), dim3(kGsThreads), 0, ctx->stream, dt, P, B->d_out.as<GsOut>());
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gicp_small<32>), dim3((unsigned) n), dim3(kGsThreads), 0, ctx->stream, dt, P, B->d_out.as<GsOut>());
    WM_HIP(ctx, hipGetLastError());
    WM_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    WM_HIP(ctx, hipMemcpyAsync(B->h_out, B->d_out.p, (size_t) n * sizeof(GsOut), hipMemcpyDeviceToHost, ctx->stream));
    WM_TRY(sync_sleeping(ctx));  // (milliseconds: the registrations of the whole batch)
#ifdef WM_COV_COUNT
    {
        double c[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void) hipMemcpyFromSymbol(c, HIP_SYMBOL(g_knn_cnt), sizeof(c));
        (void) hipMemcpyToSymbol(HIP_SYMBOL(g_knn_cnt), z, sizeof(z));
        fprintf(stderr, "[knn small] candidates %.0f, wave trips %.0f (x64 = %.0f lane slots: %.1f %% used), lane batches %.0f, wave batches %.0f, lane passes %.0f\n",
                c[0], c[1], c[1] * 64, 100.0 * c[0] / (c[1] * 64 + 1), c[2], c[3], c[4]);
    }
#endif
    if (kernel_ms) (void) hipEventElapsedTime(kernel_ms, ctx->ev_a, ctx->ev_b);
    memcpy(res, B->h_out, (size_t) n * sizeof(GsOut));
    return WM_OK;
}

}  // namespace wm

using namespace wm;

extern "C" {

int wm_gicp_batch_match(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride, int mem, const wm_gicp_params *p,
                        float res, double *T_out, wm_gicp_stats *stats, int *status, float *kernel_ms) {
    if (!ctx || !p || !status || n_items < 0 || (n_items > 0 && !items) || stride < 12 || (stride & 3)) return WM_ERR_ARG;
    if (p->corr_rand < 1 || p->corr_rand > 32 || !(p->max_corr > 0)) return WM_ERR_ARG;
    if (p->objective != WM_GICP_OBJECTIVE_PCL_SUMS && p->objective != WM_GICP_OBJECTIVE_STATISTICS) return WM_ERR_ARG;
    if (p->force_iterations <= 0 && p->max_iter <= 0) return WM_ERR_ARG;
    if (kernel_ms) *kernel_ms = 0;
    if (n_items == 0) return WM_OK;
    for (int k = 0; k < n_items; ++k) {
        const wm_batch_item &it = items[k];
        if ((it.n_src > 0 && !it.src) || (it.n_target > 0 && !it.target) || it.n_src > 0x7FFFFFF0u || it.n_target > 0x7FFFFFF0u)
            return WM_ERR_ARG;
        if (!(res > 0) && (it.n_src > (size_t) WM_GICP_BATCH_MAX_POINTS || it.n_target > (size_t) WM_GICP_BATCH_MAX_POINTS)) return WM_ERR_ARG;
    }
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats) * (size_t) n_items);
    std::vector<GsJob> jobs;
    std::vector<int> item_of;
    std::vector<int> one_by_one;
    size_t run_stride = stride;
    int run_mem = mem;
    if (res > 0) {
        // GICPMatcher::setRef / setTarget with res > 0 (gicp.cpp:38-55): every cloud of the batch through VoxelGrid in
        // one pass of device-wide kernels (wm_batch.hip); the filtered copies are what is registered
        std::vector<int> idx;
        for (int k = 0; k < n_items; ++k) idx.push_back(k);
        const float4 *filtered = nullptr;
        std::vector<unsigned> off, n_out;
        WM_TRY(batch_voxel_filter(ctx, items, idx, stride, mem, res, &filtered, off, n_out));
        for (int k = 0; k < n_items; ++k) {
            const unsigned ns = n_out[2 * (size_t) k], nt = n_out[2 * (size_t) k + 1];
            // (a leaf lattice beyond int32 -- PCL returns the cloud unfiltered --, or a filtered cloud beyond the kernel: one by one)
            if (ns == 0xFFFFFFFFu || nt == 0xFFFFFFFFu || ns > (unsigned) WM_GICP_BATCH_MAX_POINTS || nt > (unsigned) WM_GICP_BATCH_MAX_POINTS) {
                one_by_one.push_back(k);
                continue;
            }
            status[k] = (ns == 0 || nt == 0) ? WM_ERR_STATE : WM_OK;
            if (status[k] != WM_OK) continue;
            jobs.push_back(GsJob{filtered + off[2 * (size_t) k], ns, filtered + off[2 * (size_t) k + 1], nt});
            item_of.push_back(k);
        }
        run_stride = sizeof(float4);
        run_mem = WM_MEM_DEVICE;
    } else {
        for (int k = 0; k < n_items; ++k) {
            const wm_batch_item &it = items[k];
            // (wm_gicp_align on an empty cloud: WM_ERR_STATE)
            status[k] = (it.n_src == 0 || it.n_target == 0) ? WM_ERR_STATE : WM_OK;
            if (status[k] != WM_OK) continue;
            jobs.push_back(GsJob{it.src, it.n_src, it.target, it.n_target});
            item_of.push_back(k);
        }
    }
    if (!jobs.empty()) {
        std::vector<GsOut> got(jobs.size());
        WM_TRY(gicp_small_run(ctx, jobs.data(), (int) jobs.size(), run_stride, run_mem, p, got.data(), kernel_ms));
        for (size_t j = 0; j < jobs.size(); ++j) {
            const int k = item_of[j];
            const GsOut &r = got[j];
            status[k] = r.status;
            if (stats) {
                stats[k].converged = r.converged;
                stats[k].iterations = r.iterations;
                stats[k].n_corr = r.n_corr;
                stats[k].inner_total = r.inner_total;
                stats[k].evaluations = r.evaluations;
                stats[k].f_final = r.f;
            }
            if (r.status == WM_OK && T_out) memcpy(T_out + 16 * (size_t) k, r.T, sizeof(r.T));
            if (ctx->trace)
                fprintf(stderr, "[wm] gicp batch: pair %d: %d + %d points, status %d, %d outer / %d inner iterations, %d evaluations; kcycles: grids %llu, covariances %llu, searches %llu, minimisations %llu\n",
                        k, r.n_src_valid, r.n_tgt_valid, r.status, r.iterations, r.inner_total, r.evaluations, r.cyc[0] / 1000, r.cyc[1] / 1000,
                        r.cyc[2] / 1000, r.cyc[3] / 1000);
        }
    }
    for (int k : one_by_one) {
        double T[16];
        wm_gicp_stats s;
        const int rc = wm_gicp_match(ctx, items[k].src, items[k].n_src, items[k].target, items[k].n_target, stride, mem, p, res, T, &s);
        if (rc < 0 && rc != WM_ERR_STATE) return rc;
        status[k] = rc;
        if (stats) stats[k] = s;
        if (rc == WM_OK && T_out) memcpy(T_out + 16 * (size_t) k, T, sizeof(T));
    }
    return WM_OK;
}

}  // extern "C"
