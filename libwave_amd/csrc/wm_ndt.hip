// wm_ndt.hip -- pcl::NormalDistributionsTransform on device, as libwave's NDTMatcher
// drives it (wave_matching/src/ndt.cpp:18-34 setters, :48-65 setInput*/align):
//   k_ndt_key / rocPRIM sort / k_ndt_heads / k_ndt_voxel_stats / k_ndt_hash_insert, k_ndt_dense_fill
//        = pcl::VoxelGridCovariance::filter (setInputTarget, ndt.cpp:55): per-voxel
//          n, sum p, sum p p^T (double, ascending point order), mean, covariance,
//          eigenvalue inflation (>= 0.01 lambda_max), inverse; voxels with < 6 points
//          dropped; a dense cell -> record table over the target's bounding box (or, for
//          huge lattices, an open-addressing hash) replaces PCL's kd-tree over the voxel
//          means (a mean within `res` of a point lies in one of the 27 voxels around it).
//   k_ndt_derivs  = computeDerivatives / updateDerivatives / computeHessian: one lane per
//          source point; pass 1 lists the point's voxels within `res`, pass 2 adds their
//          terms: score + 6 gradient + the 21 upper-triangle Hessian sums in double,
//          fixed-order workgroup reduction.  k_sum_fetch (wm_icp.hip) adds the workgroups'
//          partials; those 28 doubles are the only thing the host sees.
//   host: Newton step (JacobiSVD solve) + More-Thuente line search
//          (computeStepLengthMT / trialValueSelectionMT / updateIntervalMT).
// [PCL registration/impl/ndt.hpp, filters/impl/voxel_grid_covariance.hpp; Magnusson 2009]
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "wm_internal.hpp"
#include "wm_sort.hpp"
#include "wm_ndt_dev.hpp"
#include "wm_ndt_ctl.hpp"

#include <chrono>
#include <thread>
#include <float.h>
#include <math.h>

namespace wm {

constexpr int kNdtBlocks = 4096;  // upper bound; ctx->tune_ndt_blocks workgroups per pass
constexpr unsigned long long kEmptyKey = ~0ull;


__host__ __device__ inline unsigned long long ndt_key(int i, int j, int k) {
    return ((unsigned long long) (unsigned) (i + (1 << 20)) << 42) |
           ((unsigned long long) (unsigned) (j + (1 << 20)) << 21) |
           (unsigned long long) (unsigned) (k + (1 << 20));
}

__device__ __forceinline__ unsigned ndt_hash(unsigned long long x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (unsigned) x;
}

// The voxel lattice over the target's bounding box.  Points are SORTED by their linear cell
// number in it, (k * ny + j) * nx + i -- the same order as (k, j, i) -- which needs only
// log2(cells) key bits: two to four radix passes instead of the eight of a 64-bit key.  The
// (k, j, i) key of a voxel (hash grid, dense table) is rebuilt from it once per voxel.
struct NdtLattice {
    int lo[3];                  // cell of the bounding box's low corner
    unsigned long long nx, ny;  // cells along x, y
    unsigned long long cells;   // nx * ny * nz = the "no cell" key of non-finite points (sorts last)
};

// (KT: the key's type -- 32 bits while the lattice has fewer than 2^32 cells: the sort then moves 8 bytes per point and
// pass instead of 12)
template <class KT>
__global__ void __launch_bounds__(kBlock)
    k_ndt_key(const float4 *__restrict__ pts, unsigned n, float inv, NdtLattice L, KT *keys,
              unsigned *perm) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    unsigned long long key = L.cells;
    if (p.x == p.x) {
        const long long a = (long long) floorf(__fmul_rn(p.x, inv)) - L.lo[0];
        const long long b = (long long) floorf(__fmul_rn(p.y, inv)) - L.lo[1];
        const long long c = (long long) floorf(__fmul_rn(p.z, inv)) - L.lo[2];
        key = ((unsigned long long) c * L.ny + (unsigned long long) b) * L.nx + (unsigned long long) a;
    }
    keys[i] = (KT) key;
    perm[i] = i;
}

__device__ __forceinline__ unsigned long long ndt_key_of_cell(unsigned long long cell, const NdtLattice &L) {
    const unsigned long long row = cell / L.nx;
    const int a = (int) (cell - row * L.nx) + L.lo[0];
    const unsigned long long lay = row / L.ny;
    const int b = (int) (row - lay * L.ny) + L.lo[1];
    const int c = (int) lay + L.lo[2];
    return ndt_key(c, b, a);  // k in the top bits
}

template <class KT>
__global__ void __launch_bounds__(kBlock)
    k_ndt_flags(const KT *__restrict__ keys, unsigned n, KT invalid,
                unsigned *flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const KT k = keys[i];
    flags[i] = (k != invalid && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}


// heads[slot] = first sorted position of voxel `slot`; heads[n_voxels] = one past the last
// finite point (seg[n] = n_voxels, the scan's total)
template <class KT>
__global__ void __launch_bounds__(kBlock)
    k_ndt_heads(const KT *__restrict__ keys, const unsigned *__restrict__ flags,
                const unsigned *__restrict__ seg, unsigned n, KT invalid,
                unsigned *__restrict__ heads) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) heads[seg[i]] = i;
    if (keys[i] != invalid && (i + 1 == n || keys[i + 1] == invalid)) heads[seg[n]] = i + 1;
}

// mean, covariance, PCL's eigenvalue conditioning and the inverse of one voxel from its sums
// (s = sum p, pp = sum p p^T over `count` points); writes the voxel's records
__device__ inline void ndt_voxel_finish(unsigned slot, unsigned long long cell, const NdtLattice &L,
                                        unsigned count, const double *s, const double *pp,
                                        NdtVoxel *__restrict__ vox,
                                        float4 *__restrict__ meanf, unsigned long long *__restrict__ vkey,
                                        unsigned *__restrict__ n_valid) {
    NdtVoxel v;
    const bool valid = ndt_voxel_record(count, s, pp, v);
    vox[slot] = v;
    // the radius test of the derivative passes runs on float means (PCL's kd-tree of centroids)
    meanf[slot] = make_float4((float) v.mean[0], (float) v.mean[1], (float) v.mean[2], 0.0f);
    vkey[slot] = valid ? ndt_key_of_cell(cell, L) : kEmptyKey;
    if (valid) atomicAdd(n_valid, 1u);
}

// The voxels' statistics: 3 sums of p and 9 of p p^T, each formed in ASCENDING POINT ORDER (the model does not depend on
// which of the two routines below forms a voxel's sums), then PCL's covariance conditioning (ndt_voxel_record: a Jacobi
// eigen-decomposition, ~15 us of dependent scalar arithmetic whoever runs it).  Two launches:
//   k_ndt_voxel_sums   a WAVE per voxel above `split` points: the twelve sums -> vsum[slot][12]
//   k_ndt_voxel_stats  a LANE per voxel: up to `split` points it walks them itself, above it takes the wave's sums; then
//                      the conditioning, 64 voxels of a wave side by side
// (until round 5 the wave kernel conditioned its voxels itself, one lane of 64 at work, and converted its operands on the
// chain: 102 us for the 2 952 crowded voxels of the 2M-point ring scan behind 34 us of the lane kernel; now 56 + 35 us at
// a split of 128 points, 70 + 16 at 32 -- profiles/r06_experiments.md.)
constexpr unsigned kVoxWaveAvg = 192;  // grids with more points per voxel on average than this: every voxel's sums are a wave's
constexpr unsigned kVoxLaneMax = 128;  // finer grids: voxels with more points than this ...
constexpr unsigned kVoxLaneMaxFew = 32;  // ... than this, when the grid has so few voxels (kVoxFew) that a wave each costs less than
constexpr unsigned kVoxFew = 32768;      // the lanes' walks: a wave pays ~4 us of dependent round trips per voxel, a lane ~1 us per 4 points
constexpr int kVoxDepth = 4;           // batches of 64 points a wave has in flight

// One WAVE per voxel (PCL's default NDT resolution of 5 m puts thousands of points in a voxel; a lidar's rings put
// 1 700 into a 0.5 m voxel next to the sensor).  The sums are still formed one point after the other, but kVoxDepth x 64
// points are gathered at a time into LDS and the twelve sums advance in twelve lanes side by side, each adding the
// points in order -- the same additions as the lane routine, in the same order.  The NEXT 256 points are requested
// before this trip's additions start.
__global__ void __launch_bounds__(kBlock)
    k_ndt_voxel_sums(const float4 *__restrict__ pts, const unsigned *__restrict__ perm, const unsigned *__restrict__ heads,
                     unsigned nvox, unsigned min_count, double *__restrict__ vsum) {
    // A voxel's cost is its chain of dependent additions (3 343 points in the fullest voxel of the 2M-point ring scan:
    // one wave, nothing to hide behind).  So nothing but the additions is ON the chain: the 64 gathering lanes convert
    // their points to double and write doubles to LDS (float operands converted by the twelve adding lanes were 32
    // converts per 16 points on the chain: 105 us for the launch on that scan, 56 now -- profiles/r06_experiments.md), a
    // row of ones serves the three plain sums (fma(x, 1, acc) = acc + x exactly: one loop for all twelve lanes), and a
    // chunk's operands are read while the previous chunk's additions run.
    constexpr unsigned kTrip = 64u * (unsigned) kVoxDepth;
    // (rows x, y, z, ones; two doubles of padding: the twelve adding lanes read the same position of three or four
    // different rows in one instruction -- rows a multiple of 256 bytes apart would be the same LDS banks)
    constexpr unsigned kRow = kTrip + 2u;
    __shared__ __attribute__((aligned(16))) double s_all[kBlock / 64][4][kRow];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __shared__ unsigned s_ids[kBlock / 64][kTrip];  // the next trip's point numbers
    double (*s_p)[kRow] = s_all[wave];
    unsigned *s_id = s_ids[wave];
#pragma unroll
    for (int q = 0; q < kVoxDepth; ++q) s_p[3][64 * q + lane] = 1.0;
    const unsigned waves_total = gridDim.x * (kBlock / 64);
    // lane 0..2: sum of p[lane] (times the ones); lane 3..11: sum of p[a] * p[b]
    const unsigned ia = lane < 3 ? lane : (lane < 12 ? (lane - 3) / 3 : 0);
    const unsigned ib = lane < 3 ? 3 : (lane < 12 ? (lane - 3) % 3 : 0);
    constexpr int kChunk = 8;  // points per chunk of the chain (eight 16-byte reads; a chunk's and the next one's stay countable in lgkmcnt)
    auto operands = [&](unsigned u, double (&x)[kChunk], double (&y)[kChunk]) {
#pragma unroll
        for (int q = 0; q < kChunk / 2; ++q) {
            const double2 a = *reinterpret_cast<const double2 *>(&s_p[ia][u + 2u * (unsigned) q]);
            const double2 b = *reinterpret_cast<const double2 *>(&s_p[ib][u + 2u * (unsigned) q]);
            x[2 * q] = a.x, x[2 * q + 1] = a.y;
            y[2 * q] = b.x, y[2 * q + 1] = b.y;
        }
    };
    for (unsigned slot = blockIdx.x * (kBlock / 64) + wave; slot < nvox; slot += waves_total) {
        // (wave-uniform, and the compiler is told so: the loops below are then scalar loops, not exec-mask ones)
        const unsigned i = (unsigned) __builtin_amdgcn_readfirstlane((int) heads[slot]);
        const unsigned j = (unsigned) __builtin_amdgcn_readfirstlane((int) heads[slot + 1]);
        if (j - i < min_count) continue;  // a small voxel: a lane's
        double acc = 0.0;
        // What goes from one trip to the next goes through LDS, not through registers: trip k gathers the points of trip
        // k + 1 (their numbers came a trip earlier) and loads the numbers of trip k + 2 into registers that live within the
        // trip; BEHIND its additions it waits for them, converts and writes them to LDS.  (With the gathered points carried
        // in registers across the loop the compiler put its copies into the carried registers -- and the wait for the
        // gather -- right behind the loads: gather, numbers and additions one after the other.)
        auto numbers = [&](unsigned t, unsigned (&id)[kVoxDepth]) {
#pragma unroll
            for (int q = 0; q < kVoxDepth; ++q) id[q] = perm[min(t + 64u * (unsigned) q + lane, j - 1u)];
        };
        // (a point as three separate words, not one 12-byte load: the compiler splits a three-register result that lives
        // across the additions into single registers -- and waits for the load to do so)
        struct P3 {
            float x, y, z;
        };
        auto gather = [&](const unsigned (&id)[kVoxDepth], P3 (&p)[kVoxDepth]) {
            const float *f = reinterpret_cast<const float *>(pts);
#pragma unroll
            for (int q = 0; q < kVoxDepth; ++q) {
                const float *fx = f + 4u * (size_t) id[q], *fy = fx + 1, *fz = fx + 2;
                asm volatile("" : "+v"(fy), "+v"(fz));  // (three addresses the compiler cannot see to be adjacent: it would merge the loads again)
                p[q].x = *fx;
                p[q].y = *fy;
                p[q].z = *fz;
            }
        };
        auto stage = [&](const P3 (&p)[kVoxDepth]) {
#pragma unroll
            for (int q = 0; q < kVoxDepth; ++q) {
                s_p[0][64 * q + lane] = (double) p[q].x;
                s_p[1][64 * q + lane] = (double) p[q].y;
                s_p[2][64 * q + lane] = (double) p[q].z;
            }
        };
        {
            unsigned id0[kVoxDepth], id1[kVoxDepth];
            P3 p0[kVoxDepth];
            numbers(i, id0);
            numbers(i + kTrip, id1);  // (clamped to the voxel's last point: harmless past its end)
            gather(id0, p0);
            __builtin_amdgcn_wave_barrier();  // (the previous voxel's reads are done: LDS operations of a wave execute in order)
#pragma unroll
            for (int q = 0; q < kVoxDepth; ++q) s_id[64 * q + lane] = id1[q];
            stage(p0);
        }
        for (unsigned t = i; t < j; t += kTrip) {
            const bool more = t + kTrip < j;  // (wave-uniform)
            P3 p1[kVoxDepth];
            unsigned id2[kVoxDepth];
            if (more) {
                unsigned id1[kVoxDepth];
#pragma unroll
                for (int q = 0; q < kVoxDepth; ++q) id1[q] = s_id[64 * q + lane];
                gather(id1, p1);
                numbers(t + 2u * kTrip, id2);
            }
            // (a wave's LDS traffic is in order: no barrier between its own write and read)
            if (lane < 12) {
                const unsigned m = j - t < kTrip ? j - t : kTrip;
                const unsigned mc = m & ~(unsigned) (kChunk - 1);
                if (mc) {
                    // (the next chunk's reads are issued UNCONDITIONALLY -- past the voxel's end they fetch what is not
                    // used, the position clamped to the row -- so that no branch separates them from the additions they
                    // hide behind: with the reads under an `if` the compiler's wait counts had to hold on both paths and
                    // every chunk waited for its own reads)
                    constexpr unsigned kLast = kTrip - (unsigned) kChunk;
                    double xa[kChunk], ya[kChunk], xb[kChunk], yb[kChunk];
                    operands(0u, xa, ya);
                    // (the scheduling barriers keep that order: left alone, the compiler gathers both chunks' reads at the top
                    // of the loop and runs sixteen additions behind them)
                    for (unsigned u = 0;; u += 2u * kChunk) {
                        operands(min(u + (unsigned) kChunk, kLast), xb, yb);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < kChunk; ++q) acc = fma(xa[q], ya[q], acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (u + kChunk >= mc) break;
                        operands(min(u + 2u * kChunk, kLast), xa, ya);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < kChunk; ++q) acc = fma(xb[q], yb[q], acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (u + 2u * kChunk >= mc) break;
                    }
                }
                for (unsigned u = mc; u < m; ++u) acc = fma(s_p[ia][u], s_p[ib][u], acc);
            }
            __builtin_amdgcn_sched_barrier(0);  // (the waits for the gather stay behind the additions)
            if (more) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < kVoxDepth; ++q) s_id[64 * q + lane] = id2[q];
                stage(p1);
            }
        }
        if (lane < 12) vsum[(size_t) slot * 12u + lane] = acc;  // [0..2] sum p, [3..11] sum p p^T
    }
}

// one lane per voxel (compacted: every lane of a wave has a voxel): the sums -- its own walk in ascending point order,
// or k_ndt_voxel_sums' --, then the conditioning
constexpr int kVoxStatBlock = 64;
template <class KT>
__global__ void __launch_bounds__(kVoxStatBlock)
    k_ndt_voxel_stats(const float4 *__restrict__ pts, const KT *__restrict__ keys,
                      const unsigned *__restrict__ perm, const unsigned *__restrict__ heads,
                      unsigned nvox, unsigned max_count, const double *__restrict__ vsum, NdtLattice L,
                      NdtVoxel *__restrict__ vox, float4 *__restrict__ meanf, unsigned long long *__restrict__ vkey,
                      unsigned *__restrict__ n_valid) {
    const unsigned slot = blockIdx.x * kVoxStatBlock + threadIdx.x;
    if (slot >= nvox) return;
    const unsigned i = heads[slot], j = heads[slot + 1];
    const unsigned long long key = (unsigned long long) keys[i];
    double s[3] = {0, 0, 0}, pp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (j - i > max_count) {  // a crowded voxel: a wave formed its sums
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] = vsum[(size_t) slot * 12u + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) pp[k] = vsum[(size_t) slot * 12u + 3u + k];
    } else {
#pragma unroll 4
        for (unsigned t = i; t < j; ++t) {
            const float4 p = pts[perm[t]];
            const double d[3] = {(double) p.x, (double) p.y, (double) p.z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s[a] += d[a];
#pragma unroll
                for (int b = 0; b < 3; ++b) pp[a * 3 + b] = fma(d[a], d[b], pp[a * 3 + b]);
            }
        }
    }
    ndt_voxel_finish(slot, key, L, j - i, s, pp, vox, meanf, vkey, n_valid);
}

__global__ void __launch_bounds__(kBlock)
    k_ndt_hash_insert(const unsigned long long *__restrict__ vkey, unsigned nvox,
                      unsigned long long *__restrict__ hkeys, unsigned *__restrict__ hvals,
                      unsigned mask) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nvox) return;
    const unsigned long long key = vkey[i];
    if (key == kEmptyKey) return;
    unsigned h = ndt_hash(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&hkeys[h], kEmptyKey, key);
        if (prev == kEmptyKey) {
            hvals[h] = i;
            return;
        }
        h = (h + 1) & mask;
    }
}

// Dense alternative to the hash grid when the voxel lattice over the target's bounding box is
// small (a few million cells): cell (i, j, k) -> voxel slot, -1 = empty / invalid.  A neighbour
// look-up is then ONE 4-byte load instead of a hash and a probe sequence.

__global__ void __launch_bounds__(kBlock)
    k_ndt_dense_fill(const unsigned long long *__restrict__ vkey, unsigned nvox, NdtDense d, int *table,
                     const float4 *__restrict__ meanf, float4 *__restrict__ cells4) {
    const unsigned s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nvox) return;
    const unsigned long long key = vkey[s];
    if (key == kEmptyKey) return;
    // ndt_key(k, j, i): k in the top bits, 21 bits each, biased by 2^20
    const int k = (int) ((key >> 42) & 0x1FFFFFu) - (1 << 20);
    const int j = (int) ((key >> 21) & 0x1FFFFFu) - (1 << 20);
    const int i = (int) (key & 0x1FFFFFu) - (1 << 20);
    const int a = i - d.i0, b = j - d.j0, c = k - d.k0;
    if (a < 0 || b < 0 || c < 0 || a >= d.nx || b >= d.ny || c >= d.nz) return;
    table[((size_t) c * d.ny + b) * d.nx + a] = (int) s;
    if (cells4) {
        const float4 m = meanf[s];
        cells4[((size_t) c * d.ny + b) * d.nx + a] = make_float4(m.x, m.y, m.z, __uint_as_float(s));
    }
}


__device__ __forceinline__ double dot3d(const double *a, const double *b) {
#pragma clang fp contract(fast)  // (used by k_ndt_derivs only: see there)
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// where the last workgroup of a pass leaves the sums for the host (ticket == nullptr: it does not; k_sum_fetch does)
constexpr unsigned kFetchGroup = 16;    // workgroups whose rows one of them adds
constexpr unsigned kFetchRowsMax = 64;  // rows one add_rows call takes at most: kFetchGroup, and the groups of a pass
struct NdtFetch {
    unsigned *ticket;  // device words, zero between passes: [0] the groups', [1 + g] group g's workgroups'
    void *dst;         // pinned host memory: 16-byte slots {double value, unsigned number of the pass, 0}
    unsigned seq;      // this pass's number (never 0)
};

// a[0] = score, a[1..6] = gradient, a[7..27] = upper triangle of the Hessian, row by row
template <bool GRAD, bool HESS>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HESS ? 3 : 4, HESS ? 3 : 4)))
    k_ndt_derivs(const float4 *__restrict__ src, unsigned n, unsigned n_total, unsigned shard_rank,
                 unsigned shard_world, const NdtVoxel *__restrict__ vox,
                 const float4 *__restrict__ meanf, const unsigned long long *__restrict__ hkeys, const unsigned *__restrict__ hvals,
                 unsigned mask, NdtDense dense, const float4 *__restrict__ cells4, NdtArgs A, double *__restrict__ partials,
                 NdtFetch F) {
    // The double-precision algebra of this kernel may fuse a multiply with the add that follows it
    // (the library is built with -ffp-contract=off for the FLOAT arithmetic that has to reproduce PCL's
    // bits: the point transform and the radius test below, written with explicit _rn intrinsics, are not
    // affected).  Score, gradient and Hessian are sums of ~10^7 terms compared with the oracle at 1e-8
    // relative; a fused term differs from an unfused one by half an ulp.  It is 46 M of the kernel's
    // 72 M wave instructions that pair up.
#pragma clang fp contract(fast)
    __shared__ unsigned s_near[27 * kBlock];  // per-lane lists, pass 1 -> pass 2 (lane-private)
    // the 8 + 15 angle-derivative vectors: read once per point, after its voxels -- as kernel arguments they
    // are 138 scalar registers the compiler spills lane by lane (171 spills); here they are LDS broadcasts
    __shared__ double s_jh[23][3];
    // (the reads must stay where they are used -- hoisted out of the point loop as loop invariants they would
    // be 138 registers: the index carries a zero the compiler cannot see through, made once per point)
    auto jh_dot = [&](const double (&x)[3], int k, int opaque0) -> double {
        const double *v = &s_jh[k + opaque0][0];
        return x[0] * v[0] + x[1] * v[1] + x[2] * v[2];
    };
    if (GRAD || HESS) {
        if (threadIdx.x < 69u) (&s_jh[0][0])[threadIdx.x] = threadIdx.x < 24u ? (&A.j[0][0])[threadIdx.x] : (&A.h[0][0])[threadIdx.x - 24u];
        __syncthreads();
    }
    constexpr int NA = HESS ? kNdtAcc : kNdtAccGrad;  // a gradient pass carries (and ships) 7 sums, not 28
    double acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0;
    // n = local count (a multiple of the chunk when sharded); the global index interleaves the
    // ranks chunk by chunk, so every rank works on a uniform sample of the Morton-ordered cloud
    for (unsigned loc = blockIdx.x * kBlock + threadIdx.x; loc < n; loc += gridDim.x * kBlock) {
        const unsigned idx = shard_world > 1
                                 ? (((loc >> kNdtChunkLog2) * shard_world + shard_rank) << kNdtChunkLog2) |
                                       (loc & ((1u << kNdtChunkLog2) - 1u))
                                 : loc;
        if (idx >= n_total) continue;
        const float4 sp = src[idx];
        const float xt0 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.Tf[0], sp.x), __fmul_rn(A.Tf[1], sp.y)),
                                              __fmul_rn(A.Tf[2], sp.z)), A.Tf[3]);
        const float xt1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.Tf[4], sp.x), __fmul_rn(A.Tf[5], sp.y)),
                                              __fmul_rn(A.Tf[6], sp.z)), A.Tf[7]);
        const float xt2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.Tf[8], sp.x), __fmul_rn(A.Tf[9], sp.y)),
                                              __fmul_rn(A.Tf[10], sp.z)), A.Tf[11]);
        const int ci = (int) floorf(__fmul_rn(xt0, A.inv_res));
        const int cj = (int) floorf(__fmul_rn(xt1, A.inv_res));
        const int ck = (int) floorf(__fmul_rn(xt2, A.inv_res));
        // Pass 1: the 27 neighbouring cells -> this lane's own list of voxels within `res`
        // (slot numbers in LDS, column = lane).  Only about one neighbour in five passes, and
        // WHICH ones differs from lane to lane; evaluating inside the 27-trip loop would run the
        // f64 body with most lanes masked off.  Pass 2 walks the compacted list, so a wave pays
        // max-over-lanes(list length) bodies instead of one per cell any lane needs.  The list
        // keeps cell order, so every lane still adds its voxels in the same order as before.
        // 1a: the occupied cells of the 3x3x3 block -> candidate list.  dense lattice: the block's
        // centre in table coordinates; the table carries two empty cells of margin, so a centre in
        // [1, n-2] reads in bounds and any other has no neighbours.  All nine row loads (three
        // adjacent cells each) are issued together.
        int n_cand = 0;
        int n_near = 0;
        if (cells4) {  // uniform
            // The lattice as float4 cells (the voxel's float mean + its slot; an empty cell's "mean" is 3.4e38, which
            // no point is near): the 27 cells ARE the 27 radius tests -- one memory round trip per plane of nine and no
            // candidate list between finding a voxel and testing it, where the slot table + the means' gather are three
            // to four dependent trips (the passes run at the two waves per SIMD the f64 algebra leaves room for, which
            // hide little).  Same float arithmetic on the same float means in the same cell order: the same lists.
            // (Measured and dropped: the wave staging the 5 x 5 x 5 cells around its first point's cell in LDS and the
            // lanes reading their 27 from there -- 136 against 109 us per pass, profiles/r05_experiments.md.)
            const int ta = ci - dense.i0, tb = cj - dense.j0, tc = ck - dense.k0;
            if (ta >= 1 && tb >= 1 && tc >= 1 && ta <= dense.nx - 2 && tb <= dense.ny - 2 && tc <= dense.nz - 2) {
                // (a plane of nine cells at a time: all 27 in flight cost the registers of the third wave per SIMD)
                constexpr int kBatch = 9;
#pragma unroll
                for (int b0 = 0; b0 < 27; b0 += kBatch) {
                    float4 cb[kBatch];
#pragma unroll
                    for (int r = 0; r < kBatch / 3; ++r) {
                        const int row_id = b0 / 3 + r;
                        const float4 *row =
                            cells4 + (((size_t) (tc + row_id / 3 - 1) * dense.ny + (tb + row_id % 3 - 1)) * dense.nx + (ta - 1));
                        cb[r * 3 + 0] = row[0];
                        cb[r * 3 + 1] = row[1];
                        cb[r * 3 + 2] = row[2];
                    }
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const float fx = __fsub_rn(xt0, cb[u].x), fy = __fsub_rn(xt1, cb[u].y), fz = __fsub_rn(xt2, cb[u].z);
                        const float dd = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));
                        // (stored whether it passes or not, kept only if it does: no branch, no exec mask to set
                        // up and restore per cell -- the next store lands on a slot that did not pass)
                        s_near[n_near * kBlock + threadIdx.x] = __float_as_uint(cb[u].w);
                        n_near += dd <= A.res2_f ? 1 : 0;
                    }
                }
            }
        } else if (dense.table) {  // uniform
            const int ta = ci - dense.i0, tb = cj - dense.j0, tc = ck - dense.k0;
            if (ta >= 1 && tb >= 1 && tc >= 1 && ta <= dense.nx - 2 && tb <= dense.ny - 2 && tc <= dense.nz - 2) {
                Int3 rows[9];
#pragma unroll
                for (int row = 0; row < 9; ++row) {  // (dk, dj); the three di cells are adjacent in x
                    const int dj = row % 3 - 1, dk = row / 3 - 1;
                    rows[row] = *(const Int3 *) (dense.table +
                                                 (((size_t) (tc + dk) * dense.ny + (tb + dj)) * dense.nx + (ta - 1)));
                }
#pragma unroll
                for (int row = 0; row < 9; ++row) {
                    const int v3[3] = {rows[row].a, rows[row].b, rows[row].c};
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        if (v3[d] != -1) {
                            s_near[n_cand * kBlock + threadIdx.x] = (unsigned) v3[d];
                            ++n_cand;
                        }
                }
            }
        } else {
#pragma unroll 1
            for (int nb = 0; nb < 27; ++nb) {
                const int di = nb % 3 - 1, dj = (nb / 3) % 3 - 1, dk = nb / 9 - 1;
                const unsigned long long key = ndt_key(ck + dk, cj + dj, ci + di);
                unsigned hpos = ndt_hash(key) & mask;
                for (;;) {
                    const unsigned long long hk = hkeys[hpos];
                    if (hk == key) {
                        s_near[n_cand * kBlock + threadIdx.x] = hvals[hpos];
                        ++n_cand;
                        break;
                    }
                    if (hk == kEmptyKey) break;
                    hpos = (hpos + 1) & mask;
                }
            }
        }
        // 1b: kd-tree radius test in float on the float means (one 16-byte load each), four
        // candidates per trip so that four loads are in flight; survivors are compacted in place
        // (the write index never passes the read index), still in cell order
#pragma unroll 1
        for (int r = 0; r < n_cand; r += 4) {
            unsigned cv[4];
            float4 cm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = s_near[min(r + u, n_cand - 1) * kBlock + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 4; ++u) cm[u] = meanf[cv[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float fx = __fsub_rn(xt0, cm[u].x), fy = __fsub_rn(xt1, cm[u].y), fz = __fsub_rn(xt2, cm[u].z);
                const float dd = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));
                if (r + u < n_cand && dd <= A.res2_f) {  // <=> (double) dd < A.res2
                    s_near[n_near * kBlock + threadIdx.x] = cv[u];
                    ++n_near;
                }
            }
        }
        // the next voxel's record (mean + inverse covariance, 96 B) is requested before the
        // current one is evaluated: with two waves per SIMD a pass-2 trip would otherwise start
        // with a full memory round trip that nothing hides
        // (PRE = false frees the 24 registers of the record in flight: the Hessian variants then fit three waves
        // per SIMD instead of two.  Round 4 measured that equal -- with 1024 workgroups, a round and a third of
        // the 768 that are resident at three waves; with a grid of exactly one round it is 87 against 100 us per pass)
        constexpr bool PRE = !HESS;
        NdtVoxel vn;
        if (PRE) {
            const unsigned v0 = n_near > 0 ? s_near[threadIdx.x] : 0u;
#pragma unroll
            for (int k = 0; k < 3; ++k) vn.mean[k] = vox[v0].mean[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) vn.icov[k] = vox[v0].icov[k];
        }
        // Pass 2.  PCL adds, per (point, voxel) pair, w (-d2 (J_i^T cx)(cx^T J_j) + J_j^T C J_i + cx . h_ij) to
        // Hessian entry (i, j) and w cx . J_i to gradient entry i (cx = C (x' - mean), C the voxel's inverse
        // covariance, w = d1 d2 exp(-d2 q / 2)).  The point Jacobian J and the second derivatives h depend on
        // the POINT only, so the pair loop gathers just
        //     g3 = sum w cx,   P = sum w cx cx^T (symmetric),   Q = sum w C
        // and the point's contribution follows once, after its voxels:
        //     gradient = J^T g3,   Hessian = J^T S J + [g3 . h_ij],   S = -d2 P + Q^T
        // -- ~35 fused multiply-adds and an exp per pair instead of ~170, the 6 x 6 algebra once per point instead
        // of once per pair, and the 28 running sums out of the inner loop's registers.  Same terms, grouped
        // by point: the sums differ from the pair-by-pair ones in rounding order only.
        double g3[3] = {0.0, 0.0, 0.0}, P[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double Q[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        bool any = false;
#pragma unroll 1
        for (int t = 0; t < n_near; ++t) {
            NdtVoxel v;
            if (PRE) {
                v = vn;
                const unsigned v1 = t + 1 < n_near ? s_near[(t + 1) * kBlock + threadIdx.x] : 0u;
#pragma unroll
                for (int k = 0; k < 3; ++k) vn.mean[k] = vox[v1].mean[k];
#pragma unroll
                for (int k = 0; k < 9; ++k) vn.icov[k] = vox[v1].icov[k];
            } else {
                const unsigned v1 = s_near[t * kBlock + threadIdx.x];
#pragma unroll
                for (int k = 0; k < 3; ++k) v.mean[k] = vox[v1].mean[k];
#pragma unroll
                for (int k = 0; k < 9; ++k) v.icov[k] = vox[v1].icov[k];
            }
            const double xx[3] = {(double) xt0 - v.mean[0], (double) xt1 - v.mean[1], (double) xt2 - v.mean[2]};
            double cx[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) cx[a] = v.icov[a * 3] * xx[0] + v.icov[a * 3 + 1] * xx[1] + v.icov[a * 3 + 2] * xx[2];
            const double q = dot3d(xx, cx);
            const double e = exp(-A.d2 * q / 2.0);
            double w = A.d2 * e;
            if (w > 1 || w < 0 || w != w) continue;
            acc[0] += -A.d1 * e;
            w *= A.d1;
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
            // point Jacobian (3x6), computePointDerivatives: Jc[c][b] = J(b, 3 + c); J(0, 3) = 0 and columns 0..2
            // are the identity
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
                acc[5] += dot3d(Jc[1], g3);
                acc[6] += dot3d(Jc[2], g3);
            }
            if (HESS) {
                // S = -d2 P + Q^T
                const double md2 = -A.d2;
                double S[3][3];
                S[0][0] = md2 * P[0] + Q[0];
                S[0][1] = md2 * P[1] + Q[3];
                S[0][2] = md2 * P[2] + Q[6];
                S[1][0] = md2 * P[1] + Q[1];
                S[1][1] = md2 * P[3] + Q[4];
                S[1][2] = md2 * P[4] + Q[7];
                S[2][0] = md2 * P[2] + Q[2];
                S[2][1] = md2 * P[4] + Q[5];
                S[2][2] = md2 * P[5] + Q[8];
                // rows 0..2: S itself (columns 0..2) and S J_c (columns 3..5)
                double SJ[3][3];  // SJ[a][c] = sum_b S[a][b] Jc[c][b]
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    SJ[a][0] = S[a][1] * Jc[0][1] + S[a][2] * Jc[0][2];
                    SJ[a][1] = S[a][0] * Jc[1][0] + S[a][1] * Jc[1][1] + S[a][2] * Jc[1][2];
                    SJ[a][2] = S[a][0] * Jc[2][0] + S[a][1] * Jc[2][1] + S[a][2] * Jc[2][2];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = i; j < 3; ++j) acc[ndt_tri(i, j)] += S[i][j];
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[ndt_tri(i, 3 + c)] += SJ[i][c];
                }
                // rows 3..5: J_ci^T S J_cj + g3 . h(ci, cj); the second derivatives a, b, c / b, d, e / c, e, f
                // (computePointDerivatives; a, b, c have no x component) are formed where they are used: three
                // values live at a time, not eighteen
                // h vectors by block: a = (0, x.h0, x.h1), b = (0, x.h2, x.h3), c = (0, x.h4, x.h5),
                // d = x.h[6..8], e = x.h[9..11], f = x.h[12..14]
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int cj = ci; cj < 3; ++cj) {
                        const int sel = (ci == 0 && cj == 0) ? 0 : ((ci + cj == 1) ? 1 : ((ci + cj == 2 && ci != cj) ? 2 : ((ci == 1 && cj == 1) ? 3 : ((ci + cj == 3) ? 4 : 5))));
                        double t = Jc[ci][1] * SJ[1][cj] + Jc[ci][2] * SJ[2][cj];
                        if (ci > 0) t += Jc[ci][0] * SJ[0][cj];
                        if (sel < 3) {
                            t += g3[1] * jh_dot(x, 8 + 2 * sel, z0) + g3[2] * jh_dot(x, 8 + 2 * sel + 1, z0);
                        } else {
                            const int k0 = 8 + 6 + 3 * (sel - 3);
                            t += g3[0] * jh_dot(x, k0, z0) + g3[1] * jh_dot(x, k0 + 1, z0) + g3[2] * jh_dot(x, k0 + 2, z0);
                        }
                        acc[ndt_tri(3 + ci, 3 + cj)] += t;
                    }
            }
        }
    }
    // fixed-order reduction: wave xor-tree, then the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    __shared__ double lds[kBlock / 64][NA];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NA; ++k) lds[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < NA) {
        double s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += lds[w][threadIdx.x];
        // (written through to where every XCD sees it -- an agent-scope store --, for the workgroup that adds the rows)
        __hip_atomic_store(&partials[(size_t) blockIdx.x * NA + threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!F.ticket) return;  // (the rows are added by a launch of their own: the sharded passes)
    // The workgroups add the rows themselves and the LAST one hands the sums to the host -- what k_sum_fetch did in a
    // launch of its own: a dependent launch (~6 us of dispatch) and 5 us of one workgroup behind every pass.  Two levels
    // (one workgroup adding all 768 rows is a chain of dependent L2 round trips): the last of every kFetchGroup
    // consecutive workgroups adds that group's rows into one, the last of those adds the groups' rows.  Groups and
    // orders are fixed by the workgroup numbers, not by who finishes when: the sums do not depend on it.
    // No fence anywhere (an agent-scope release writes the XCD's whole L2 back, a system-scope one too: with
    // __threadfence() around the tickets a pass took 157 us instead of 89): rows are stored write-through, a workgroup
    // waits for its own stores (s_waitcnt), draws its ticket with a relaxed atomic, the last one reads the rows at
    // agent scope, and the sums reach the host as 16-byte slots {value, number of the pass} in ONE store each --
    // the host takes a slot once it carries the number it waits for (the recipe of the GICP evaluator, wm_gicp.hip).
    constexpr unsigned kLanes = (unsigned) kBlock / (unsigned) NA;  // row-lanes of NA columns
    __shared__ double s1[kLanes][NA];
    __shared__ unsigned s_last;
    const unsigned t = threadIdx.x, c = t % (unsigned) NA, g = t / (unsigned) NA;
    const unsigned ngroups = (gridDim.x + kFetchGroup - 1u) / kFetchGroup, grp = blockIdx.x / kFetchGroup;
    const unsigned members = min(kFetchGroup, gridDim.x - grp * kFetchGroup);
    double *grows = partials + (size_t) gridDim.x * NA;  // the groups' rows, behind the workgroups'
    if (t < 64u) {  // (the row's writers are lanes of wave 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) s_last = __hip_atomic_fetch_add(F.ticket + 1u + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    auto add_rows = [&](const double *rows, unsigned n) -> double {  // (thread t < NA returns rows[0][t] + rows[1][t] + ...; fixed order)
        if (g < kLanes) {
            double v[(kFetchRowsMax + kLanes - 1u) / kLanes];
#pragma unroll
            for (unsigned u = 0; u < (kFetchRowsMax + kLanes - 1u) / kLanes; ++u) {
                const unsigned r = g + u * kLanes;
                v[u] = r < n ? __hip_atomic_load(&rows[(size_t) r * NA + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
            double a = 0.0;
#pragma unroll
            for (unsigned u = 0; u < (kFetchRowsMax + kLanes - 1u) / kLanes; ++u) a += v[u];
            s1[g][c] = a;
        }
        __syncthreads();
        double r = 0.0;
        if (t < (unsigned) NA)
            for (unsigned gg = 0; gg < kLanes; ++gg) r += s1[gg][t];
        return r;
    };
    {
        const double r = add_rows(partials + (size_t) grp * kFetchGroup * NA, members);
        if (t < (unsigned) NA) __hip_atomic_store(&grows[(size_t) grp * NA + t], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // (s1 is used again below; and wave 0 draws the next ticket behind its own stores)
    if (t < 64u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) {
            __hip_atomic_store(F.ticket + 1u + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next pass)
            s_last = __hip_atomic_fetch_add(F.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    {
        const double r = add_rows(grows, ngroups);
        if (t < (unsigned) NA) {
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            const unsigned long long rb = (unsigned long long) __double_as_longlong(r);
            const u4v out = {(unsigned) rb, (unsigned) (rb >> 32), F.seq, 0u};
            u4v *dst = reinterpret_cast<u4v *>(F.dst) + t;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(out) : "memory");
        }
        if (t == 0) __hip_atomic_store(F.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ----------------------------------------------------------------- host
struct NdtModel {
    double res = -1;
    size_t n_target = 0;
    unsigned nvox = 0, nvalid = 0, hmask = 0;
};

template <class KT>
static int ndt_build_t(wm_ctx *ctx, double res) {
    const size_t n = ctx->n_tgt_input;
    const float4 *pts = ctx->tgt_orig.as<float4>();
    const unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    WM_HIP(ctx, ctx->ndt_keys.reserve(n * 8));
    WM_HIP(ctx, ctx->ndt_keys2.reserve(n * 8));
    // (scratch of its own, not the voxel filter's / the Morton sort's vg_*: the source's sort may be running on the
    // side stream while this model is built, wm_ndt_align)
    WM_HIP(ctx, ctx->ndt_perm.reserve((n + 1) * 4));  // later: the voxels' head positions (+ end)
    WM_HIP(ctx, ctx->ndt_perm2.reserve(n * 4));
    WM_HIP(ctx, ctx->ndt_flags.reserve(n * 4));
    WM_HIP(ctx, ctx->ndt_seg.reserve((n + 1) * 4));
    KT *k1 = ctx->ndt_keys.as<KT>(), *k2 = ctx->ndt_keys2.as<KT>();
    unsigned *p1 = ctx->ndt_perm.as<unsigned>(), *p2 = ctx->ndt_perm2.as<unsigned>();
    unsigned *flags = ctx->ndt_flags.as<unsigned>(), *seg = ctx->ndt_seg.as<unsigned>();
    // the lattice of voxels over the target's bounding box (cell numbers from the same float
    // product as the keys)
    NdtLattice L{};
    unsigned long long dimv[3];
    {
        const float inv = 1.0f / (float) res;
        const Bbox &bb = ctx->tgt_bbox;
        for (int d = 0; d < 3; ++d) {
            L.lo[d] = (int) floorf(bb.lo[d] * inv);
            const long long dd = (long long) floorf(bb.hi[d] * inv) - L.lo[d] + 1;
            if (ctx->n_tgt > 0 && (dd < 1 || dd > (1ll << 20))) {  // ndt_key holds 21 bits per axis
                ctx->last_error = "NDT: the target spans more than 2^20 voxels along an axis";
                return WM_ERR_ARG;
            }
            dimv[d] = ctx->n_tgt > 0 ? (unsigned long long) dd : 1ull;
        }
        L.nx = dimv[0];
        L.ny = dimv[1];
        L.cells = dimv[0] * dimv[1] * dimv[2];
    }
    unsigned key_bits = 1;
    while (key_bits < 64 && (L.cells >> key_bits) != 0ull) ++key_bits;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_key<KT>), dim3(blocks), dim3(kBlock), 0, ctx->stream, pts, (unsigned) n,
                       1.0f / (float) res, L, k1, p1);
    size_t tmp = 0;
    WM_HIP(ctx, sort_pairs_low_bits(nullptr, tmp, k1, k2, p1, p2, n, key_bits, ctx->stream,
                                    (size_t) ctx->tune_radix_min, ctx->tune_sort));
    WM_HIP(ctx, ctx->ndt_tmp.reserve(tmp));
    WM_HIP(ctx, sort_pairs_low_bits(ctx->ndt_tmp.p, tmp, k1, k2, p1, p2, n, key_bits, ctx->stream,
                                    (size_t) ctx->tune_radix_min, ctx->tune_sort));
    // (a source's Morton sort that wm_ndt_align held back goes to the side stream NOW: behind this model's key kernel and
    // radix sort -- the registration's critical path, ~150 us of device time that this thread's ~15 launches of the other
    // sort fit into -- and before the launches below, the last of which this thread then waits for)
    WM_TRY(enqueue_deferred_sort(ctx));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_flags<KT>), dim3(blocks), dim3(kBlock), 0, ctx->stream, (const KT *) k2, (unsigned) n, (KT) L.cells,
                       flags);
    WM_TRY(exclusive_scan(ctx, flags, n, seg));
    unsigned *h_word = (unsigned *) pinned_scratch(ctx, 0);
    if (!h_word) return WM_ERR_HIP;
    WM_TRY(fast_fetch(ctx, h_word, seg + n, 4));  // the scan's total = number of voxels
    const unsigned nvox = *h_word;
    ctx->ndt_nvox = nvox;
    ctx->ndt_nvalid = 0;
    unsigned cap = 16;
    while (cap < 2 * nvox + 2) cap <<= 1;
    ctx->ndt_hmask = cap - 1;
    WM_HIP(ctx, ctx->ndt_vox.reserve((size_t) (nvox > 0 ? nvox : 1) * sizeof(NdtVoxel)));
    WM_HIP(ctx, ctx->ndt_vkey.reserve((size_t) (nvox > 0 ? nvox : 1) * 8 + 8));
    WM_HIP(ctx, ctx->ndt_meanf.reserve((size_t) (nvox > 0 ? nvox : 1) * sizeof(float4)));
    WM_HIP(ctx, ctx->ndt_hkeys.reserve((size_t) cap * 8));
    WM_HIP(ctx, ctx->ndt_hvals.reserve((size_t) cap * 4));
    WM_HIP(ctx, hipMemsetAsync(ctx->ndt_hkeys.p, 0xFF, (size_t) cap * 8, ctx->stream));
    unsigned *d_nvalid = ctx->bbox_buf.as<unsigned>();
    WM_HIP(ctx, ctx->bbox_buf.reserve(64));
    d_nvalid = ctx->bbox_buf.as<unsigned>();
    WM_HIP(ctx, hipMemsetAsync(d_nvalid, 0, 4, ctx->stream));
    if (nvox > 0) {
        unsigned *heads = p1;  // the sort's input permutation is dead by now
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_heads<KT>), dim3(blocks), dim3(kBlock), 0, ctx->stream, (const KT *) k2, (const unsigned *) flags, (const unsigned *) seg, (unsigned) n,
                           (KT) L.cells, heads);
        // Coarse grids: every voxel a wave's.  Finer ones: a lane's -- except their crowded voxels: a lidar's
        // rings put thousands of points into the voxels next to the sensor (1 700 in a 0.5 m voxel of a
        // 2M-point 64-ring scan whose average is 100), and one lane walking those alone held the whole launch
        // back (552 us; both routines form the same sums in the same order, so who takes a voxel does not
        // change the model).
        const bool coarse = (unsigned long long) n > (unsigned long long) kVoxWaveAvg * nvox;
        const unsigned split = coarse ? 0u : ctx->tune_ndt_vox_split >= 0 ? (unsigned) ctx->tune_ndt_vox_split : (nvox <= kVoxFew ? kVoxLaneMaxFew : kVoxLaneMax);
        WM_HIP(ctx, ctx->ndt_vsum.reserve((size_t) nvox * 12 * sizeof(double)));
        hipLaunchKernelGGL(k_ndt_voxel_sums, dim3(nvox < 8192u ? (nvox + 3u) / 4u : 2048u), dim3(kBlock), 0, ctx->stream, pts,
                           p2, heads, nvox, split + 1u, ctx->ndt_vsum.as<double>());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_voxel_stats<KT>), dim3((nvox + kVoxStatBlock - 1) / kVoxStatBlock), dim3(kVoxStatBlock), 0,
                           ctx->stream, pts, (const KT *) k2, (const unsigned *) p2, (const unsigned *) heads, nvox, split, (const double *) ctx->ndt_vsum.as<double>(), L,
                           ctx->ndt_vox.as<NdtVoxel>(), ctx->ndt_meanf.as<float4>(),
                           ctx->ndt_vkey.as<unsigned long long>(), d_nvalid);
        hipLaunchKernelGGL(k_ndt_hash_insert, dim3((nvox + kBlock - 1) / kBlock), dim3(kBlock), 0,
                           ctx->stream, ctx->ndt_vkey.as<unsigned long long>(), nvox,
                           ctx->ndt_hkeys.as<unsigned long long>(), ctx->ndt_hvals.as<unsigned>(),
                           ctx->ndt_hmask);
        WM_HIP(ctx, hipGetLastError());
    }
    // dense cell -> slot table over the target's bounding box, when that lattice is small
    ctx->ndt_dense_on = false;
    if (nvox > 0) {
        const float inv = 1.0f / (float) res;
        const Bbox &bb = ctx->tgt_bbox;
        int lo[3], dim[3];
        int64_t cells = 1;
        for (int d = 0; d < 3; ++d) {
            // two empty cells of margin on every side: a query whose own cell lies within one
            // cell of the occupied box reads its 3x3x3 block without a bounds test, and one
            // further out has no neighbours at all
            lo[d] = (int) floorf(bb.lo[d] * inv) - 2;  // the same float product as k_ndt_key
            dim[d] = (int) floorf(bb.hi[d] * inv) - lo[d] + 3;
            cells *= dim[d] > 0 ? dim[d] : 1;
        }
        if (cells <= (int64_t) 32 << 20) {
            WM_HIP(ctx, ctx->ndt_dense.reserve((size_t) cells * 4));
            WM_HIP(ctx, hipMemsetAsync(ctx->ndt_dense.p, 0xFF, (size_t) cells * 4, ctx->stream));
            const NdtDense d{ctx->ndt_dense.as<int>(), lo[0], lo[1], lo[2], dim[0], dim[1], dim[2]};
            // ... and, while that lattice is at most 4 M cells (64 MB), the same with the voxel's float mean in the cell
            // (k_ndt_derivs: the cells are the radius tests); every byte 0x7f = a "mean" of 3.4e38 in an empty cell
            ctx->ndt_cells4_on = false;
            float4 *cells4 = nullptr;
            if (cells <= (int64_t) 4 << 20 && ctx->tune_ndt_dense >= 2) {
                WM_HIP(ctx, ctx->ndt_cells4.reserve((size_t) cells * sizeof(float4)));
                WM_HIP(ctx, hipMemsetAsync(ctx->ndt_cells4.p, 0x7F, (size_t) cells * sizeof(float4), ctx->stream));
                cells4 = ctx->ndt_cells4.as<float4>();
                ctx->ndt_cells4_on = true;
            }
            hipLaunchKernelGGL(k_ndt_dense_fill, dim3((nvox + kBlock - 1) / kBlock), dim3(kBlock), 0,
                               ctx->stream, ctx->ndt_vkey.as<unsigned long long>(), nvox, d,
                               ctx->ndt_dense.as<int>(), ctx->ndt_meanf.as<float4>(), cells4);
            WM_HIP(ctx, hipGetLastError());
            for (int k = 0; k < 3; ++k) {
                ctx->ndt_dense_lo[k] = lo[k];
                ctx->ndt_dense_dim[k] = dim[k];
            }
            ctx->ndt_dense_on = true;
        }
    }
    WM_TRY(fast_fetch(ctx, h_word, d_nvalid, 4));
    ctx->ndt_nvalid = *h_word;
    ctx->ndt_res = res;
    ctx->ndt_built = true;
    return WM_OK;
}

// the voxel model of the target at resolution res (32-bit sort keys while the lattice over its bounding box has fewer than
// 2^32 cells -- every grid a scan's bounding box and a sane resolution give)
static int ndt_build(wm_ctx *ctx, double res) {
    const float inv = 1.0f / (float) res;
    const Bbox &bb = ctx->tgt_bbox;
    double cells = 1.0;
    for (int d = 0; d < 3; ++d) {
        const double dd = (double) floorf(bb.hi[d] * inv) - (double) floorf(bb.lo[d] * inv) + 1.0;
        cells *= dd > 1.0 ? dd : 1.0;
    }
    if (ctx->n_tgt > 0 && cells < 4294967295.0 && !ctx->tune_ndt_keys64) return ndt_build_t<unsigned>(ctx, res);
    return ndt_build_t<unsigned long long>(ctx, res);
}


struct NdtEval {
    wm_ctx *ctx;
    const wm_ndt_params *prm;
    double d1, d2;
    int evals = 0;
    float kernel_ms = 0;
    double host_launch_us = 0, host_wait_us = 0;  // WM_NDT_PROFILE=2: wall time inside the launch calls / the wait
    int ls_hist[12] = {0};  // line searches by their number of extra trials (developer: WM_TRACE)
    int rc = WM_OK;
    // (what ndt_align_loop / step_length_mt ask of an objective, wm_ndt_ctl.hpp)
    double eval(const double p[6], double *grad, double *hess);
    bool failed() const { return rc != WM_OK; }
    bool skip_line_search() const { return prm->skip_line_search != 0; }
    bool spec_hessian() const { return ctx->tune_ndt_spec_hessian != 0; }
    void note_line_search(int trials) { ls_hist[trials < 11 ? trials : 11]++; }
};

// score (+ gradient, + Hessian) at pose p; returns <0 on HIP error via *rc
static double ndt_eval(NdtEval &E, const double p[6], double *grad, double *hess, int *rc) {
    wm_ctx *ctx = E.ctx;
    NdtArgs A;
    float Tf[16];
    pose_to_matrix_f(p, Tf);
    for (int k = 0; k < 12; ++k) A.Tf[k] = Tf[k];
    A.inv_res = 1.0f / (float) E.prm->res;
    A.res2 = E.prm->res * E.prm->res;
    A.res2_f = threshold_d2_strict(E.prm->res);
    A.d1 = E.d1;
    A.d2 = E.d2;
    angle_derivatives(p, E.prm->pcl_d1_sign, &A);
    // this rank's share of the Morton-ordered source (all of it unless wm_ndt_set_shard): chunks
    // of 4096 points dealt out round-robin, so every rank sees the whole scene at 1/world density
    // (contiguous slices would give the ranks different neighbour counts, i.e. different times)
    const bool sharded = ctx->ndt_world > 1 && (ctx->ndt_reduce || ctx->ndt_comm);
    const unsigned n_total = (unsigned) ctx->n_src;
    const unsigned s_rank = sharded ? (unsigned) ctx->ndt_rank : 0u, s_world = sharded ? (unsigned) ctx->ndt_world : 1u;
    unsigned n = n_total;
    if (sharded) {
        const unsigned chunks = (n_total + (1u << kNdtChunkLog2) - 1u) >> kNdtChunkLog2;
        const unsigned mine = chunks > s_rank ? (chunks - s_rank + s_world - 1u) / s_world : 0u;
        n = mine << kNdtChunkLog2;
    }
    int nb = (int) ((n + kBlock - 1) / kBlock);
    // one full round of resident workgroups: the Hessian variants run three waves per SIMD (three workgroups per compute
    // unit), the gradient variant four -- 768 / 1024 on the 256 compute units of an MI355X (1024 for all, round 4, made
    // the Hessian passes a round and a third: 102 against 87 us)
    if (ctx->ndt_cus == 0) {
        int cus = 0;
        ctx->ndt_cus = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) ? cus : 256;
    }
    const int nb_cap = ctx->tune_ndt_blocks > 0 ? ctx->tune_ndt_blocks : ctx->ndt_cus * (hess ? 3 : 4);
    if (nb > nb_cap) nb = nb_cap;
    if (nb > kNdtBlocks) nb = kNdtBlocks;
    if (nb < 1) nb = 1;
    double *partials = ctx->partials.as<double>();
    const NdtVoxel *vox = ctx->ndt_vox.as<NdtVoxel>();
    const unsigned long long *hk = ctx->ndt_hkeys.as<unsigned long long>();
    const unsigned *hv = ctx->ndt_hvals.as<unsigned>();
    const float4 *src = ctx->src_sorted.as<float4>();
    NdtDense dense{nullptr, 0, 0, 0, 0, 0, 0};
    if (ctx->ndt_dense_on && ctx->tune_ndt_dense)
        dense = NdtDense{ctx->ndt_dense.as<int>(), ctx->ndt_dense_lo[0], ctx->ndt_dense_lo[1],
                         ctx->ndt_dense_lo[2], ctx->ndt_dense_dim[0], ctx->ndt_dense_dim[1],
                         ctx->ndt_dense_dim[2]};
    const float4 *cells4 = (dense.table && ctx->ndt_cells4_on && ctx->tune_ndt_dense >= 2) ? ctx->ndt_cells4.as<float4>() : nullptr;
    // the sums come back from the pass's own last workgroup (not when the ranks' sums are all-reduced on the device first)
    NdtFetch F{nullptr, nullptr, 0u};
    const bool fused_fetch = !(sharded && ctx->ndt_comm) && ctx->tune_ndt_fused_fetch &&
                             (unsigned) nb <= kFetchGroup * kFetchRowsMax;  // (two levels of at most 16 and 64 rows)
    // (pinned: [0, 64) the sums as k_sum_fetch leaves them; [64, 128) the fused hand-over's 16-byte slots, zeroed once)
    if (!ctx->h_ndt && (hipHostMalloc((void **) &ctx->h_ndt, 128 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
                        !memset(ctx->h_ndt, 0, 128 * sizeof(double)))) {
        ctx->last_error = "ndt_eval: pinned allocation failed";
        *rc = WM_ERR_HIP;
        return 0;
    }
    if (fused_fetch) {
        if (!ctx->ndt_ticket.p) {
            if (ctx->ndt_ticket.reserve(4096) != hipSuccess || hipMemsetAsync(ctx->ndt_ticket.p, 0, 4096, ctx->stream) != hipSuccess) {
                *rc = WM_ERR_HIP;
                return 0;
            }
        }
        if (++ctx->ndt_seq == 0u) ctx->ndt_seq = 1u;
        F = NdtFetch{ctx->ndt_ticket.as<unsigned>(), ctx->h_ndt + 64, ctx->ndt_seq};
    }
    const auto t_launch0 = std::chrono::steady_clock::now();
    if (ctx->ndt_profile) (void) hipEventRecord(ctx->ev_a, ctx->stream);
    if (hess && grad)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_derivs<true, true>), dim3(nb), dim3(kBlock), 0,
                           ctx->stream, src, n, n_total, s_rank, s_world, vox, ctx->ndt_meanf.as<float4>(), hk,
                           hv, ctx->ndt_hmask, dense, cells4, A,
                           partials, F);
    else if (grad)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_derivs<true, false>), dim3(nb), dim3(kBlock), 0,
                           ctx->stream, src, n, n_total, s_rank, s_world, vox, ctx->ndt_meanf.as<float4>(), hk,
                           hv, ctx->ndt_hmask, dense, cells4, A,
                           partials, F);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ndt_derivs<false, true>), dim3(nb), dim3(kBlock), 0,
                           ctx->stream, src, n, n_total, s_rank, s_world, vox, ctx->ndt_meanf.as<float4>(), hk,
                           hv, ctx->ndt_hmask, dense, cells4, A,
                           partials, F);
    if (ctx->ndt_profile) (void) hipEventRecord(ctx->ev_b, ctx->stream);
    const auto t_launch1 = std::chrono::steady_clock::now();
    // sums over blocks, formed on the device (fixed order); 224 bytes come back
    const int n_acc = hess ? kNdtAcc : kNdtAccGrad;
    if (sharded && ctx->ndt_comm) {
        // sums over blocks -> device buffer -> all-reduce over the ranks on the stream (RCCL) -> host
        if (ctx->ndt_sum_dev.reserve(64 * sizeof(double)) != hipSuccess ||
            sum_to_device(ctx, ctx->ndt_sum_dev.as<double>(), partials, (unsigned) nb, (unsigned) n_acc) != WM_OK ||
            comm_allreduce(ctx, ctx->ndt_comm, ctx->ndt_sum_dev.as<double>(), n_acc) != WM_OK ||
            fast_fetch(ctx, ctx->h_ndt, ctx->ndt_sum_dev.p, (size_t) n_acc * sizeof(double)) != WM_OK) {
            if (ctx->last_error.empty()) ctx->last_error = "ndt_eval: device all-reduce failed";
            *rc = WM_ERR_HIP;
            return 0;
        }
    } else if ((fused_fetch ? wait_slots(ctx, ctx->h_ndt + 64, n_acc, F.seq)
                            : fast_fetch_sum(ctx, ctx->h_ndt, partials, (unsigned) nb, (unsigned) n_acc)) != WM_OK) {
        ctx->last_error = "ndt_eval: HIP error";
        *rc = WM_ERR_HIP;
        return 0;
    }
    const auto t_wait1 = std::chrono::steady_clock::now();
    E.host_launch_us += std::chrono::duration<double, std::micro>(t_launch1 - t_launch0).count();
    E.host_wait_us += std::chrono::duration<double, std::micro>(t_wait1 - t_launch1).count();
    float ms = 0;
    if (ctx->ndt_profile) {  // event timing needs the stream drained; off unless asked for
        (void) hipStreamSynchronize(ctx->stream);
        (void) hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
    }
    E.kernel_ms += ms;
    E.evals += 1;
    double a[kNdtAcc] = {0};
    for (int k = 0; k < n_acc; ++k) a[k] = fused_fetch ? ctx->h_ndt[64 + 2 * k] : ctx->h_ndt[k];  // (slots of 16 bytes: the value first)
    if (sharded && !ctx->ndt_comm && ctx->ndt_reduce(a, n_acc, ctx->ndt_reduce_user) != 0) {
        ctx->last_error = "ndt_eval: the all-reduce callback failed";
        *rc = WM_ERR_STATE;
        return 0;
    }
    if (grad)
        for (int k = 0; k < 6; ++k) grad[k] = a[1 + k];
    if (hess) {
        // PCL fills all 36 entries; H(j,i) differs from H(i,j) only in the rounding of two
        // commuted products (~1e-16 relative).  The kernel accumulates the upper triangle -- 21
        // instead of 36 f64 accumulators and 40 % less arithmetic in its dominant loop -- and
        // the lower one is its mirror.
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) hess[i * 6 + j] = hess[j * 6 + i] = a[ndt_tri(i, j)];
    }
    return a[0];
}

double NdtEval::eval(const double p[6], double *grad, double *hess) { return ndt_eval(*this, p, grad, hess, &rc); }


}  // namespace wm

using namespace wm;

extern "C" {

void wm_ndt_default_params(wm_ndt_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->res = 5;         // ndt.hpp:40
    p->step_size = 3;   // ndt.hpp:37 (an int in the reference)
    p->t_eps = 1e-8;    // ndt.hpp:39
    p->max_iter = 100;  // ndt.hpp:38
    p->outlier_ratio = 0.55;
    p->skip_line_search = 0;
    p->pcl_d1_sign = 1;
}

int wm_ndt_set_shard(wm_ctx *ctx, int rank, int world, wm_allreduce_fn reduce, void *user) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return WM_ERR_ARG;
    ctx->ndt_rank = rank;
    ctx->ndt_world = world;
    ctx->ndt_reduce = reduce;
    ctx->ndt_reduce_user = user;
    return WM_OK;
}

// pcl::NormalDistributionsTransform::setInputTarget builds the voxel grid at once (ndt.cpp:55):
// later align() calls on the same target reuse it.
int wm_ndt_build_model(wm_ctx *ctx, double res) {
    if (!ctx || !(res > 0)) return WM_ERR_ARG;
    if (ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx));
    if (!ctx->ndt_built || ctx->ndt_res != res) {
        WM_TRY(ndt_build(ctx, res));
        ctx->ndt_model_builds++;
    }
    return WM_OK;
}

int wm_ndt_set_comm(wm_ctx *ctx, struct wm_comm *comm) {
    if (!ctx) return WM_ERR_ARG;
    ctx->ndt_comm = comm;
    ctx->ndt_reduce = nullptr;
    ctx->ndt_reduce_user = nullptr;
    ctx->ndt_rank = comm ? wm_comm_rank(comm) : 0;
    ctx->ndt_world = comm ? wm_comm_world(comm) : 1;
    return WM_OK;
}

int wm_ndt_align(wm_ctx *ctx, const wm_ndt_params *prm, double T_out[16], wm_ndt_stats *stats) {
    if (!ctx || !prm || !T_out || !(prm->res > 0) || !(prm->step_size > 0)) return WM_ERR_ARG;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    // a new source's Morton sort and a new target's voxel model are independent chains of small launches (230 + 370 us
    // at 2M points, one behind the other on one stream in rounds 1-4): the sort goes to the side stream, the model is
    // built on this one meanwhile, and the passes wait for both.  The MODEL's chain is the longer one and has a wait for
    // the device in its middle (the number of voxels): its first part is enqueued first, the sort right before that wait
    // (finalize_clouds mode 2 + ndt_build's enqueue_deferred_sort; with the sort's ~15 launches enqueued first, round
    // 5's order, the model's first kernel started 210 us into the call: this thread was still enqueueing the sort)
    const bool will_build = !ctx->ndt_built || ctx->ndt_res != prm->res;
    WM_TRY(finalize_clouds(ctx, -1.0, 0, will_build ? 2 : 0));
    // (from here on the source's sort may be running on the side stream: whatever ends this call early joins it
    // first -- a later wm_set_source / pack on the main stream must not overwrite what the sort still reads)
    struct JoinOnExit {
        wm_ctx *c;
        ~JoinOnExit() { (void) join_source_sort(c); }
    } join_on_exit{ctx};
    WM_HIP(ctx, ctx->partials.reserve((size_t) (kNdtBlocks + 256) * kNdtAcc * sizeof(double)));
    if (!ctx->ndt_built || ctx->ndt_res != prm->res) {
        WM_TRY(ndt_build(ctx, prm->res));
        ctx->ndt_model_builds++;
    }
    WM_TRY(enqueue_deferred_sort(ctx));  // (ndt_build has done it)
    WM_TRY(join_source_sort(ctx));
    NdtEval E;
    E.ctx = ctx;
    E.prm = prm;
    {
        const double c1 = 10.0 * (1.0 - prm->outlier_ratio), c2 = prm->outlier_ratio / pow(prm->res, 3);
        const double d3 = -log(c2);
        E.d1 = -log(c1 + c2) - d3;
        E.d2 = -2.0 * log((-log(c1 * exp(-0.5) + c2) - d3) / E.d1);
    }
    NdtLoopOut lo;
    ndt_align_loop(E, prm->step_size, prm->t_eps, prm->max_iter, prm->force_iterations, &lo);
    if (E.rc != WM_OK) return E.rc;
    const double *p = lo.p;
    const double score = lo.score;
    const int iter = lo.iterations;
    const bool converged = lo.converged;
    if (stats) {
        stats->converged = converged;
        stats->iterations = iter;
        stats->n_voxels = (int) ctx->ndt_nvalid;
        stats->model_builds = ctx->ndt_model_builds;
        stats->evaluations = E.evals;
        stats->score = ctx->n_src > 0 ? score / (double) ctx->n_src_input : 0;
        stats->deriv_kernel_ms = E.kernel_ms;
        if (ctx->trace)
            fprintf(stderr, "[wm] ndt: %d passes, host time in launches %.0f us, in waits %.0f us, kernels %.3f ms\n",
                    E.evals, E.host_launch_us, E.host_wait_us, E.kernel_ms);
        if (ctx->trace)
            fprintf(stderr, "[wm] ndt: line searches by extra trials 0..11+: %d %d %d %d %d %d %d %d %d %d %d %d\n", E.ls_hist[0],
                    E.ls_hist[1], E.ls_hist[2], E.ls_hist[3], E.ls_hist[4], E.ls_hist[5], E.ls_hist[6], E.ls_hist[7],
                    E.ls_hist[8], E.ls_hist[9], E.ls_hist[10], E.ls_hist[11]);
    }
    if (!converged) return WM_NOT_CONVERGED;
    float Tf[16];
    pose_to_matrix_f(p, Tf);
    for (int k = 0; k < 16; ++k) T_out[k] = (double) Tf[k];
    return WM_OK;
}

// score / gradient / Hessian at a given pose (kernel-level parity)
int wm_ndt_derivatives(wm_ctx *ctx, const wm_ndt_params *prm, const double pose[6], double *score,
                       double grad[6], double hess[36], int *n_voxels) {
    if (!ctx || !prm || !pose || !score || !(prm->res > 0)) return WM_ERR_ARG;
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx));
    WM_HIP(ctx, ctx->partials.reserve((size_t) (kNdtBlocks + 256) * kNdtAcc * sizeof(double)));
    if (!ctx->ndt_built || ctx->ndt_res != prm->res) {
        WM_TRY(ndt_build(ctx, prm->res));
        ctx->ndt_model_builds++;
    }
    NdtEval E;
    E.ctx = ctx;
    E.prm = prm;
    const double c1 = 10.0 * (1.0 - prm->outlier_ratio), c2 = prm->outlier_ratio / pow(prm->res, 3);
    const double d3 = -log(c2);
    E.d1 = -log(c1 + c2) - d3;
    E.d2 = -2.0 * log((-log(c1 * exp(-0.5) + c2) - d3) / E.d1);
    int rc = WM_OK;
    double g[6], h[36];
    *score = ndt_eval(E, pose, g, h, &rc);
    if (grad) memcpy(grad, g, sizeof(g));
    if (hess) memcpy(hess, h, sizeof(h));
    if (n_voxels) *n_voxels = (int) ctx->ndt_nvalid;
    return rc;
}

}  // extern "C"
