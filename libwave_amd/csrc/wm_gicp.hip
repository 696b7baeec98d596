// wm_gicp.hip -- pcl::GeneralizedIterativeClosestPoint on device, as libwave's
// GICPMatcher drives it (wave_matching/src/gicp.cpp:31-34 setters, :37-64
// setInput*/align):
//   k_gicp_cov    computeCovariances: exact k-NN of every point in its own cloud
//                 (certified box search on that cloud's grid, (d2, index) order), float
//                 products into double sums, 3x3 SVD, spectrum replaced by (1, 1, eps)
//   k_nn_grid     (wm_nn.hip) the per-iteration 1-NN with the strict d2 < 25 gate
//   k_gicp_mahal  M_i = (C2_j + R C1_i R^T)^-1 for every matched pair
//   k_gicp_fdf    OptimizationFunctorWithIndices::fdf: f, sum M r (3), sum p (M r)^T (9)
//                 -> 13 doubles per evaluation; the only thing the optimiser sees
//   k_gicp_quad   (opt-in, wm_gicp_params::objective = WM_GICP_OBJECTIVE_STATISTICS) the same objective as 74
//                 sufficient statistics of the pairs, formed ONCE per outer iteration: the ~40 evaluations of a
//                 minimisation are then scalar work on the host, no pass over the pairs (wm_gicp_quad.hpp);
//                 k_gicp_mahal / k_gicp_fdf / the resident evaluator serve WM_GICP_OBJECTIVE_PCL_SUMS
//   host          estimateRigidTransformationBFGS: pcl::BFGS (GSL vector_bfgs2 with
//                 Fletcher's line search), applyState, the outer delta test
// [PCL registration/impl/gicp.hpp, registration/bfgs.h]
#include "wm_internal.hpp"
#include "wm_gicp_dev.hpp"
#include "wm_bfgs.hpp"
#include "wm_gicp_quad.hpp"
#ifndef WM_COV_BLOCK
#define WM_COV_BLOCK 64
#endif

#include <float.h>
#include <stddef.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace wm {

constexpr int kGicpBlocksMax = 4096;  // partial rows per objective evaluation (ctx->tune_gicp_blocks)


// computeCovariances for every point of `qpts` (any order); neighbours come from grid g
// (built over the same cloud), coordinates are gathered from `orig` by caller index.
// ONE wave per workgroup, six waves per SIMD at PCL's k = 10 (80 registers, three of them spilled outside the loops): the search is a chain of
// dependent look-ups and gathers per wave (a wave issues for an eighth of its life), so what counts is how many waves are
// resident and how soon a finished one is replaced -- a 256-thread workgroup waits for four free slots, one per SIMD,
// i.e. for its predecessor's slowest wave.  231 -> 209 us per launch at 500k points, k = 10 (profiles/r06_experiments.md,
// where the two things that did NOT pay are logged too: dealing a workgroup's queries out to its waves by the size of
// their first box -- 20 % fewer wave trips, 2 % less time -- and regrouping the unsettled queries after the first pass).
constexpr int kCovBlock = WM_COV_BLOCK;  // queries (threads) of a workgroup of k_gicp_cov
template <int K>
__global__ void __launch_bounds__(kCovBlock) __attribute__((amdgpu_waves_per_eu(K <= 10 ? 6 : (K <= 12 ? 5 : 1))))
    k_gicp_cov(GridDev g, const float4 *__restrict__ qpts, unsigned n,
               const float4 *__restrict__ orig, int k, double eps, double *__restrict__ cov_out,
               int by_w, float r0_cells) {
    __shared__ uint2 s_runs[kKnnRows * kCovBlock];  // per-lane run lists of knn_search (lane-private)
    const unsigned i = blockIdx.x * kCovBlock + threadIdx.x;
    if (i >= n) return;
    const float4 q = qpts[i];
    const unsigned slot = (by_w & 1) ? __float_as_uint(q.w) : i;
    double *out = cov_out + (size_t) slot * 9;
    if (!(q.x == q.x)) {  // non-finite point: never matched; keep a defined value
#pragma unroll
        for (int a = 0; a < 9; ++a) out[a] = (a % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    unsigned long long best[K];
    // (developer timing experiment, WM_TUNE_COV_DBG -- WRONG results: 512 = no neighbour search, 256 = no SVD.
    // Measured at 500k points: 329 us as is, 314 without the SVD, 49 without the search, 27 without both:
    // the k-NN search is 85 % of this kernel)
    if (by_w & 512) {
#pragma unroll
        for (int j = 0; j < K; ++j) best[j] = (unsigned long long) min(i + (unsigned) j, n - 1u);
    } else
    knn_search<K>(g, q.x, q.y, q.z, k, r0_cells, best, s_runs, threadIdx.x, kCovBlock);
    gicp_cov_of_list<K>(best, k, eps, [&](unsigned idx) { return orig[idx]; }, out, (by_w & 256) != 0);
}


// mahalanobis_[i] = (R C1_i R^T + C2_j)^-1 for matched i (sorted-source order)
__global__ void __launch_bounds__(kBlock)
    k_gicp_mahal(unsigned n, const unsigned long long *__restrict__ keys,
                 const double *__restrict__ C1, const double *__restrict__ C2, Mat3d R,
                 double *__restrict__ mahal) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned j = (unsigned) keys[i];
    if (j == kNoIdx) return;
    double o[9];
    gicp_mahal_of(C1 + (size_t) i * 9, C2 + (size_t) j * 9, R.m, o);
    // component-major (nine arrays of n): the objective kernel, which reads these ~170 times per
    // registration, then gets fully coalesced 8-byte loads instead of a 72-byte stride per lane
#pragma unroll
    for (int a = 0; a < 9; ++a) mahal[(size_t) a * n + i] = o[a];
}


// the pairs i0, i0 + stride, ... < n straight from HBM
__device__ __forceinline__ void gicp_fdf_stream(double (&hi)[kGicpAcc], double (&lo)[kGicpAcc], unsigned i0,
                                                unsigned stride, const float4 *__restrict__ src, unsigned n,
                                                const unsigned long long *__restrict__ keys,
                                                const float4 *__restrict__ tgt, const double *__restrict__ mahal,
                                                const FdfArgs &A) {
    for (unsigned i = i0; i < n; i += stride) {
        // everything this pair needs is loaded up front (one round trip); an unmatched point
        // (rare) wastes its loads
        const unsigned j = (unsigned) keys[i];
        const float4 p = src[i], q = tgt[i];  // tgt = match coordinates per source point
        double M[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) M[k] = mahal[(size_t) k * n + i];
        if (j == kNoIdx) continue;
        gicp_fdf_point(hi, lo, A, p.x, p.y, p.z, q.x, q.y, q.z, M);
    }
}

// the workgroup's threads' pairs -> (h, l) of component threadIdx.x in threads 0..12 (contains a barrier; every
// thread calls; a second call needs a barrier of the caller's in between: the LDS array is the same)
__device__ __forceinline__ void gicp_block_reduce(double (&hi)[kGicpAcc], double (&lo)[kGicpAcc], double &h, double &l) {
    // wave reduction by recursive halving (as the search kernel's statistics, wm_nn.hip): at the
    // step for lane bit M a lane keeps one half of its pairs and sends the other half to lane ^ M --
    // 7+4+2+1+1+1 = 16 pair exchanges instead of 13 x 6.  Component k ends in the lane dd_comp_of_lane names.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    dd_halve<kGicpAcc, 32>(hi, lo, (unsigned) lane);
    __shared__ double lds[kBlock / 64][kGicpAcc][2];
    const int comp = dd_comp_of_lane((unsigned) lane);
    if (comp >= 0) {
        lds[wave][comp][0] = hi[0];
        lds[wave][comp][1] = lo[0];
    }
    __syncthreads();
    h = l = 0;
    if (threadIdx.x < kGicpAcc) {
        for (int w = 0; w < kBlock / 64; ++w) {
            dd_add(h, l, lds[w][threadIdx.x][0]);
            l += lds[w][threadIdx.x][1];
        }
    }
}

// ... -> the workgroup's row of `partials`
__device__ __forceinline__ void gicp_fdf_finish(double (&hi)[kGicpAcc], double (&lo)[kGicpAcc],
                                                double *__restrict__ partials) {
    double h, l;
    gicp_block_reduce(hi, lo, h, l);
    if (threadIdx.x < kGicpAcc) {
        double *row = partials + ((size_t) blockIdx.x * kGicpAcc + threadIdx.x) * 2;
        row[0] = h;
        row[1] = l;
    }
}

__global__ void __launch_bounds__(kBlock)
    k_gicp_fdf(const float4 *__restrict__ src, unsigned n,
               const unsigned long long *__restrict__ keys, const float4 *__restrict__ tgt,
               const double *__restrict__ mahal, FdfArgs A, double *__restrict__ partials) {
    double hi[kGicpAcc], lo[kGicpAcc];
#pragma unroll
    for (int k = 0; k < kGicpAcc; ++k) hi[k] = lo[k] = 0.0;
    gicp_fdf_stream(hi, lo, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, src, n, keys, tgt, mahal, A);
    gicp_fdf_finish(hi, lo, partials);
}

// The block rows of a LAUNCHED evaluation -> the thirteen sums, by ONE workgroup of kBlock threads
// (k_gicp_sum_fetch; the resident evaluator adds its rows in another order -- double-double sums: the same
// value whatever the order).  The rows' (hi, lo) pairs are one contiguous array of rows x 13
// 16-byte pairs; thread t < 247 = 19 x 13 adds pairs t, t + 247, t + 494, ... (component t % 13, rows
// t / 13 + 19 m) in double-double -- consecutive lanes read consecutive pairs, a dozen cache lines per
// wave instruction instead of 64 --, then thread c < 13 adds the 19 partial pairs of component c in
// order and rounds hi + lo once.  Returns the sum in threads 0..12 (every thread must call).
constexpr int kGicpSumGroups = kBlock / kGicpAcc;  // 19
typedef double gicp_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double gicp_sum_rows(const double *__restrict__ src, unsigned rows) {
    __shared__ double s_h[kGicpSumGroups][kGicpAcc], s_l[kGicpSumGroups][kGicpAcc];
    constexpr unsigned S = (unsigned) (kGicpSumGroups * kGicpAcc);
    const unsigned t = threadIdx.x, total = rows * (unsigned) kGicpAcc;
    if (t < S) {
        double h = 0, l = 0;
        constexpr int U = 8;
        for (unsigned e0 = t; e0 < total; e0 += U * S) {
            gicp_d2v v[U];
            const gicp_d2v *p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned e = e0 + (unsigned) u * S;
                p[u] = (const gicp_d2v *) src + (e < total ? e : 0u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *p[u];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e0 + (unsigned) u * S < total) {
                    dd_add(h, l, v[u].x);
                    l += v[u].y;
                }
        }
        s_h[t / kGicpAcc][t % kGicpAcc] = h;
        s_l[t / kGicpAcc][t % kGicpAcc] = l;
    }
    __syncthreads();
    double out = 0.0;
    if (t < (unsigned) kGicpAcc) {
        double h = 0, l = 0;
#pragma unroll
        for (int g = 0; g < kGicpSumGroups; ++g) {
            dd_add(h, l, s_h[g][t]);
            l += s_l[g][t];
        }
        out = h + l;
    }
    __syncthreads();  // (the arrays may be used again by the next round)
    return out;
}

// The block pairs -> kGicpAcc doubles in pinned memory, then the fence + flag of fast_fetch.
__global__ void __launch_bounds__(kBlock)
    k_gicp_sum_fetch(double *dst, const double *__restrict__ src, unsigned rows, unsigned *flag, unsigned seq) {
    const double v = gicp_sum_rows(src, rows);
    if (threadIdx.x < (unsigned) kGicpAcc) dst[threadIdx.x] = v;
    if (threadIdx.x < 64) {
        __threadfence_system();
        if (threadIdx.x == 0) *(volatile unsigned *) flag = seq;
    }
}

// ---- the objective as sufficient statistics (wm_gicp_quad.hpp).  One pass per outer iteration: every matched
// pair's Mahalanobis matrix is formed (never stored) and its 74 terms go into double-double sums.  A thread keeps
// HALF of the accumulators (37 (hi, lo) pairs: all 74 would be 300 registers); blockIdx.y says which half, both
// halves read the same pairs (the second reading comes from L2).  Rows: [gridDim.x][74] (hi, lo) pairs.
constexpr int kQuadHalf = kQuadN / 2;
static_assert(kQuadHalf * 2 == kQuadN, "two equal halves");
struct QuadT0 {
    float m[12];
};
template <int PART>
__device__ __forceinline__ void gicp_quad_part(const float4 *__restrict__ src, unsigned n,
                                               const unsigned long long *__restrict__ keys,
                                               const float4 *__restrict__ match, const double *__restrict__ C1,
                                               const double *__restrict__ C2, const QuadT0 &T0,
                                               double *__restrict__ rows) {
    double hi[kQuadHalf], lo[kQuadHalf];
#pragma unroll
    for (int k = 0; k < kQuadHalf; ++k) hi[k] = lo[k] = 0.0;
    double R[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) R[a * 3 + b] = (double) T0.m[a * 4 + b];
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned j = (unsigned) keys[i];
        const float4 p = src[i], q = match[i];
        if (j == kNoIdx) continue;
        double M[9];
        gicp_mahal_of(C1 + (size_t) i * 9, C2 + (size_t) j * 9, R, M);
        gicp_quad_terms(T0.m, p.x, p.y, p.z, q.x, q.y, q.z, M, [&](int idx, double term) {
            if (idx >= PART * kQuadHalf && idx < (PART + 1) * kQuadHalf) ddn_add(hi[idx - PART * kQuadHalf], lo[idx - PART * kQuadHalf], term);
        });
    }
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    ddn_halve<kQuadHalf, kQuadHalf, 32>(hi, lo, lane);
    __shared__ double lds[kBlock / 64][kQuadHalf][2];
    const int comp = ddn_comp_of_lane<kQuadHalf>(lane);
    if (comp >= 0) {
        lds[wave][comp][0] = hi[0];
        lds[wave][comp][1] = lo[0];
    }
    __syncthreads();
    if (threadIdx.x < (unsigned) kQuadHalf) {
        double h = 0, l = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            ddn_add(h, l, lds[w][threadIdx.x][0]);
            l += lds[w][threadIdx.x][1];
        }
        double *row = rows + ((size_t) blockIdx.x * kQuadN + PART * kQuadHalf + threadIdx.x) * 2;
        row[0] = h;
        row[1] = l;
    }
}
__global__ void __launch_bounds__(kBlock)
    k_gicp_quad(const float4 *__restrict__ src, unsigned n, const unsigned long long *__restrict__ keys,
                const float4 *__restrict__ match, const double *__restrict__ C1, const double *__restrict__ C2,
                QuadT0 T0, double *__restrict__ rows) {
    if (blockIdx.y == 0) gicp_quad_part<0>(src, n, keys, match, C1, C2, T0, rows);
    else gicp_quad_part<1>(src, n, keys, match, C1, C2, T0, rows);
}
// the rows -> the 74 sums (each hi + lo rounded once), in pinned memory, then the fence + flag of fast_fetch.
// One workgroup of 1024: thread t < 962 = 13 x 74 adds the rows t / 74, t / 74 + 13, ... of column t % 74
// (consecutive lanes read consecutive pairs), wave 0 adds the thirteen partial pairs of every column in order.
__global__ void __launch_bounds__(1024)
    k_gicp_quad_fetch(double *dst, const double *__restrict__ rows, unsigned nrows, unsigned *flag, unsigned seq) {
    constexpr unsigned kGroups = 1024u / (unsigned) kQuadN;  // 13
    __shared__ double part[kGroups][kQuadN][2];
    const unsigned t = threadIdx.x;
    if (t < kGroups * (unsigned) kQuadN) {
        const unsigned c = t % (unsigned) kQuadN, g = t / (unsigned) kQuadN;
        double h = 0, l = 0;
        for (unsigned r = g; r < nrows; r += kGroups) {
            const double *e = rows + ((size_t) r * kQuadN + c) * 2;
            ddn_add(h, l, e[0]);
            l += e[1];
        }
        part[g][c][0] = h;
        part[g][c][1] = l;
    }
    __syncthreads();
    if (t < 64u) {
        for (unsigned c = t; c < (unsigned) kQuadN; c += 64u) {
            double h = 0, l = 0;
            for (unsigned g = 0; g < kGroups; ++g) {
                ddn_add(h, l, part[g][c][0]);
                l += part[g][c][1];
            }
            dst[c] = h + l;
        }
        if (flag) {
            __threadfence_system();
            if (t == 0) *(volatile unsigned *) flag = seq;
        }
    }
}

// ---- served evaluations.  One BFGS minimisation asks for ~20 evaluations of the objective, each at
// a point that depends on the previous answer, and the optimiser stays on the host (its libm calls
// are part of what has to match PCL bit for bit).  Launching a kernel pair per evaluation costs
// ~15 us of launch + completion latency against ~14 us of work.  Instead ONE kernel stays resident
// for the whole minimisation: its workgroups wait for the next trial point in a mailbox in DEVICE
// memory, which the host writes straight through the PCIe BAR (a posted write: the kernel sees it
// ~1 us later; polling costs the GPU nothing but its own memory), evaluate exactly as k_gicp_fdf
// does, and the workgroup that finishes last adds the rows exactly as k_gicp_sum_fetch does and
// writes the thirteen sums + flag into pinned host memory.  Same partial rows, same order of
// addition: the same bits.  The kernel leaves when told to, or by itself when no command arrives
// for kServeGuardTicks (a host that died or was descheduled: the host side then falls back to
// launching, see gicp_fdf).
struct GicpMailbox {
    float T[12], B[12];   // FdfArgs of the evaluation
    unsigned cmd;         // 1: evaluate, 2: leave
    unsigned seq;         // command number (written last); the words above belong to it
    unsigned abandoned;   // the kernel gave up waiting (set by the device)
    unsigned pad[5];
};
struct alignas(16) GicpSlot {  // pinned host memory: one of the thirteen sums + the command it belongs to
    double v;
    unsigned long long seq;
};
constexpr int kServeCached = 8;  // pairs per thread kept on chip by the cached variant
constexpr unsigned long long kServeGuardTicks = 20000000ull;  // 0.2 s of the 100 MHz wall clock
constexpr size_t kServeRowsOffset = 4096, kServeRowsBytes = (size_t) kGicpBlocksMax * 2 * kGicpAcc * 16;  // (in the mailbox's allocation)

__device__ __forceinline__ unsigned ld_sys(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// PL: how many of a thread's pairs (i, i + stride, ...) stay ON CHIP for the whole minimisation -- the
// coordinates in registers, the Mahalanobis matrices in LDS (72 B x PL x 256 threads: 147 KB of the CU's
// 160 KB at PL = 8) --, loaded once when the kernel starts: a 500k pair on 256 workgroups is 7.6 pairs per
// thread, so an evaluation reads nothing from HBM at all (11 us of streaming become ~3 us of LDS reads
// and f64 arithmetic).  Pairs beyond PL per thread are streamed as before.  Same terms in the same
// order either way.
template <int PL>
__global__ void __launch_bounds__(kBlock)
    k_gicp_fdf_served(const float4 *__restrict__ src, unsigned n, const unsigned long long *__restrict__ keys,
                      const float4 *__restrict__ tgt, const double *__restrict__ mahal, GicpMailbox *mb,
                      unsigned first_seq, unsigned __attribute__((ext_vector_type(4))) *rows, GicpSlot *h_slots,
                      unsigned long long *dbg) {
    __shared__ float s_args[24];
    __shared__ unsigned s_cmd;
    extern __shared__ double s_M[];  // [PL][9][kBlock]: column = thread (private to it: no barrier needed)
    const unsigned stride = gridDim.x * kBlock, i_first = blockIdx.x * kBlock + threadIdx.x;
    float cpx[PL > 0 ? PL : 1], cpy[PL > 0 ? PL : 1], cpz[PL > 0 ? PL : 1];
    float cqx[PL > 0 ? PL : 1], cqy[PL > 0 ? PL : 1], cqz[PL > 0 ? PL : 1];
    unsigned cvalid = 0;
    if constexpr (PL > 0) {
#pragma unroll
        for (int k = 0; k < PL; ++k) {
            const unsigned i = i_first + (unsigned) k * stride;
            cpx[k] = cpy[k] = cpz[k] = cqx[k] = cqy[k] = cqz[k] = 0.f;
            if (i < n) {
                const unsigned j = (unsigned) keys[i];
                const float4 p = src[i], q = tgt[i];
#pragma unroll
                for (int c = 0; c < 9; ++c) s_M[(k * 9 + c) * kBlock + threadIdx.x] = mahal[(size_t) c * n + i];
                cpx[k] = p.x, cpy[k] = p.y, cpz[k] = p.z;
                cqx[k] = q.x, cqy[k] = q.y, cqz[k] = q.z;
                if (j != kNoIdx) cvalid |= 1u << k;
            }
        }
    }
    for (unsigned seq = first_seq;; ++seq) {
        if (threadIdx.x < 64) {
            bool ok = true;
            if (threadIdx.x == 0) {
                const unsigned long long t0 = wall_clock64();
                // (one look every ~0.5 us per workgroup: hundreds of workgroups looking as fast as they can
                // keep the memory channel of that one line so busy that the stragglers of the evaluation
                // under way, and the next command's arrival, are delayed by tens of microseconds)
                for (;;) {
                    const unsigned cur = ld_sys(&mb->seq);
                    if (cur == seq) break;
                    // a command PAST this workgroup's next one: it was not resident while the others worked
                    // (the host has given up on the round by now): leave
                    if ((int) (cur - seq) > 0 || wall_clock64() - t0 > kServeGuardTicks) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
                if (!ok) __hip_atomic_store(&mb->abandoned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            ok = __builtin_amdgcn_readfirstlane((int) ok) != 0;
            if (threadIdx.x < 24)
                s_args[threadIdx.x] = __uint_as_float(ld_sys((const unsigned *) mb->T + threadIdx.x));  // (T and B are adjacent)
            if (threadIdx.x == 24) s_cmd = ok ? ld_sys(&mb->cmd) : 2u;
        }
        __syncthreads();
        if (s_cmd != 1u) return;  // (uniform)
        const unsigned round = seq - first_seq;
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0 && round < 64u) dbg[round * 4 + 0] = wall_clock64();
        FdfArgs A;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            A.T[k] = __uint_as_float((unsigned) __builtin_amdgcn_readfirstlane((int) __float_as_uint(s_args[k])));
            A.B[k] = __uint_as_float((unsigned) __builtin_amdgcn_readfirstlane((int) __float_as_uint(s_args[12 + k])));
        }
        double hi[kGicpAcc], lo[kGicpAcc];
#pragma unroll
        for (int k = 0; k < kGicpAcc; ++k) hi[k] = lo[k] = 0.0;
        if constexpr (PL > 0) {
#pragma unroll
            for (int k = 0; k < PL; ++k)
                if ((cvalid >> k) & 1u) {
                    double M[9];
#pragma unroll
                    for (int c = 0; c < 9; ++c) M[c] = s_M[(k * 9 + c) * kBlock + threadIdx.x];
                    gicp_fdf_point(hi, lo, A, cpx[k], cpy[k], cpz[k], cqx[k], cqy[k], cqz[k], M);
                }
        }
        gicp_fdf_stream(hi, lo, i_first + (unsigned) PL * stride, stride, src, n, keys, tgt, mahal, A);
        double bh, bl;
        gicp_block_reduce(hi, lo, bh, bl);
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0 && round < 64u) dbg[round * 4 + 1] = wall_clock64();
        // The workgroup's row goes out as 26 16-byte slots {value, number of the round}, one store each, written
        // through to where every XCD sees it (sc1); slot (c, row) lies at c * rows + row.  Workgroup 0 adds the rows
        // up: its thread t reads row t -- 26 loads that are consecutive across the wave --, again until every slot
        // carries this round's number, so the wait for the other workgroups IS the fetch of their rows.  (Rounds 2-3:
        // rows, then a ticket drawn with an atomic per workgroup -- 256 atomics on one address retire one after the
        // other, ~11 ns each --, then the last workgroup fetched the rows: two trips to memory and 3 us behind the
        // last workgroup's last store.  No agent-scope FENCE anywhere: a release fence writes the XCD's whole L2 back.)
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        if (threadIdx.x < (unsigned) kGicpAcc) {
            const unsigned long long hb = (unsigned long long) __double_as_longlong(bh), lb = (unsigned long long) __double_as_longlong(bl);
            const u4v oh = {(unsigned) hb, (unsigned) (hb >> 32), seq, 0u}, ol = {(unsigned) lb, (unsigned) (lb >> 32), seq, 0u};
            u4v *dh = rows + (size_t) (2u * threadIdx.x) * gridDim.x + blockIdx.x, *dl = dh + gridDim.x;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dh), "v"(oh) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dl), "v"(ol) : "memory");
        }
        if (blockIdx.x == 0u) {  // (uniform)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) rows, 0, gridDim.x * 2u * (unsigned) kGicpAcc * 16u, 0x00020000);
            const unsigned long long t0 = wall_clock64();
            double sh[kGicpAcc], sl[kGicpAcc];
#pragma unroll
            for (int k = 0; k < kGicpAcc; ++k) sh[k] = sl[k] = 0.0;
            bool ok = true;
            for (unsigned row = threadIdx.x; row < gridDim.x && ok; row += kBlock) {
                u4v v[2 * kGicpAcc];
                for (;;) {
                    unsigned off = row * 16u;
                    asm volatile("" : "+v"(off));  // (a fresh look every trip: nothing of it may be kept from the previous one)
                    bool all = true;
#pragma unroll
                    for (int c = 0; c < 2 * kGicpAcc; ++c) {
                        v[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, (unsigned) c * gridDim.x * 16u, 16);  // sc1
                        all = all && v[c].z == seq;
                    }
                    if (all) break;
                    if (wall_clock64() - t0 > kServeGuardTicks) {  // (a workgroup that never came: no answer; the host's own limit ends the round)
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (ok) {
#pragma unroll
                    for (int c = 0; c < kGicpAcc; ++c) {
                        dd_add(sh[c], sl[c], __longlong_as_double((long long) (((unsigned long long) v[2 * c].y << 32) | v[2 * c].x)));
                        sl[c] += __longlong_as_double((long long) (((unsigned long long) v[2 * c + 1].y << 32) | v[2 * c + 1].x));
                    }
                }
            }
            if (dbg && threadIdx.x == 0 && round < 64u) dbg[round * 4 + 2] = wall_clock64();
            // (the barrier the second use of the reduction's LDS array needs, and: did any thread give up?)
            if (__syncthreads_and((int) ok)) {
                double h, l;
                gicp_block_reduce(sh, sl, h, l);
                if (threadIdx.x < (unsigned) kGicpAcc) {
                    // the sum and the number of the command it answers in ONE 16-byte store to pinned host
                    // memory: the host takes a slot once it carries the number it waits for -- no flag
                    // behind the data, hence no system-scope fence (an L2 write-back: ~5 us) before one.
                    // (written through at system scope -- sc0 sc1 --, as an atomic store would be; there is no
                    // 16-byte atomic store to ask the compiler for)
                    const double v = h + l;
                    const unsigned long long vb = (unsigned long long) __double_as_longlong(v);
                    const u4v out = {(unsigned) vb, (unsigned) (vb >> 32), seq, 0u};
                    GicpSlot *dst = &h_slots[threadIdx.x];
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(out) : "memory");
                }
            }
            if (dbg && threadIdx.x == 0 && round < 64u) dbg[round * 4 + 3] = wall_clock64();
        }
    }
}

__global__ void __launch_bounds__(kBlock)
    k_count_matched(const unsigned long long *__restrict__ keys, unsigned n, unsigned *out) {
    unsigned c = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        c += ((unsigned) keys[i] != kNoIdx);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    // one atomic per WORKGROUP, and few workgroups: same-address atomics retire one at a time
    // (~11 ns each); one per wave of a 1024-block grid made this count take 49 us
    __shared__ unsigned s_c[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += s_c[w];
        if (t) atomicAdd(out, t);
    }
}

// number of matched source points -> host (pinned fetch: no pageable staging, no stream sync)
static int count_matched(wm_ctx *ctx, size_t n, unsigned *cnt) {
    WM_HIP(ctx, ctx->bbox_buf.reserve(64));
    unsigned *d_cnt = ctx->bbox_buf.as<unsigned>();
    WM_HIP(ctx, hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
    unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    if (blocks > 128) blocks = 128;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_count_matched, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       ctx->keys.as<unsigned long long>(), (unsigned) n, d_cnt);
    WM_HIP(ctx, hipGetLastError());
    unsigned *h = (unsigned *) pinned_scratch(ctx, 0);
    if (!h) return WM_ERR_HIP;
    WM_TRY(fast_fetch(ctx, h, d_cnt, 4));
    *cnt = *h;
    return WM_OK;
}

// ------------------------------------------------------------------ host
struct GicpFn {
    wm_ctx *ctx;
    const double *base;
    int m = 0;  // matched pairs
    // (what bfgs_minimize asks of its objective, wm_bfgs.hpp)
    double fdf(const double x[6], double g[6]);
    int pairs() const { return m; }
    bool failed() const { return rc != WM_OK; }
    bool test_at_start() const { return false; }  // (PCL's per-pair objective: pcl::BFGS as it is)
    int evals = 0;
    float kernel_ms = 0;
    int rc = WM_OK;
    // served evaluations (k_gicp_fdf_served resident for the minimisation under way)
    bool served = false;
    unsigned served_evals = 0;
    int served_blocks = 0;  // this evaluator's share of the device's budget (1/1024ths)
    int served_total = 0, served_fallbacks = 0;
    double host_us[64] = {0};
};

// Resident evaluators of one device and process share a budget of workgroups (what the device can keep
// resident at once): two resident kernels that each hold part of the GPU while waiting for their
// hosts must never keep each other's remaining workgroups from starting.  A 500k pair takes the whole
// budget (256 workgroups); the matchers of a MultiMatcher pool working on 20k-point pairs (79
// workgroups each) get three evaluators side by side, the others launch their evaluations meanwhile.
// (the budget itself: resident_admit / resident_release, wm_nn.hip -- shared with the resident ICP kernel)
static int gicp_blocks(const wm_ctx *ctx) {
    int nb = (int) ((ctx->n_src + kBlock - 1) / kBlock);
    if (nb > ctx->tune_gicp_blocks) nb = ctx->tune_gicp_blocks;
    if (nb < 1) nb = 1;
    return nb;
}

static void serve_post(wm_ctx *ctx, const FdfArgs *A, unsigned cmd) {
    // plain stores into device memory through the BAR; the command number goes last, behind a store fence
    // (16-byte stores: every store into this mapping is a PCIe write of its own, ~0.1 us each)
    volatile GicpMailbox *mb = (volatile GicpMailbox *) ctx->gicp_mailbox.p;
    typedef float v4f __attribute__((vector_size(16)));
    if (A) {
        static_assert(sizeof(FdfArgs) == 96 && offsetof(GicpMailbox, T) == 0 && offsetof(GicpMailbox, B) == 48, "layout");
        v4f w[6];
        memcpy(w, A, sizeof(w));
        for (int k = 0; k < 6; ++k) ((volatile v4f *) mb)[k] = w[k];
    }
    mb->cmd = cmd;
    __sync_synchronize();
    mb->seq = ++ctx->gicp_serve_seq;
    __sync_synchronize();
}

// start the resident evaluator for one minimisation (quietly not, when it cannot be used here)
static void serve_begin(GicpFn &F) {
    wm_ctx *ctx = F.ctx;
    F.served = false;
    if (!ctx->tune_gicp_served || ctx->gicp_profile || getenv("WM_GICP_TRACE")) return;
    if (ctx->device < 0 || ctx->device >= 64) return;
    if (ctx->gicp_serve_ok < 0) return;
    if (ctx->gicp_serve_ok == 0) {  // first use: is device memory host-writable, and does the grid fit?
        ctx->gicp_serve_ok = -1;
        int large_bar = 0, cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess || !large_bar) return;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) return;
        // the variant that keeps eight pairs per thread on chip needs 147 KB of LDS per workgroup
        constexpr size_t kCacheLds = (size_t) kServeCached * 9 * kBlock * sizeof(double);
        ctx->gicp_serve_cached = 0;
        if (hipFuncSetAttribute((const void *) k_gicp_fdf_served<kServeCached>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int) kCacheLds) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gicp_fdf_served<kServeCached>, kBlock, kCacheLds) == hipSuccess &&
            per_cu >= 1) {
            ctx->gicp_serve_cached = 1;
        } else {
            (void) hipGetLastError();
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gicp_fdf_served<0>, kBlock, 0) != hipSuccess) return;
        }
        ctx->gicp_serve_capacity = cus * per_cu;
        if (ctx->gicp_mailbox.reserve(kServeRowsOffset + kServeRowsBytes) != hipSuccess) return;
        if (hipMemsetAsync(ctx->gicp_mailbox.p, 0, kServeRowsOffset + kServeRowsBytes, ctx->stream) != hipSuccess) return;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return;
        ctx->gicp_serve_seq = 0;
        ctx->gicp_serve_ok = 1;
    }
    int nb = gicp_blocks(ctx);
    if (nb > ctx->gicp_serve_capacity) return;  // every workgroup has to be resident at once
    // (128 doubles wherever it is allocated: the statistics objective fetches 74 into it)
    if (!ctx->h_gicp && hipHostMalloc((void **) &ctx->h_gicp, sizeof(double) * 128, hipHostMallocDefault) != hipSuccess) return;
    if (!ctx->h_gicp_slots) {
        if (hipHostMalloc((void **) &ctx->h_gicp_slots, sizeof(GicpSlot) * 16, hipHostMallocDefault) != hipSuccess) return;
        memset(ctx->h_gicp_slots, 0, sizeof(GicpSlot) * 16);
    }
    int share = resident_admit(ctx->device, nb, ctx->gicp_serve_capacity);
    if (share <= 0 && ctx->gicp_serve_cached) {
        // The device's budget is taken by other matchers' evaluators (a MultiMatcher pool of small pairs): come
        // in COMPACT -- up to kServeCached pairs per thread instead of one, an eighth of the workgroups; an
        // evaluation takes a microsecond or two longer, and eight times as many evaluators fit side by side.
        // (The rows are added in double-double: the sums do not depend on how the pairs are dealt out.)
        const int compact = (int) ((ctx->n_src + (size_t) kBlock * kServeCached - 1) / ((size_t) kBlock * kServeCached));
        if (compact >= 1 && compact < nb) {
            share = resident_admit(ctx->device, compact, ctx->gicp_serve_capacity);
            if (share > 0) nb = compact;
        }
    }
    if (share <= 0) return;
    F.served_blocks = share;
    F.served_evals = 0;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    u4v *rows = (u4v *) ((char *) ctx->gicp_mailbox.p + kServeRowsOffset);  // the workgroups' rows of the round under way ({value, round} slots)
    // A round that ended in a fallback leaves `abandoned` set: it goes back to 0 on the stream before every
    // launch (NOT the command number).  The rows need no reset: their slots carry numbers of rounds that are over,
    // and no later round has one of those again.
    if (hipMemsetAsync((char *) ctx->gicp_mailbox.p + offsetof(GicpMailbox, abandoned), 0, sizeof(unsigned), ctx->stream) != hipSuccess) {
        (void) hipGetLastError();
        resident_release(ctx->device, share);
        return;
    }
    unsigned long long *dbg = getenv("WM_GICP_SERVE_DEBUG") ? (unsigned long long *) ((char *) ctx->gicp_mailbox.p + 256) : nullptr;
    if (ctx->gicp_serve_cached && ctx->tune_gicp_served != 2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gicp_fdf_served<kServeCached>), dim3(nb), dim3(kBlock),
                           (size_t) kServeCached * 9 * kBlock * sizeof(double), ctx->stream, ctx->src_sorted.as<float4>(),
                           (unsigned) ctx->n_src, ctx->keys.as<unsigned long long>(), ctx->match_pt.as<float4>(),
                           ctx->gicp_mahal.as<double>(), (GicpMailbox *) ctx->gicp_mailbox.p, ctx->gicp_serve_seq + 1u,
                           rows, (GicpSlot *) ctx->h_gicp_slots, dbg);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gicp_fdf_served<0>), dim3(nb), dim3(kBlock), 0, ctx->stream,
                           ctx->src_sorted.as<float4>(), (unsigned) ctx->n_src, ctx->keys.as<unsigned long long>(),
                           ctx->match_pt.as<float4>(), ctx->gicp_mahal.as<double>(), (GicpMailbox *) ctx->gicp_mailbox.p,
                           ctx->gicp_serve_seq + 1u, rows, (GicpSlot *) ctx->h_gicp_slots, dbg);
    if (hipGetLastError() != hipSuccess) {
        resident_release(ctx->device, share);
        return;
    }
    F.served = true;
}

// tell the evaluator to leave and wait until it has (also after a failure: nothing of it may be
// left on the stream when the next kernels are queued)
static void serve_end(GicpFn &F) {
    if (!F.served) return;
    wm_ctx *ctx = F.ctx;
    serve_post(ctx, nullptr, 2u);
    (void) hipStreamSynchronize(ctx->stream);
    if (F.served_fallbacks > 0) {
        // (diagnostic, after a fallback only: a read through the BAR is a PCIe round trip) did workgroups leave by themselves?
        const unsigned ab = ((volatile GicpMailbox *) ctx->gicp_mailbox.p)->abandoned;
        if (ab && ctx->trace) fprintf(stderr, "[wm] gicp: resident evaluator abandoned a round (guard / late workgroups)\n");
        ctx->gicp_serve_abandoned += ab ? 1 : 0;
    }
    if (getenv("WM_GICP_SERVE_DEBUG")) {  // developer: device-side stamps of the first rounds (100 MHz ticks)
        unsigned long long d[64 * 4];
        if (hipMemcpy(d, (char *) ctx->gicp_mailbox.p + 256, sizeof(d), hipMemcpyDeviceToHost) == hipSuccess) {
            for (unsigned r = 0; r < F.served_evals && r < 24; ++r)
                fprintf(stderr, "  round %2u: seen->computed %6.2f us, computed->last arrives %6.2f us, sum+reply %6.2f us | since previous reply %6.2f us | host: post->answer %6.2f us\n",
                        r, (d[r * 4 + 1] - d[r * 4]) * 0.01, ((long long) d[r * 4 + 2] - (long long) d[r * 4 + 1]) * 0.01,
                        (d[r * 4 + 3] - d[r * 4 + 2]) * 0.01, r ? ((long long) d[r * 4] - (long long) d[(r - 1) * 4 + 3]) * 0.01 : 0.0,
                        r < 64 ? F.host_us[r] : 0.0);
        }
    }
    F.served = false;
    resident_release(ctx->device, F.served_blocks);
}

// one served evaluation; false: no answer (the evaluator gave up or is stuck) -> it has been shut down
static bool serve_eval(GicpFn &F, const FdfArgs &A) {
    wm_ctx *ctx = F.ctx;
    if (ctx->gicp_serve_test_stall_ms > 0 && F.served_evals == 2) {
        // test hook (wm_set_option "gicp_serve_test_stall_ms"): a host that goes away for longer than the
        // evaluator's guard -- the kernel must have left by itself, and this call must recover by launching
        std::this_thread::sleep_for(std::chrono::milliseconds(ctx->gicp_serve_test_stall_ms));
    }
    const auto t0 = std::chrono::steady_clock::now();
    serve_post(ctx, &A, 1u);
    const unsigned long long expect = ctx->gicp_serve_seq;
    volatile GicpSlot *slots = (volatile GicpSlot *) ctx->h_gicp_slots;
    int have = 0;  // slots 0 .. have - 1 carry this command's number
    for (unsigned spins = 1; have < kGicpAcc; ++spins) {
        while (have < kGicpAcc && slots[have].seq == expect) ++have;
        if (have == kGicpAcc) break;
        cpu_relax();
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100)) {
            serve_end(F);  // (waits for the kernel: it leaves on the command, or by its own guard)
            F.served_fallbacks++;
            return false;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int k = 0; k < kGicpAcc; ++k) ctx->h_gicp[k] = slots[k].v;
    if (F.served_evals < 64) F.host_us[F.served_evals] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    F.served_evals++;
    F.served_total++;
    return true;
}

static double gicp_fdf(GicpFn &F, const double x[6], double g[6]) {
    wm_ctx *ctx = F.ctx;
    FdfArgs A;
    float T[16];
    state_to_matrix_f(F.base, x, T);
    for (int k = 0; k < 12; ++k) {
        A.T[k] = T[k];
        A.B[k] = (float) F.base[k];
    }
    const unsigned n = (unsigned) ctx->n_src;
    const int nb = gicp_blocks(ctx);
    // The block partials stay in device memory; one workgroup adds them up and writes the
    // kGicpAcc sums into pinned memory (fast_fetch_sum): no copy engine, no pageable staging,
    // 104 bytes over PCIe -- this loop runs ~180 times per registration and is latency-bound.
    if (!ctx->h_gicp &&
        hipHostMalloc((void **) &ctx->h_gicp, sizeof(double) * 128,
                      hipHostMallocDefault) != hipSuccess) {
        F.rc = WM_ERR_HIP;
        ctx->last_error = "gicp_fdf: hipHostMalloc failed";
        return 0;
    }
    if (!(F.served && serve_eval(F, A))) {
        if (ctx->gicp_profile) (void) hipEventRecord(ctx->ev_a, ctx->stream);
        hipLaunchKernelGGL(k_gicp_fdf, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->src_sorted.as<float4>(),
                           n, ctx->keys.as<unsigned long long>(), ctx->match_pt.as<float4>(),
                           ctx->gicp_mahal.as<double>(), A, ctx->partials.as<double>());
        if (ctx->gicp_profile) (void) hipEventRecord(ctx->ev_b, ctx->stream);
        if (fast_fetch_custom(ctx, [&](unsigned *flag, unsigned seq) {
                hipLaunchKernelGGL(k_gicp_sum_fetch, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->h_gicp,
                                   ctx->partials.as<double>(), (unsigned) nb, flag, seq);
            }) != WM_OK) {
            F.rc = WM_ERR_HIP;
            return 0;
        }
    }
    if (ctx->gicp_profile) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
        F.kernel_ms += ms;
    }
    F.evals++;
    const double *a = ctx->h_gicp;  // the kGicpAcc sums over blocks
    const double m = (double) F.m;
    if (g) {
        double Racc[9];
        for (int k = 0; k < 3; ++k) g[k] = a[1 + k] * 2.0 / m;
        for (int k = 0; k < 9; ++k) Racc[k] = a[4 + k] * 2.0 / m;
        r_derivative(x, Racc, g);
    }
    if (const char *path = getenv("WM_GICP_TRACE")) {  // developer: every evaluation, in hex floats
        if (FILE *fp = fopen(path, "a")) {
            fprintf(fp, "%d", F.m);
            for (int k = 0; k < 6; ++k) fprintf(fp, " %.17g", x[k]);
            fprintf(fp, " | %.17g |", a[0] / m);
            if (g) for (int k = 0; k < 6; ++k) fprintf(fp, " %.17g", g[k]);
            fprintf(fp, "\n");
            fclose(fp);
        }
    }
    return a[0] / m;
}

double GicpFn::fdf(const double x[6], double g[6]) { return gicp_fdf(*this, x, g); }

// ---- the statistics objective on the host: what bfgs_minimize asks of its objective (wm_bfgs.hpp), answered from
// the 74 sums of the pairs (no device work per evaluation)
struct GicpQuadFn {
    double Q[kQuadN];
    float T0[12];
    const double *base;
    int m = 0;
    int evals = 0;
    int pairs() const { return m; }
    bool failed() const { return false; }
    bool test_at_start() const { return true; }  // (see bfgs_minimize)
    double fdf(const double x[6], double g[6]) {
        ++evals;
        const double f = gicp_quad_eval(Q, T0, base, x, g);
        if (const char *path = getenv("WM_GICP_TRACE")) {  // developer: every evaluation (as gicp_fdf prints them)
            if (FILE *fp = fopen(path, "a")) {
                fprintf(fp, "%d", m);
                for (int k = 0; k < 6; ++k) fprintf(fp, " %.17g", x[k]);
                fprintf(fp, " | %.17g |", f);
                if (g) for (int k = 0; k < 6; ++k) fprintf(fp, " %.17g", g[k]);
                fprintf(fp, "\n");
                fclose(fp);
            }
        }
        return f;
    }
};
// the 74 sums of the pairs the last search left (keys / match_pt), found under the float transform T -> F
static int gicp_quad_statistics(wm_ctx *ctx, const float T[16], GicpQuadFn &F, float *kernel_ms) {
    const unsigned n = (unsigned) ctx->n_src;
    const int nb = gicp_blocks(ctx);
    WM_HIP(ctx, ctx->partials.reserve((size_t) kGicpBlocksMax * kQuadN * 2 * sizeof(double) + 64));
    if (!ctx->h_gicp) WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_gicp, sizeof(double) * 128, hipHostMallocDefault));
    QuadT0 T0;
    for (int k = 0; k < 12; ++k) T0.m[k] = F.T0[k] = T[k];
    if (ctx->gicp_profile) (void) hipEventRecord(ctx->ev_a, ctx->stream);
    hipLaunchKernelGGL(k_gicp_quad, dim3((unsigned) nb, 2), dim3(kBlock), 0, ctx->stream, ctx->src_sorted.as<float4>(), n,
                       ctx->keys.as<unsigned long long>(), ctx->match_pt.as<float4>(), ctx->gicp_c1.as<double>(),
                       ctx->gicp_c2.as<double>(), T0, ctx->partials.as<double>());
    WM_HIP(ctx, hipGetLastError());
    if (ctx->gicp_profile) (void) hipEventRecord(ctx->ev_b, ctx->stream);
    WM_TRY(fast_fetch_custom(ctx, [&](unsigned *flag, unsigned seq) {
        hipLaunchKernelGGL(k_gicp_quad_fetch, dim3(1), dim3(1024), 0, ctx->stream, ctx->h_gicp, ctx->partials.as<double>(),
                           (unsigned) nb, flag, seq);
    }));
    if (ctx->gicp_profile && kernel_ms) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
        *kernel_ms += ms;
    }
    for (int k = 0; k < kQuadN; ++k) F.Q[k] = ctx->h_gicp[k];
    F.m = (int) F.Q[kQuadOffCount];
    return WM_OK;
}

static float choose_cell(const Bbox &bb, size_t n) {
    double vol = 1;
    for (int d = 0; d < 3; ++d) vol *= fmax((double) bb.hi[d] - bb.lo[d], 1e-3);
    return (float) fmax(cbrt(vol / fmax((double) n, 1.0)) * 1.5, 1e-4);
}

template <int K>
static int launch_cov(wm_ctx *ctx, const GridDev &g, const float4 *q, size_t n, const float4 *orig,
                      int k, double eps, double *out, int by_w) {
    if (n == 0) return WM_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gicp_cov<K>), dim3((unsigned) ((n + kCovBlock - 1) / kCovBlock)),
                       dim3(kCovBlock), 0, ctx->stream, g, q, (unsigned) n, orig, k, eps, out, by_w | ctx->tune_cov_dbg,
                       ctx->tune_knn_r0 > 0 ? ctx->tune_knn_r0 : (k <= 12 ? 1.0f : 1.5f));
    WM_HIP(ctx, hipGetLastError());
#ifdef WM_COV_COUNT
    double c[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipMemcpyFromSymbol(c, HIP_SYMBOL(g_knn_cnt), sizeof(c));
    (void) hipMemcpyToSymbol(HIP_SYMBOL(g_knn_cnt), z, sizeof(z));
    const double waves = (double) n / 64.0;
    fprintf(stderr, "[knn] n %zu: candidates/query %.1f, wave trips/wave %.1f, batches/query %.2f, wave batches/wave %.2f, passes/query %.2f\n", n,
            c[0] / n, c[1] / waves, c[2] / n, c[3] / waves, c[4] / n);
#endif
    return WM_OK;
}

// the neighbour list lives in registers, so its length is a template parameter: the smallest
// instantiated size >= k keeps both the insertion cost (K compare-swaps per accepted candidate,
// executed by the whole wave) and the register footprint down
static int launch_cov_k(wm_ctx *ctx, const GridDev &g, const float4 *q, size_t n, const float4 *orig,
                        int k, double eps, double *out, int by_w) {
    if (k <= 8) return launch_cov<8>(ctx, g, q, n, orig, k, eps, out, by_w);
    if (k <= 10) return launch_cov<10>(ctx, g, q, n, orig, k, eps, out, by_w);
    if (k <= 12) return launch_cov<12>(ctx, g, q, n, orig, k, eps, out, by_w);
    if (k <= 16) return launch_cov<16>(ctx, g, q, n, orig, k, eps, out, by_w);
    if (k <= 20) return launch_cov<20>(ctx, g, q, n, orig, k, eps, out, by_w);
    if (k <= 24) return launch_cov<24>(ctx, g, q, n, orig, k, eps, out, by_w);
    return launch_cov<32>(ctx, g, q, n, orig, k, eps, out, by_w);
}

static int compute_covariances(wm_ctx *ctx, int k, double eps) {
    if (k > 32) return WM_ERR_ARG;
    const bool same = ctx->gicp_cov_k == k && ctx->gicp_cov_eps == eps;
    // both clouds new (the usual case): the target's pass runs on the side stream while this one
    // builds the source's grid (a dozen small, host-bound launches) and starts the source's pass
    hipStream_t main_stream = ctx->stream;
    bool forked = false;
    if (!(ctx->gicp_cov_tgt_valid && same)) {
        // target: neighbours from the level-0 search grid
        if (!ctx->levels[0].built) WM_TRY(ensure_levels(ctx, -1.0));
        WM_HIP(ctx, ctx->gicp_c2.reserve((ctx->n_tgt_input > 0 ? ctx->n_tgt_input : 1) * 9 * sizeof(double)));
        if (ctx->tune_two_streams && ctx->side_stream && !(ctx->gicp_cov_src_valid && same)) {
            WM_HIP(ctx, hipEventRecord(ctx->ev_fork, main_stream));
            WM_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
            ctx->stream = ctx->side_stream;
            forked = true;
        }
        const GridDev &g = ctx->levels[0].d;
        const float4 *orig = ctx->tgt_orig.as<float4>();
        // queries in the grid's own (cell-sorted) order -- a wave's 64 queries then scan the same
        // few rows of cells -- with the result stored under each point's caller index (.w).
        // Only finite points are in the grid; a cloud with non-finite ones takes caller order,
        // which also writes their (never read) placeholder covariances.
        const int rc = ctx->n_tgt == ctx->n_tgt_input
                           ? launch_cov_k(ctx, g, g.pts, ctx->n_tgt, orig, k, eps, ctx->gicp_c2.as<double>(), 1)
                           : launch_cov_k(ctx, g, orig, ctx->n_tgt_input, orig, k, eps, ctx->gicp_c2.as<double>(), 0);
        if (forked) {  // (back on the main stream whatever happened)
            const hipError_t e = hipEventRecord(ctx->ev_join, ctx->side_stream);
            ctx->stream = main_stream;
            WM_HIP(ctx, e);
        }
        WM_TRY(rc);
        ctx->gicp_cov_tgt_valid = true;
    }
    if (!(ctx->gicp_cov_src_valid && same)) {
        // source: its own grid; covariances stored in Morton (src_sorted) order
        WM_HIP(ctx, ctx->gicp_c1.reserve((ctx->n_src > 0 ? ctx->n_src : 1) * 9 * sizeof(double)));
        double occ = 0, vol = 1;
        for (int d = 0; d < 3; ++d) vol *= fmax((double) ctx->src_bbox.hi[d] - ctx->src_bbox.lo[d], 1e-3);
        float h = choose_cell(ctx->src_bbox, ctx->n_src);
        // a source like the previous one (consecutive scans of one sensor): start from the cell
        // size that was tuned for it, which usually passes the occupancy check at once and saves
        // the second build (the level-0 grid of the target does the same)
        bool trust = false;  // a cloud this close to the one the cell size was measured on: no occupancy check
                             // (a device -> host round trip) -- but look again every 16th time, as build_level0 does
        if (ctx->tuned_src_h > 0 && ctx->tuned_src_n > 0) {
            const double rn = (double) ctx->n_src / (double) ctx->tuned_src_n, rv = vol / ctx->tuned_src_vol;
            if (rn > 0.8 && rn < 1.25 && rv > 0.6 && rv < 1.6) h = (float) ctx->tuned_src_h;
            trust = rn > 0.9 && rn < 1.1 && rv > 0.8 && rv < 1.25 && ++ctx->tuned_src_uses < 16;
        }
        if (!trust) ctx->tuned_src_uses = 0;
        WM_TRY(build_grid_level(ctx, ctx->src_orig.as<float4>(), ctx->n_src_input, ctx->src_bbox, h,
                                &ctx->src_grid, trust ? nullptr : &occ));
        if (occ > 6.0 || (occ > 0 && occ < 1.5)) {
            h = (float) (h * sqrt(3.0 / occ));
            WM_TRY(build_grid_level(ctx, ctx->src_orig.as<float4>(), ctx->n_src_input, ctx->src_bbox, h,
                                    &ctx->src_grid, nullptr));
        }
        ctx->tuned_src_h = h;
        ctx->tuned_src_n = ctx->n_src;
        ctx->tuned_src_vol = vol;
        const float4 *q = ctx->src_sorted.as<float4>();
        WM_TRY(launch_cov_k(ctx, ctx->src_grid.d, q, ctx->n_src, ctx->src_orig.as<float4>(), k, eps,
                            ctx->gicp_c1.as<double>(), 0));
        ctx->gicp_cov_src_valid = true;
    }
    if (forked) WM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    ctx->gicp_cov_k = k;
    ctx->gicp_cov_eps = eps;
    return WM_OK;
}

}  // namespace wm

using namespace wm;

extern "C" {

void wm_gicp_default_params(wm_gicp_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->corr_rand = 10;       // gicp.hpp:34
    p->max_iter = 100;       // gicp.hpp:35
    p->r_eps = 1e-8;         // gicp.hpp:36
    p->t_eps = 5e-4;         // PCL GICP transformation_epsilon_ (libwave never sets it)
    p->max_corr = 5.0;       // PCL GICP corr_dist_threshold_  (libwave never sets it)
    p->gicp_epsilon = 1e-3;  // PCL GICP gicp_epsilon_
    p->max_inner = 20;       // max_inner_iterations_
}

int wm_gicp_covariances(wm_ctx *ctx, int k, double eps, double *cov_source, double *cov_target) {
    if (!ctx || k < 1 || k > 32) return WM_ERR_ARG;
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx));
    if ((size_t) k > ctx->n_src || (size_t) k > ctx->n_tgt) return WM_NOT_CONVERGED;
    WM_TRY(compute_covariances(ctx, k, eps));
    if (cov_target)
        WM_TRY(copy_to_caller(ctx, cov_target, ctx->gicp_c2.p, ctx->n_tgt_input * 9 * sizeof(double)));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cov_source) {
        // un-permute from Morton order to caller order
        std::vector<double> tmp(ctx->n_src * 9);
        std::vector<float> pts(ctx->n_src * 4);
        WM_HIP(ctx, hipMemcpy(tmp.data(), ctx->gicp_c1.p, tmp.size() * 8, hipMemcpyDeviceToHost));
        WM_HIP(ctx, hipMemcpy(pts.data(), ctx->src_sorted.p, pts.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ctx->n_src; ++i) {
            unsigned o;
            memcpy(&o, &pts[i * 4 + 3], 4);
            memcpy(cov_source + (size_t) o * 9, &tmp[i * 9], 9 * sizeof(double));
        }
    }
    return WM_OK;
}

int wm_gicp_align(wm_ctx *ctx, const wm_gicp_params *prm, double T_out[16], wm_gicp_stats *stats) {
    if (!ctx || !prm || !T_out || prm->corr_rand < 1 || prm->corr_rand > 32 || !(prm->max_corr > 0))
        return WM_ERR_ARG;
    if (prm->objective != WM_GICP_OBJECTIVE_PCL_SUMS && prm->objective != WM_GICP_OBJECTIVE_STATISTICS) return WM_ERR_ARG;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx, prm->max_corr, WM_NN_AUTO));
    // PCL: "Number of points in cloud is less than k_correspondences_" -> no alignment
    if ((size_t) prm->corr_rand > ctx->n_src || (size_t) prm->corr_rand > ctx->n_tgt) return WM_NOT_CONVERGED;
    WM_TRY(compute_covariances(ctx, prm->corr_rand, prm->gicp_epsilon));
    const size_t n = ctx->n_src;
    if (prm->objective == WM_GICP_OBJECTIVE_PCL_SUMS) WM_HIP(ctx, ctx->gicp_mahal.reserve(n * 9 * sizeof(double)));
    WM_HIP(ctx, ctx->partials.reserve((size_t) kGicpBlocksMax * kQuadN * 2 * sizeof(double) + 64));
    const float thr = threshold_d2_strict(prm->max_corr);
    double base[16];
    mat4_identity(base);
    float T[16], prevT[16];
    for (int i = 0; i < 16; ++i) T[i] = prevT[i] = (i % 5 == 0) ? 1.f : 0.f;
    GicpFn F;
    F.ctx = ctx;
    F.base = base;
    const bool statistics = prm->objective != WM_GICP_OBJECTIVE_PCL_SUMS;
    GicpQuadFn Q;
    Q.base = base;
    const int max_it = prm->force_iterations > 0 ? prm->force_iterations : prm->max_iter;
    int iter = 0, inner_total = 0;
    bool converged = false;
    double f_last = 0;
    unsigned cnt = 0;
    const unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    while (!converged) {
        double Td[16];
        for (int i = 0; i < 16; ++i) Td[i] = (double) T[i];
        // 1-NN of every transformed source point, d2 < max_corr^2 (strict)
        // (the previous matches are worth starting from only if the last minimisation moved the cloud by
        // less than about a grid cell: a seed farther away than that makes the search scan a ball of
        // its radius, which costs more than finding the neighbour from scratch)
        bool seeded = iter > 0;
        if (seeded) {
            double dt2 = 0, dr2 = 0, rad2 = 0;
            for (int a = 0; a < 3; ++a) {
                const double d = (double) T[a * 4 + 3] - (double) prevT[a * 4 + 3];
                dt2 += d * d;
                for (int b = 0; b < 3; ++b) {
                    const double e = (double) T[a * 4 + b] - (double) prevT[a * 4 + b];
                    dr2 += e * e;
                }
                const double ext = fmax(fabs((double) ctx->src_bbox.lo[a]), fabs((double) ctx->src_bbox.hi[a]));
                rad2 += ext * ext;
            }
            const double cell = ctx->levels[0].built ? (double) ctx->levels[0].d.h : 0.0;
            seeded = sqrt(dt2) + sqrt(dr2 * rad2) < 0.75 * cell;
        }
        WM_TRY(nn_pass(ctx, Td, thr, prm->max_corr, seeded, false, 0.f, 0.f, /*wait=*/false));  // (the fetch below waits)
        int inner;
        double x[6] = {T[3], T[7], T[11], atan2(T[9], T[10]), asin(-T[8]), atan2(T[4], T[0])};
        if (statistics) {
            // ONE pass over the pairs: Mahalanobis matrices formed on the fly, 74 sums out; the minimisation's
            // evaluations are then scalar work here (wm_gicp_quad.hpp)
            WM_TRY(gicp_quad_statistics(ctx, T, Q, &F.kernel_ms));
            cnt = (unsigned) Q.m;
            memcpy(prevT, T, sizeof(T));
            inner = bfgs_minimize(Q, x, prm->max_inner, &f_last);
        } else {
            Mat3d R;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) R.m[a * 3 + b] = (double) T[a * 4 + b];
            hipLaunchKernelGGL(k_gicp_mahal, dim3(blocks), dim3(kBlock), 0, ctx->stream, (unsigned) n,
                               ctx->keys.as<unsigned long long>(), ctx->gicp_c1.as<double>(),
                               ctx->gicp_c2.as<double>(), R, ctx->gicp_mahal.as<double>());
            WM_TRY(count_matched(ctx, n, &cnt));
            memcpy(prevT, T, sizeof(T));
            F.m = (int) cnt;
            serve_begin(F);
            inner = bfgs_minimize(F, x, prm->max_inner, &f_last);
            serve_end(F);
            if (F.rc != WM_OK) return F.rc;
        }
        if (inner < 0) break;  // NotEnoughPointsException: loop breaks, converged_ stays false
        inner_total += inner;
        state_to_matrix_f(base, x, T);
        double delta = 0;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const double ratio = (a < 3 && b < 3) ? 1.0 / prm->r_eps : 1.0 / prm->t_eps;
                const double cd = ratio * fabs((double) prevT[a * 4 + b] - (double) T[a * 4 + b]);
                if (cd > delta) delta = cd;
            }
        ++iter;
        if (prm->force_iterations > 0 ? (iter >= max_it) : (iter >= max_it || delta < 1)) {
            converged = true;
            memcpy(prevT, T, sizeof(T));
        }
    }
    ctx->have_corr = true;
    ctx->last_align_valid = false;
    if (stats) {
        stats->converged = converged;
        stats->iterations = iter;
        stats->n_corr = (int) cnt;
        stats->inner_total = inner_total;
        stats->evaluations = F.evals + Q.evals;
        stats->f_final = f_last;
        stats->fdf_kernel_ms = F.kernel_ms;
        stats->served_evaluations = F.served_total;
    }
    if (!converged) return cnt < 4 ? WM_TOO_FEW_CORRESPONDENCES : WM_NOT_CONVERGED;
    for (int i = 0; i < 16; ++i) T_out[i] = (double) prevT[i];
    return WM_OK;
}

// One evaluation of the GICP objective and gradient (kernel-level parity): pairs and
// Mahalanobis matrices are formed with T_pair exactly as one outer iteration does, then
// f and its gradient are evaluated at state x on top of the identity base.
int wm_gicp_eval(wm_ctx *ctx, const wm_gicp_params *prm, const double T_pair[16], const double x[6],
                 double *f, double g[6], int *n_pairs) {
    if (!ctx || !prm || !T_pair || !x || !f) return WM_ERR_ARG;
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx, prm->max_corr, WM_NN_AUTO));
    if ((size_t) prm->corr_rand > ctx->n_src || (size_t) prm->corr_rand > ctx->n_tgt) return WM_NOT_CONVERGED;
    WM_TRY(compute_covariances(ctx, prm->corr_rand, prm->gicp_epsilon));
    const size_t n = ctx->n_src;
    WM_HIP(ctx, ctx->gicp_mahal.reserve(n * 9 * sizeof(double)));
    WM_HIP(ctx, ctx->partials.reserve((size_t) kGicpBlocksMax * kQuadN * 2 * sizeof(double) + 64));
    double Td[16];
    Mat3d R;
    for (int i = 0; i < 16; ++i) Td[i] = (double) (float) T_pair[i];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R.m[a * 3 + b] = Td[a * 4 + b];
    WM_TRY(nn_pass(ctx, Td, threshold_d2_strict(prm->max_corr), prm->max_corr, false));
    if (prm->objective != WM_GICP_OBJECTIVE_PCL_SUMS) {  // the statistics objective: the 74 sums under T_pair, evaluated at x
        double base[16];
        mat4_identity(base);
        GicpQuadFn Q;
        Q.base = base;
        float Tf[16];
        for (int i = 0; i < 16; ++i) Tf[i] = (float) T_pair[i];
        WM_TRY(gicp_quad_statistics(ctx, Tf, Q, nullptr));
        ctx->have_corr = true;
        ctx->last_align_valid = false;
        if (n_pairs) *n_pairs = Q.m;
        if (Q.m == 0) return WM_TOO_FEW_CORRESPONDENCES;
        double gg[6];
        *f = Q.fdf(x, gg);
        if (g) memcpy(g, gg, sizeof(gg));
        return WM_OK;
    }
    const unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    unsigned cnt = 0;
    hipLaunchKernelGGL(k_gicp_mahal, dim3(blocks), dim3(kBlock), 0, ctx->stream, (unsigned) n,
                       ctx->keys.as<unsigned long long>(), ctx->gicp_c1.as<double>(),
                       ctx->gicp_c2.as<double>(), R, ctx->gicp_mahal.as<double>());
    WM_TRY(count_matched(ctx, n, &cnt));
    ctx->have_corr = true;
    ctx->last_align_valid = false;
    if (n_pairs) *n_pairs = (int) cnt;
    if (cnt == 0) return WM_TOO_FEW_CORRESPONDENCES;
    double base[16];
    mat4_identity(base);
    GicpFn F;
    F.ctx = ctx;
    F.base = base;
    F.m = (int) cnt;
    double gg[6];
    *f = gicp_fdf(F, x, gg);
    if (g) memcpy(g, gg, sizeof(gg));
    return F.rc;
}

// GICPMatcher: setRef / setTarget (voxel filter when res > 0, gicp.cpp:37-55) + match
int wm_gicp_match(wm_ctx *ctx, const void *ref, size_t n_ref, const void *target, size_t n_target,
                  size_t stride, int mem, const wm_gicp_params *p, float res, double T_out[16],
                  wm_gicp_stats *stats) {
    if (!ctx || !p || !T_out || (n_ref > 0 && !ref) || (n_target > 0 && !target) || stride < 12 ||
        (stride & 3) || n_ref > 0x7FFFFFF0u || n_target > 0x7FFFFFF0u)
        return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (!(res > 0)) {
        WM_TRY(wm_set_source(ctx, ref, n_ref, stride, mem));
        WM_TRY(wm_set_target(ctx, target, n_target, stride, mem));
        return wm_gicp_align(ctx, p, T_out, stats);
    }
    WM_TRY(wm_set_source_filtered(ctx, ref, n_ref, stride, mem, res));
    WM_TRY(wm_set_target_filtered(ctx, target, n_target, stride, mem, res));
    return wm_gicp_align(ctx, p, T_out, stats);
}

// GICPMatcher::setRef / setTarget with res > 0 (gicp.cpp:38-45, 48-55): the cloud is voxel-filtered
// and the FILTERED copy becomes the registration's input at the time of the call (a snapshot: later
// changes of the caller's cloud are not seen).
static int set_filtered(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem, float leaf, bool source) {
    if (!ctx || (n > 0 && !pts) || stride < 12 || (stride & 3) || n > 0x7FFFFFF0u || !(leaf > 0)) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf &raw = source ? ctx->match_ref : ctx->match_tgt, &ds = source ? ctx->ds_ref : ctx->ds_tgt;
    const size_t cap = n > 0 ? n : 1;
    WM_HIP(ctx, raw.reserve(cap * sizeof(float4)));
    WM_HIP(ctx, ds.reserve(cap * sizeof(float4)));
    WM_TRY(pack_cloud(ctx, pts, n, stride, mem, raw.as<float4>()));
    size_t nf = 0;
    WM_TRY(voxel_downsample_dev(ctx, raw.as<float4>(), n, leaf, ds.as<float4>(), &nf));
    return source ? wm_set_source(ctx, ds.p, nf, sizeof(float4), WM_MEM_DEVICE)
                  : wm_set_target(ctx, ds.p, nf, sizeof(float4), WM_MEM_DEVICE);
}

int wm_set_source_filtered(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem, float leaf) {
    return set_filtered(ctx, pts, n, stride, mem, leaf, true);
}

int wm_set_target_filtered(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem, float leaf) {
    return set_filtered(ctx, pts, n, stride, mem, leaf, false);
}

}  // extern "C"
