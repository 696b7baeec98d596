// wm_icp.hip -- the ICP iteration on device + the C ABI entry points.
//
// One iteration of pcl::IterativeClosestPoint::computeTransformation
// [PCL registration/impl/icp.hpp], as driven by libwave's ICPMatcher::match()
// (wave_matching/src/icp.cpp:95,116,126), is three kernel classes:
//   1. correspondence search (wm_nn.hip)                 -> 8-byte key per source point
//   2. k_icp_stats: streaming reduction of the matched pairs to 17 doubles
//      per workgroup (n, sum p, sum q, sum q p^T, sum d2  |  GN: n, sum p,
//      A^T A, J^T r, sum d2)
//   3. k_reduce_solve: fixed-order sum of the partials, the 3x3 SVD (Umeyama) or
//      6x6 solve (Gauss-Newton), T <- T_k T, and PCL's DefaultConvergenceCriteria,
//      all by one workgroup, so the whole registration runs with the host out of
//      the loop.  In the multi-GPU path the 32-double statistics block is what
//      the RCCL all-reduce carries between (2) and (3).
#include "wm_internal.hpp"
#include "wm_icp_step.hpp"
#include "wm_bins.hpp"
#include "wm_xchg.hpp"

#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include <float.h>
#include <math.h>
#include <string.h>

#include <stdlib.h>

#include <new>

namespace wm {

constexpr int kMaxStatBlocks = 256;
static void set_step_scale(wm_ctx *ctx);
constexpr int kStatUnroll = 4;  // points per thread per trip of the statistics kernel

__device__ __forceinline__ void xform_pt(const float *T, const float4 &p, float &x, float &y,
                                         float &z) {
    x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], p.x), __fmul_rn(T[1], p.y)),
                            __fmul_rn(T[2], p.z)), T[3]);
    y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], p.x), __fmul_rn(T[5], p.y)),
                            __fmul_rn(T[6], p.z)), T[7]);
    z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], p.x), __fmul_rn(T[9], p.y)),
                            __fmul_rn(T[10], p.z)), T[11]);
}

// Per-workgroup partial sums, written as partials[block][kAcc].
template <int MODE>
__global__ void __launch_bounds__(kBlock)
    k_icp_stats(const float4 *__restrict__ src, unsigned n,
                const unsigned long long *__restrict__ keys, const float4 *__restrict__ tgt,
                const IcpDevState *__restrict__ st, double *__restrict__ partials) {
    if (st->done) return;
    double a[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) a[k] = 0.0;
    const bool slab_on = st->slab_on != 0;
    const float slab_lo = st->slab_lo, slab_hi = st->slab_hi;
    // kStatUnroll points per trip: all their loads are issued before the first use, so a wave keeps
    // ~10 KB in flight (the kernel is a pure HBM/L2 stream); accumulation order is unchanged
    const unsigned stride = gridDim.x * kBlock;
    for (unsigned i0 = blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kStatUnroll * stride) {
        float4 p4v[kStatUnroll], q4v[kStatUnroll];
        unsigned long long keyv[kStatUnroll];
#pragma unroll
        for (int u = 0; u < kStatUnroll; ++u) {
            const unsigned i = i0 + u * stride;
            const unsigned ic = i < n ? i : i0;
            p4v[u] = src[ic];
            keyv[u] = keys[ic];
            q4v[u] = tgt[ic];  // match coordinates, written by the search (coalesced)
        }
#pragma unroll
        for (int u = 0; u < kStatUnroll; ++u) {
            if (i0 + u * stride >= n) break;
            float fx, fy, fz;
            xform_pt(st->Tf, p4v[u], fx, fy, fz);
            // sharded registration: the same ownership test as the search kernel
            if (slab_on && !(fx >= slab_lo && fx < slab_hi)) continue;
            a[17] += 1.0;
            const unsigned long long key = keyv[u];
            const unsigned idx = (unsigned) key;
            if (idx == kNoIdx) continue;
            const float4 q4 = q4v[u];
            const double px = fx, py = fy, pz = fz, qx = q4.x, qy = q4.y, qz = q4.z;
            const double d2 = (double) __uint_as_float((unsigned) (key >> 32));
            a[0] += 1.0;
            a[1] += px;
            a[2] += py;
            a[3] += pz;
            if (MODE == WM_ICP_SVD) {
                a[4] += qx;
                a[5] += qy;
                a[6] += qz;
                a[7] += qx * px;
                a[8] += qx * py;
                a[9] += qx * pz;
                a[10] += qy * px;
                a[11] += qy * py;
                a[12] += qy * pz;
                a[13] += qz * px;
                a[14] += qz * py;
                a[15] += qz * pz;
            } else {
                const double rx = px - qx, ry = py - qy, rz = pz - qz;
                a[4] += py * py + pz * pz;  // (A^T A)(0,0)
                a[5] += -px * py;           // (0,1)
                a[6] += -px * pz;           // (0,2)
                a[7] += px * px + pz * pz;  // (1,1)
                a[8] += -py * pz;           // (1,2)
                a[9] += px * px + py * py;  // (2,2)
                a[10] += rx;
                a[11] += ry;
                a[12] += rz;
                a[13] += py * rz - pz * ry;  // p x r
                a[14] += pz * rx - px * rz;
                a[15] += px * ry - py * rx;
            }
            a[16] += d2;
    }
    }
    // wave reduction (fixed xor-tree order), then across the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < kAcc; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    __shared__ double lds[kBlock / 64][kAcc];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < kAcc; ++k) lds[wave][k] = a[k];
    __syncthreads();
    if (threadIdx.x < kAcc) {
        double s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += lds[w][threadIdx.x];
        partials[(size_t) blockIdx.x * kAcc + threadIdx.x] = s;
    }
}

// Pre-reduction for very many partial rows (fused statistics of clouds beyond ~1M points per GPU,
// or one-wave workgroups): block b adds rows [128 b, 128 b + 128) in a fixed order -> out row b.
constexpr int kPreRows = 128;
__global__ void __launch_bounds__(kBlock)
    k_reduce_rows(const double *__restrict__ partials, int rows, const IcpDevState *__restrict__ st,
                  double *__restrict__ out) {
    if (st->done) return;
    constexpr int kLanes = kBlock / kAcc;  // 14 row-lanes x kAcc components
    __shared__ double lds[kLanes][kAcc];
    const int c = threadIdx.x % kAcc, r = threadIdx.x / kAcc;
    const int r0 = blockIdx.x * kPreRows, r1 = min(r0 + kPreRows, rows);
    if (r < kLanes) {
        double v[(kPreRows + kLanes - 1) / kLanes];
        int k = 0;
#pragma unroll
        for (int b = r0 + r, u = 0; u < (kPreRows + kLanes - 1) / kLanes; b += kLanes, ++u, ++k)
            v[u] = b < r1 ? partials[(size_t) b * kAcc + c] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < (kPreRows + kLanes - 1) / kLanes; ++u) s += v[u];
        lds[r][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < kAcc) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < kLanes; ++l) t += lds[l][threadIdx.x];
        out[(size_t) blockIdx.x * kAcc + threadIdx.x] = t;
    }
}

// What the host steers by while it runs ahead of the device (wm_icp_align): one 8-byte word in pinned memory -- done
// flag, iterations finished, the step's size -- in ONE system-scope store (pub[0]: the latest; pub[k]: iteration k's
// own record, so that what the host decides from does not depend on when it looks).
__device__ __forceinline__ void publish_step(const IcpDevState *s, unsigned long long *pub, int pub_slots) {
    if (!pub) return;
    // [iteration : 16 | step size as bfloat16 : 16 | changed matches : 16 | searched by the certificate kernel : 16]
    // -- fractions in 1 / 65535
    const unsigned f_ch = (unsigned) (fminf(fmaxf(s->frac_changed, 0.f), 1.f) * 65535.f + 0.5f);
    const unsigned f_un = (unsigned) (fminf(fmaxf(s->frac_unsettled, 0.f), 1.f) * 65535.f + 0.5f);
    const unsigned long long w = ((unsigned long long) ((unsigned) s->iter & 0xFFFFu) << 48) |
                                 ((unsigned long long) (__float_as_uint(s->step_disp) >> 16) << 32) |
                                 ((unsigned long long) f_ch << 16) | (unsigned long long) f_un;
    if (s->iter >= 1 && s->iter <= pub_slots) __hip_atomic_store(pub + s->iter, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ([0]: bit 0 = done, above it the number of iterations finished by then -- ONE word, so that a host that sees
    // `done` before the last record knows whether that record is still to come)
    __hip_atomic_store(pub, s->done ? (1ull | ((unsigned long long) (unsigned) s->iter << 1)) : 0ull, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// The iteration's solve from the BINS the search kernel's waves added their sums into (wm_bins.hpp): the unsharded
// loop's replacement for k_reduce_rows + k_reduce_solve<3>.  One workgroup: thread (g, j) adds word j (limb, component)
// of bins g, g + kGroups, ... -- integers: exact in any order --, the kGroups partial totals meet in LDS, one thread per
// component turns its three limb totals back into a double (bins_value), thread 0 runs the solve, PCL's stopping rules
// and publishes the record exactly as k_reduce_solve does; the words read are set back to zero for the next iteration.
__global__ void __launch_bounds__(kBlock)
    k_bins_solve(long long *__restrict__ bins, IcpDevState *st, unsigned long long *pub, int pub_slots) {
    __shared__ IcpDevState s_st;
    __shared__ BinsLds s_b;
    const unsigned long long t_start = clock64();
    static_assert(sizeof(IcpDevState) % 4 == 0, "word-wise staging");
    constexpr unsigned kWords = sizeof(IcpDevState) / 4;
    constexpr unsigned kStage = (kWords + kBlock - 1) / kBlock;
    unsigned stage[kStage];
#pragma unroll
    for (unsigned k = 0; k < kStage; ++k) {  // (the state's loads and the bins' in ONE round trip)
        const unsigned w = threadIdx.x + k * kBlock;
        stage[k] = w < kWords ? reinterpret_cast<const unsigned *>(st)[w] : 0u;
    }
    bins_collect<kBlock>(bins, s_b);
#pragma unroll
    for (unsigned k = 0; k < kStage; ++k) {
        const unsigned w = threadIdx.x + k * kBlock;
        if (w < kWords) reinterpret_cast<unsigned *>(&s_st)[w] = stage[k];
    }
    __syncthreads();
    if (s_st.done) return;  // (uniform; a launch queued behind a `done`: the bins were all zero and stay so)
    if (threadIdx.x == 0) {
        s_st.dbg[0] = t_start;
        s_st.dbg[1] = clock64();
        double a[kAcc], ex[kStatsLen];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) a[k] = s_b.poison ? 0.0 : s_b.tot[k];  // (poisoned: "no correspondences", icp_apply_stats)
        expand_stats(s_st.mode, a, ex, s_st.changed_mask);
        s_st.local_handled = ex[kStatsLen - 1];
#pragma unroll
        for (int k = 0; k < kStatsLen; ++k) s_st.stats[k] = ex[k];
        s_st.dbg[2] = clock64();
        icp_apply_stats(&s_st, ex, (long long) s_b.tot[kAcc]);
        publish_step(&s_st, pub, pub_slots);
        s_st.dbg[3] = clock64();
    }
    __syncthreads();
    for (unsigned w = threadIdx.x; w < kWords; w += kBlock)
        reinterpret_cast<unsigned *>(st)[w] = reinterpret_cast<const unsigned *>(&s_st)[w];
}

// PHASE 1: sum partials -> st->stats.   PHASE 2: solve + criteria from st->stats.
// Single GPU launches <1|2>; the sharded path launches <1>, all-reduces
// st->stats over RCCL, then launches <2>.
// THREADS = 256 for the few rows of k_icp_stats, 1024 for the thousands of rows the fused search
// kernel leaves (one row per workgroup).
// PHASES == 7 (sharded loop, mailboxes available): 1, then the exchange of the block with the other ranks INSIDE this
// kernel (wm_xchg.hpp), then 2 -- one launch per iteration where <1>, ncclAllReduce, <2> are three.
template <int PHASES, int THREADS>
__global__ void __launch_bounds__(THREADS)
    k_reduce_solve(const double *__restrict__ partials, int nblocks, IcpDevState *st,
                   double *stats_io, unsigned long long *pub, int pub_slots, int blk_ext, XchgDev xd, long long *bins) {
    // (bins != nullptr, phases with bit 0: this rank's sums come out of the iteration's bins -- exact integer limbs the
    // search kernel's waves added into, wm_bins.hpp -- instead of rows of partial sums: no k_reduce_rows in front)
    // (blk_ext: stats_io is the sharded loop's kBlkLen block, not a caller's WM_STATS_LEN one)
    // The solve runs in ONE lane and touches two dozen fields of the state: read from HBM one
    // dependent access at a time that is most of this kernel's ~10 us.  So the whole state is
    // staged in LDS by all threads (one round trip), worked on there, and written back whole.
    __shared__ IcpDevState s_st;
    const unsigned long long t_start = clock64();
    static_assert(sizeof(IcpDevState) % 4 == 0, "word-wise staging");
    constexpr unsigned kWords = sizeof(IcpDevState) / 4;
    // (the state's loads are issued here and land in LDS after the rows' loads have been issued
    // too: one memory round trip for both, not two)
    constexpr unsigned kStage = (kWords + THREADS - 1) / THREADS;
    unsigned stage[kStage];
#pragma unroll
    for (unsigned k = 0; k < kStage; ++k) {
        const unsigned w = threadIdx.x + k * THREADS;
        stage[k] = w < kWords ? reinterpret_cast<const unsigned *>(st)[w] : 0u;
    }
    constexpr int kRows = THREADS / kAcc;  // row-lanes x kAcc components <= THREADS threads
    constexpr int kGroups = 8;
    __shared__ double lds[kRows][kAcc];
    __shared__ double lds2[kGroups][kAcc];
    __shared__ double tot[kAcc];
    __shared__ BinsLds s_b;
    const bool from_bins = (PHASES & 1) && bins != nullptr;
    if (from_bins) {
        bins_collect<THREADS>(bins, s_b);
    } else if (PHASES & 1) {
        // thread (r, c) adds rows r, r + kRows, ... of column c: a wave reads 64 consecutive
        // doubles per load; 16 independent accumulators keep 16 loads in flight (every dependent
        // load -> add would cost a memory latency)
        const int c = threadIdx.x % kAcc, r = threadIdx.x / kAcc;
        if (r < kRows) {
            constexpr int U = 16;
            double s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] = 0.0;
            for (int b = r; b < nblocks; b += U * kRows) {  // every batch: U loads, then U adds
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int bb = b + u * kRows;
                    v[u] = bb < nblocks ? partials[(size_t) bb * kAcc + c] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] += v[u];
            }
#pragma unroll
            for (int w = U / 2; w > 0; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) s[u] += s[u + w];
            lds[r][c] = s[0];
        }
    }
#pragma unroll
    for (unsigned k = 0; k < kStage; ++k) {
        const unsigned w = threadIdx.x + k * THREADS;
        if (w < kWords) reinterpret_cast<unsigned *>(&s_st)[w] = stage[k];
    }
    if (from_bins) {
        if (threadIdx.x < (unsigned) kAcc) tot[threadIdx.x] = s_b.poison ? 0.0 : s_b.tot[threadIdx.x];
    } else if (PHASES & 1) {
        const int c = threadIdx.x % kAcc;
        __syncthreads();
        if (threadIdx.x < kGroups * kAcc) {  // row-lanes g, g + 8, ... of column c
            const int g = threadIdx.x / kAcc;
            double t = 0.0;
            for (int sl = g; sl < kRows; sl += kGroups) t += lds[sl][c];
            lds2[g][c] = t;
        }
        __syncthreads();
        if (threadIdx.x < kAcc) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < kGroups; ++g) t += lds2[g][threadIdx.x];
            tot[threadIdx.x] = t;
        }
    }
    __syncthreads();
    if (s_st.done) return;  // (uniform: every thread reads the staged copy)
    // the certificate kernel's 64 partial counts of searched queries: added (and zeroed) by the first
    // wave here, not one LDS round trip after the other by the lane that solves
    __shared__ unsigned s_uns;
    if (threadIdx.x < 64) {
        unsigned v = s_st.cert_unsettled[threadIdx.x];
        s_st.cert_unsettled[threadIdx.x] = 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += (unsigned) __shfl_xor((int) v, off);
        if (threadIdx.x == 0) s_uns = from_bins ? (unsigned) s_b.tot[kAcc] : v;  // (bins: the count is one of their components)
    }
    if constexpr (PHASES == 7) {
        // this rank's block -> every rank's mailbox; every rank's block -> the sum, in rank order (all threads)
        __shared__ double s_blk[kBlkLen], s_sum[kBlkLen];
        __shared__ unsigned s_half[kXMaxWorld * kXWords], s_ctl[2];
        if (threadIdx.x == 0) {
            double a[kAcc];
#pragma unroll
            for (int k = 0; k < kAcc; ++k) a[k] = tot[k];
            double ex[kStatsLen];
            expand_stats(s_st.mode, a, ex, s_st.changed_mask);
            s_st.local_handled = ex[kStatsLen - 1];
            ex[kStatsLen - 2] = s_st.stripe_finite;  // (as in <1>: summed, it is the cloud's count)
#pragma unroll
            for (int k = 0; k < kStatsLen; ++k) s_blk[k] = ex[k];
            s_blk[kStatsLen] = (double) s_uns;
            s_blk[kStatsLen + 1] = 0.0;
        }
        __syncthreads();
        // ([kStatsLen + 1]: 0 in an iteration's block, 1 in the block of a rank's COMMIT round (k_xchg_commit): a rank
        // that gave up on an earlier round is a round behind and sends its commit where the others expect an
        // iteration -- they then fail at once instead of solving from it)
        const bool arrived = xchg_allreduce<THREADS>(xd, s_blk, s_sum, s_half, s_ctl) && s_sum[kStatsLen + 1] == 0.0;
        if (threadIdx.x == 0) {
            s_st.dbg[0] = t_start;
            s_st.dbg[1] = clock64();
            if (!arrived) {  // a peer never delivered (or has given up): the registration ends here, on this rank, with an error
                s_st.xchg_failed = 1;
                s_st.done = 1;
                s_st.converged = 0;
                if (pub) __hip_atomic_store(pub, 1ull | ((unsigned long long) (unsigned) s_st.iter << 1), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                double stats[kStatsLen];
#pragma unroll
                for (int k = 0; k < kStatsLen; ++k) stats[k] = s_st.stats[k] = s_sum[k];
                s_st.uns_global = 1;
                s_st.dbg[2] = clock64();
                icp_apply_stats(&s_st, stats, (long long) s_sum[kStatsLen]);
                if (stats_io)  // (the reduced block, for a caller that looks: wm_icp_shard_* / tests)
                    for (int k = 0; k < kBlkLen; ++k) stats_io[k] = s_sum[k];
                publish_step(&s_st, pub, pub_slots);
            }
            s_st.dbg[3] = clock64();
        }
    } else if (threadIdx.x == 0) {
        s_st.dbg[0] = t_start;
        s_st.dbg[1] = clock64();  // state staged, rows added
        if (PHASES & 1) {
            double a[kAcc];
#pragma unroll
            for (int k = 0; k < kAcc; ++k) a[k] = tot[k];
            double ex[kStatsLen];
            expand_stats(s_st.mode, a, ex, s_st.changed_mask);
            s_st.local_handled = ex[kStatsLen - 1];
#pragma unroll
            for (int k = 0; k < kStatsLen; ++k) s_st.stats[k] = ex[k];
            if (PHASES == 1 && stats_io) {  // the block the all-reduce works on
                ex[kStatsLen - 2] = s_st.stripe_finite;  // (a free slot of either layout: summed, it is the cloud's count)
#pragma unroll
                for (int k = 0; k < kStatsLen; ++k) stats_io[k] = ex[k];
                // (the searches this rank's certificate launch made: summed over the ranks, so that every
                // rank's host steers by the same share and picks the same kernel)
                if (blk_ext) {
                    stats_io[kStatsLen] = (double) s_uns;
                    stats_io[kStatsLen + 1] = 0.0;
                }
            }
        } else if (stats_io) {  // PHASES == 2: the all-reduced block comes in
#pragma unroll
            for (int k = 0; k < kStatsLen; ++k) s_st.stats[k] = stats_io[k];
            s_st.uns_global = blk_ext ? 1 : 0;
        }
        if (PHASES & 2) {
            double stats[kStatsLen];
#pragma unroll
            for (int k = 0; k < kStatsLen; ++k) stats[k] = s_st.stats[k];
            s_st.dbg[2] = clock64();
            const long long uns_total = (PHASES == 2 && stats_io && blk_ext) ? (long long) stats_io[kStatsLen] : (long long) s_uns;
            icp_apply_stats(&s_st, stats, uns_total);
            // what the host steers by while it runs ahead of the device (wm_icp_align): one 8-byte word in
            // pinned memory -- done flag, iterations finished, the step's size -- in ONE system-scope store
            // (pub[0]: the latest; pub[k]: iteration k's own record, so that what the host decides from
            // does not depend on when it looks)
            publish_step(&s_st, pub, pub_slots);
        }
        s_st.dbg[3] = clock64();
    }
    __syncthreads();
    for (unsigned w = threadIdx.x; w < kWords; w += THREADS)
        reinterpret_cast<unsigned *>(st)[w] = reinterpret_cast<const unsigned *>(&s_st)[w];
}

// The COMMIT round of a sharded registration whose exchange ran through the mailboxes (wm_xchg.hpp): after the last
// iteration every rank sends {did every round of mine arrive in time?, 1} and adds up what the others sent.  A rank
// whose wait timed out in the LAST executed round would otherwise end with an error while a slow peer that still got
// every block ends well -- and the next registration would find one of them in ncclAllReduce and the other polling its
// mailbox.  With this round the verdict is the same on every rank: all of them ended well, or all of them fail this
// registration (and all of them leave the mailboxes for the collective, wm_shard.hip).
__global__ void __launch_bounds__(kBlock) k_xchg_commit(IcpDevState *st, XchgDev xd) {
    __shared__ double s_blk[kBlkLen], s_sum[kBlkLen];
    __shared__ unsigned s_half[kXMaxWorld * kXWords], s_ctl[2];
    if (threadIdx.x < (unsigned) kBlkLen) s_blk[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_blk[0] = st->xchg_failed ? 0.0 : 1.0;
        s_blk[kStatsLen + 1] = 1.0;
    }
    __syncthreads();
    const bool arrived = xchg_allreduce<kBlock>(xd, s_blk, s_sum, s_half, s_ctl);
    if (threadIdx.x == 0 && (!arrived || s_sum[0] != (double) xd.world || s_sum[kStatsLen + 1] != (double) xd.world)) {
        st->xchg_failed = 1;
        st->converged = 0;
    }
}

// keys (source-sorted order) -> caller-order (match index, d2)
__global__ void __launch_bounds__(kBlock)
    k_unpack_corr(const float4 *__restrict__ src, unsigned n,
                  const unsigned long long *__restrict__ keys, int *__restrict__ match,
                  float *__restrict__ d2) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned orig = __float_as_uint(src[i].w);
    const unsigned long long key = keys[i];
    const unsigned idx = (unsigned) key;
    match[orig] = idx == kNoIdx ? -1 : (int) idx;
    d2[orig] = __uint_as_float((unsigned) (key >> 32));
}

// a plain float4 copy: what this GPU's HBM delivers to a streaming kernel (read + write).  NT:
// four loads in flight per lane, non-temporal both ways (scripts/dev/copy_probe.hip: which shape
// wins varies from box to box by ~10 %, so wm_debug_copy_bandwidth reports the best of three)
typedef float copy_f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) k_copy_f4(const copy_f4v *__restrict__ a, copy_f4v *__restrict__ b, size_t n) {
    const size_t stride = (size_t) gridDim.x * 256u;
    size_t i = (size_t) blockIdx.x * 256u + threadIdx.x;
    if constexpr (NT) {
        for (; i + 3 * stride < n; i += 4 * stride) {
            copy_f4v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(a + i + u * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], b + i + u * stride);
        }
    }
    for (; i < n; i += stride) b[i] = a[i];
}

// ------------------------------------------------------------------ host
static int stat_blocks(size_t n) {
    size_t b = (n + kBlock - 1) / kBlock;
    if (b > kMaxStatBlocks) b = kMaxStatBlocks;
    if (b < 1) b = 1;
    return (int) b;
}

static int launch_stats(wm_ctx *ctx, int mode) {
    const unsigned n = (unsigned) ctx->n_src;
    const int nb = stat_blocks(n);
    const IcpDevState *st = ctx->d_state.as<IcpDevState>();
    if (mode == WM_ICP_SVD)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_icp_stats<WM_ICP_SVD>), dim3(nb), dim3(kBlock), 0,
                           ctx->stream, ctx->src_sorted.as<float4>(), n,
                           ctx->keys.as<unsigned long long>(), ctx->match_pt.as<float4>(), st,
                           ctx->partials.as<double>());
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_icp_stats<WM_ICP_GN6>), dim3(nb), dim3(kBlock), 0,
                           ctx->stream, ctx->src_sorted.as<float4>(), n,
                           ctx->keys.as<unsigned long long>(), ctx->match_pt.as<float4>(), st,
                           ctx->partials.as<double>());
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// the iteration's bins (wm_bins.hpp): allocated, and all zero -- zeroed here when first allocated or when the last loop
// that used them did not end normally; kept at zero by k_bins_solve otherwise
int bins_ready(wm_ctx *ctx) {
    if (!ctx->bins.p || ctx->bins_dirty) {
        WM_HIP(ctx, ctx->bins.reserve(kBinWords * sizeof(long long)));
        WM_HIP(ctx, hipMemsetAsync(ctx->bins.p, 0, kBinWords * sizeof(long long), ctx->stream));
        ctx->bins_dirty = false;
    }
    return WM_OK;
}

static int launch_bins_solve(wm_ctx *ctx, unsigned long long *pub, int pub_slots) {
    hipLaunchKernelGGL(k_bins_solve, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->bins.as<long long>(),
                       ctx->d_state.as<IcpDevState>(), pub, pub_slots);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// Sum `rows` partial rows (ctx->partials) and run the requested phases of the iteration's solve.
template <int PHASES>
static int launch_reduce_solve(wm_ctx *ctx, unsigned rows, double *stats_io, unsigned long long *pub = nullptr,
                               int pub_slots = 0, int blk_ext = 0, const XchgDev *xchg = nullptr, long long *bins = nullptr) {
    if (bins) rows = 0;  // (the sums are in the bins: nothing to pre-reduce, the small instantiation)
    const XchgDev xd = xchg ? *xchg : XchgDev{nullptr, nullptr, 0, 0, 0u};
    IcpDevState *st = ctx->d_state.as<IcpDevState>();
    const double *part = ctx->partials.as<double>();
    if (rows > 2048u) {  // one workgroup cannot add that many rows quickly: 128 rows -> 1 first
        const unsigned rows2 = (rows + kPreRows - 1) / kPreRows;
        WM_HIP(ctx, ctx->partials2.reserve((size_t) rows2 * kAcc * sizeof(double)));
        hipLaunchKernelGGL(k_reduce_rows, dim3(rows2), dim3(kBlock), 0, ctx->stream, part, (int) rows, st,
                           ctx->partials2.as<double>());
        part = ctx->partials2.as<double>();
        rows = rows2;
    }
    if (rows > 512u)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce_solve<PHASES, 1024>), dim3(1), dim3(1024), 0, ctx->stream,
                           part, (int) rows, st, stats_io, pub, pub_slots, blk_ext, xd, bins);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce_solve<PHASES, kBlock>), dim3(1), dim3(kBlock), 0, ctx->stream,
                           part, (int) rows, st, stats_io, pub, pub_slots, blk_ext, xd, bins);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// one iteration's correspondence search + statistics on the grid: fused (the search kernel leaves
// the partial rows) or as two passes; *rows = partial rows to add up
static int launch_search_and_stats(wm_ctx *ctx, float thr, int mode, hipEvent_t e0, hipEvent_t e1,
                                   hipEvent_t e1b, unsigned *rows, bool use_bins = false) {
    if (ctx->tune_fuse_stats) return launch_nn_grid(ctx, thr, e0, e1, e1b, mode, rows, use_bins);
    WM_TRY(launch_nn_grid(ctx, thr, e0, e1, e1b));
    WM_TRY(launch_stats(ctx, mode));
    *rows = (unsigned) stat_blocks(ctx->n_src);
    return WM_OK;
}

static int prepare_work(wm_ctx *ctx) {
    const size_t n = ctx->n_src > 0 ? ctx->n_src : 1;
    WM_HIP(ctx, ctx->keys.reserve(n * sizeof(unsigned long long)));
    WM_HIP(ctx, ctx->match_pt.reserve(n * sizeof(float4)));
    // rows of the fused search + statistics kernel: one per workgroup (at most one per 64 queries, plus grid padding)
    WM_HIP(ctx, ctx->partials.reserve((n / 64 + 1024) * kAcc * sizeof(double)));
    WM_HIP(ctx, ctx->d_state.reserve(sizeof(IcpDevState)));
    if (!ctx->h_state)
        WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_state, sizeof(IcpDevState), hipHostMallocDefault));
    return WM_OK;
}

static void init_state(IcpDevState *s, const double *T, const wm_icp_params *p, double prev_mse) {
    memset(s, 0, sizeof(*s));
    for (int k = 0; k < 16; ++k) s->T[k] = T[k];
    for (int k = 0; k < 12; ++k) s->Tf[k] = s->Tf_search[k] = (float) T[k];
    mat4_identity(s->Tk);
    s->prev_mse = prev_mse;
    if (p) {
        s->forced = p->force_iterations > 0;
        s->max_iter = s->forced ? p->force_iterations : p->max_iter;
        s->mode = p->mode;
        s->rot_thr = 1.0 - p->t_eps;
        s->trans_thr = p->t_eps;
        s->fit_eps = p->fit_eps;
    }
}

static int upload_state(wm_ctx *ctx) {
    ctx->h_state->changed_mask = changed_mask_for((unsigned) ctx->n_src);
    WM_HIP(ctx, hipMemcpyAsync(ctx->d_state.p, ctx->h_state, sizeof(IcpDevState),
                               hipMemcpyHostToDevice, ctx->stream));
    return WM_OK;
}

// Small results the host has to wait for (bounding-box / occupancy partials, the iteration
// state, a voxel count, the GICP objective's sums) are produced in DEVICE memory and then
// fetched by ONE wavefront that copies them into pinned host memory, executes a system-scope
// fence in every lane, and only then raises the completion flag the host polls.  Anything
// weaker was seen to fail a few times in a hundred runs on some machines: a flag written by a
// later kernel (or a DMA copy followed by a signalling kernel) can reach host memory BEFORE
// data written by other compute units / engines, which travel other routes through the fabric
// -- the host then reads stale partials (a bounding-box count larger than the cloud, a stale
// voxel count) and the next kernel walks off the end of a buffer.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
    k_fetch_signal(unsigned *dst, const unsigned *src, unsigned words, unsigned *flag, unsigned seq) {
    if ((words & 3u) == 0 && (((size_t) dst | (size_t) src) & 15u) == 0) {
        const uint4 *s4 = (const uint4 *) src;
        uint4 *d4 = (uint4 *) dst;
        for (unsigned w = threadIdx.x; w < words / 4; w += THREADS) d4[w] = s4[w];
    } else {
        for (unsigned w = threadIdx.x; w < words; w += THREADS) dst[w] = src[w];
    }
    __threadfence_system();  // every lane: all of this wave's stores are performed system-wide
    // more than one wave (large fetches): each has fenced its own stores before it arrives here,
    // and the flag is written after all of them have
    if (THREADS > 64) __syncthreads();
    if (threadIdx.x == 0) *(volatile unsigned *) flag = seq;
}

static int wait_flag(wm_ctx *ctx, unsigned seq) {
    // Three stages.  Spin (most waits are tens of microseconds); then poll with a yield between
    // looks, so that worker threads sharing a core take turns instead of starving each other
    // (kernels of a few hundred microseconds: the NDT passes at 2M points); only then let the
    // runtime block -- its wake-up costs 0.1-0.2 ms on some hosts, which a registration that
    // waits a hundred times cannot afford, but it is the right thing for a wait of milliseconds
    // and it is what reports a failed kernel.
    volatile unsigned *flag = ctx->h_sig;
    const auto t0 = std::chrono::steady_clock::now();
    bool yielding = false;
    for (unsigned spins = 1; *flag != seq; ++spins) {
        if (yielding)
            std::this_thread::yield();
        else
            cpu_relax();
        if ((spins & 63u) == 0 || yielding) {
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited > std::chrono::milliseconds(4)) {
                WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
                break;
            }
            yielding = waited > std::chrono::microseconds(ctx->tune_spin_us);
        }
    }
    return WM_OK;
}

// wait until the n 16-byte slots {value, number} at `slots` (pinned memory) all carry `seq` (wait_flag's three stages:
// spin, poll with yields, and after 4 ms let the runtime block -- which is also what reports a failed kernel)
int wait_slots(wm_ctx *ctx, const double *slots, int n, unsigned seq) {
    const volatile unsigned *w = reinterpret_cast<const volatile unsigned *>(slots);
    auto all_there = [&]() {
        for (int k = n - 1; k >= 0; --k)
            if (w[4 * k + 2] != seq) return false;
        return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    bool yielding = false;
    for (unsigned spins = 1; !all_there(); ++spins) {
        if (yielding)
            std::this_thread::yield();
        else
            cpu_relax();
        if ((spins & 63u) == 0 || yielding) {
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited > std::chrono::milliseconds(4)) {
                WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (!all_there()) {
                    ctx->last_error = "the kernel ended without delivering its sums";
                    return WM_ERR_HIP;
                }
                break;
            }
            yielding = waited > std::chrono::microseconds(ctx->tune_spin_us);
        }
    }
    return WM_OK;
}

int fast_fetch_begin(wm_ctx *ctx, unsigned **flag, unsigned *seq) {
    if (!ctx->h_sig) {
        WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_sig, 64, hipHostMallocDefault));
        *ctx->h_sig = 0;
    }
    *flag = ctx->h_sig;
    *seq = ++ctx->sig_seq;
    return WM_OK;
}

int fast_fetch_wait(wm_ctx *ctx, unsigned seq) { return wait_flag(ctx, seq); }

int fast_fetch(wm_ctx *ctx, void *dst_pinned, const void *src_dev, size_t bytes) {
    if (bytes & 3) return WM_ERR_ARG;
    if (!ctx->h_sig) {
        WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_sig, 64, hipHostMallocDefault));
        *ctx->h_sig = 0;
    }
    const unsigned seq = ++ctx->sig_seq;
    if (bytes <= 4096)  // one wave: nothing to wait for but its own stores
        hipLaunchKernelGGL(k_fetch_signal<64>, dim3(1), dim3(64), 0, ctx->stream, (unsigned *) dst_pinned,
                           (const unsigned *) src_dev, (unsigned) (bytes / 4), ctx->h_sig, seq);
    else
        hipLaunchKernelGGL(k_fetch_signal<1024>, dim3(1), dim3(1024), 0, ctx->stream, (unsigned *) dst_pinned,
                           (const unsigned *) src_dev, (unsigned) (bytes / 4), ctx->h_sig, seq);
    WM_HIP(ctx, hipGetLastError());
    return wait_flag(ctx, seq);
}

// Column sums of a [rows][k] block of f64 partials (k <= 32), reduced ON THE DEVICE by one
// workgroup and delivered as k doubles: what the host needs from a GICP objective or an NDT
// derivative pass is the sum over blocks, and shipping every block's partials over PCIe to add
// them on the host cost more than the pass's own launch.  Fixed order, no atomics: thread t
// adds elements t, t + S, t + 2S, ... (S = the largest multiple of k <= 1024, so a thread stays
// in one column and a wave reads consecutive doubles), eight threads per column then add those
// partial sums group by group, one thread per column adds the eight.  The k results are written
// and fenced by lanes of wave 0, which also writes the flag (k_fetch_signal's rule).
__global__ void __launch_bounds__(1024)
    k_sum_fetch(double *dst, const double *__restrict__ src, unsigned rows, unsigned k, unsigned *flag,
                unsigned seq) {
    __shared__ double s1[1024];
    __shared__ double s2[8][32];
    const unsigned t = threadIdx.x;
    const unsigned groups = 1024u / k, stride = groups * k, total = rows * k;
    double a = 0.0;
    if (t < stride) {
        unsigned e = t;
        for (; e + 3 * stride < total; e += 4 * stride) {  // four loads in flight, added in order
            const double v0 = src[e], v1 = src[e + stride], v2 = src[e + 2 * stride], v3 = src[e + 3 * stride];
            a += v0;
            a += v1;
            a += v2;
            a += v3;
        }
        for (; e < total; e += stride) a += src[e];
    }
    s1[t] = a;
    __syncthreads();
    if (t < 8 * k) {
        const unsigned c = t % k, g = t / k;
        double b = 0.0;
        for (unsigned gg = g; gg < groups; gg += 8) b += s1[gg * k + c];
        s2[g][c] = b;
    }
    __syncthreads();
    if (t < k) {
        double r = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) r += s2[g][t];
        dst[t] = r;
    }
    if (t < 64 && flag) {
        __threadfence_system();
        if (t == 0) *(volatile unsigned *) flag = seq;
    }
}

int fast_fetch_sum(wm_ctx *ctx, double *dst_pinned, const double *src_dev, unsigned rows, unsigned k) {
    if (k < 1 || k > 32 || rows < 1) return WM_ERR_ARG;
    if (!ctx->h_sig) {
        WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_sig, 64, hipHostMallocDefault));
        *ctx->h_sig = 0;
    }
    const unsigned seq = ++ctx->sig_seq;
    hipLaunchKernelGGL(k_sum_fetch, dim3(1), dim3(1024), 0, ctx->stream, dst_pinned, src_dev, rows, k,
                       ctx->h_sig, seq);
    WM_HIP(ctx, hipGetLastError());
    return wait_flag(ctx, seq);
}

int sum_to_device(wm_ctx *ctx, double *dst_dev, const double *src_dev, unsigned rows, unsigned k) {
    if (k < 1 || k > 32 || rows < 1) return WM_ERR_ARG;
    hipLaunchKernelGGL(k_sum_fetch, dim3(1), dim3(1024), 0, ctx->stream, dst_dev, src_dev, rows, k,
                       (unsigned *) nullptr, 0u);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

static int download_state(wm_ctx *ctx) {
    return fast_fetch(ctx, ctx->h_state, ctx->d_state.p, sizeof(IcpDevState));
}

int sync_sleeping(wm_ctx *ctx) {
    // an event behind what is queued, looked at every ~50 us between short sleeps: a few per cent of a core
    // per waiting thread, and the wait ends within ~0.1 ms of the work (a BLOCKING event synchronise -- the
    // runtime's interrupt path -- was seen to add up to a millisecond per wait: 59 000 -> 48 500 pairs/s for a
    // single context's 256-pair batches)
    if (!ctx->ev_block && hipEventCreateWithFlags(&ctx->ev_block, hipEventDisableTiming) != hipSuccess) {
        (void) hipGetLastError();
        ctx->ev_block = nullptr;
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return WM_OK;
    }
    WM_HIP(ctx, hipEventRecord(ctx->ev_block, ctx->stream));
    for (;;) {
        const hipError_t e = hipEventQuery(ctx->ev_block);
        if (e == hipSuccess) return WM_OK;
        if (e != hipErrorNotReady) WM_HIP(ctx, e);
        std::this_thread::sleep_for(std::chrono::microseconds(40));
    }
}

int copy_to_caller(wm_ctx *ctx, void *dst, const void *src_dev, size_t bytes) {
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes) WM_HIP(ctx, hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost));
    return WM_OK;
}

void *pinned_scratch(wm_ctx *ctx, size_t bytes) {
    if (bytes < (64u << 10)) bytes = 64u << 10;
    if (ctx->h_scratch_bytes < bytes) {
        if (ctx->h_scratch) {
            (void) hipStreamSynchronize(ctx->stream);
            (void) hipHostFree(ctx->h_scratch);
        }
        ctx->h_scratch = nullptr;
        ctx->h_scratch_bytes = 0;
        if (hipHostMalloc(&ctx->h_scratch, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
        ctx->h_scratch_bytes = bytes;
    }
    return ctx->h_scratch;
}

static bool use_brute(const wm_ctx *ctx, int nn_method) {
    if (nn_method == WM_NN_BRUTE) return true;
    if (nn_method == WM_NN_GRID) return false;
    // all-pairs is cheaper than indexing below ~4M pair tests
    return (double) ctx->n_src * (double) ctx->n_tgt_input <= 4.0e6;
}

static hipEvent_t get_event(wm_ctx *ctx, size_t k) {
    while (ctx->ev_pool.size() <= k) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[k];
}

int nn_pass(wm_ctx *ctx, const double T[16], float thr_d2, double max_corr, bool predict, bool slab, float slab_lo,
            float slab_hi, bool wait) {
    WM_TRY(finalize_clouds(ctx, max_corr, WM_NN_AUTO));
    WM_TRY(prepare_work(ctx));
    const bool brute = use_brute(ctx, WM_NN_AUTO) || ctx->n_tgt == 0;
    if (!brute) WM_TRY(ensure_levels(ctx, max_corr));
    float keep_search[12];
    for (int k = 0; k < 12; ++k) keep_search[k] = ctx->h_state->Tf_search[k];
    init_state(ctx->h_state, T, nullptr, DBL_MAX);
    for (int k = 0; k < 12; ++k) ctx->h_state->Tf_search[k] = keep_search[k];  // (still what the align's keys refer to)
    ctx->h_state->have_prev = predict ? 1 : 0;
    if (slab) {  // a rank of a sharded registration searches the queries it owns under this pose
        ctx->h_state->slab_on = 1;
        ctx->h_state->slab_lo = slab_lo;
        ctx->h_state->slab_hi = slab_hi;
    }
    WM_TRY(upload_state(ctx));
    if (brute)
        WM_TRY(launch_nn_brute(ctx, thr_d2, nullptr, nullptr));
    else
        WM_TRY(launch_nn_grid(ctx, thr_d2, nullptr, nullptr, nullptr));
    // (wait = false: the caller queues more work behind the search and waits for that)
    if (wait) WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WM_OK;
}

int join_source_sort(wm_ctx *ctx) {
    if (!ctx->sort_join_pending) return WM_OK;
    ctx->sort_join_pending = false;
    WM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return WM_OK;
}

// the source's Morton sort that finalize_clouds (mode 2) left for later: on the side stream, behind ev_fork
int enqueue_deferred_sort(wm_ctx *ctx) {
    if (!ctx->sort_deferred) return WM_OK;
    ctx->sort_deferred = false;
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->side_stream;
    const int rc = morton_sort(ctx, ctx->src_orig.as<float4>(), ctx->n_src_input, ctx->src_bbox, ctx->n_src,
                               ctx->src_sorted.as<float4>());
    ctx->stream = main_stream;
    if (rc != WM_OK) return rc;
    WM_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
    ctx->sort_join_pending = true;
    return WM_OK;
}

int finalize_clouds(wm_ctx *ctx, double max_corr, int nn_method, int sort_aside) {
    WM_TRY(enqueue_deferred_sort(ctx));  // (left behind by a call that failed before it got there)
    WM_TRY(join_source_sort(ctx));  // (left behind by a call that failed before its own join)
    if (ctx->src_pending || ctx->tgt_pending) {
        // ONE round trip for both clouds' partials (they sit in one device buffer)
        float *res = (float *) pinned_scratch(ctx, 2 * 8 * sizeof(float) * kBboxBlocks);
        if (!res) return WM_ERR_HIP;
        const size_t slot = 8 * (size_t) kBboxBlocks;
        if (ctx->src_pending && ctx->tgt_pending) {
            WM_TRY(fast_fetch(ctx, res, ctx->cloud_bbox.p, 2 * slot * sizeof(float)));
        } else if (ctx->src_pending) {
            WM_TRY(fast_fetch(ctx, res, ctx->cloud_bbox.p, 8 * sizeof(float) * ctx->src_bbox_blocks));
        } else {
            WM_TRY(fast_fetch(ctx, res + slot, ctx->cloud_bbox.as<float>() + slot,
                              8 * sizeof(float) * ctx->tgt_bbox_blocks));
        }
    }
    // Both results are in: the target's grid ladder is enqueued FIRST (main stream), the source's Morton
    // sort behind it on the side stream.  The preparation is bound by how fast the host can enqueue its
    // ~55 small launches, not by the device: with the target's chain (the longer one on the device:
    // ~250 us at 1M points) enqueued first, the device works through it while the host is still
    // enqueueing the sort (sort first: the target's chain could not start before the sort's last launch
    // had been issued -- ~100 us later).
    const bool sort_src = ctx->src_pending;
    size_t src_valid = 0;
    if (sort_src) {
        ctx->src_pending = false;
        const float *res = (const float *) ctx->h_scratch;
        finish_bbox(res, ctx->src_bbox_blocks, &ctx->src_bbox, &src_valid);
        if (ctx->trace)
            fprintf(stderr, "[wm] source: valid=%zu lo=(%g %g %g) hi=(%g %g %g)\n", src_valid, ctx->src_bbox.lo[0],
                    ctx->src_bbox.lo[1], ctx->src_bbox.lo[2], ctx->src_bbox.hi[0], ctx->src_bbox.hi[1],
                    ctx->src_bbox.hi[2]);
    }
    const bool tgt_new = ctx->tgt_pending;
    if (tgt_new) {
        ctx->tgt_pending = false;
        const float *res = (const float *) ctx->h_scratch + 8 * (size_t) kBboxBlocks;
        size_t valid = 0;
        finish_bbox(res, ctx->tgt_bbox_blocks, &ctx->tgt_bbox, &valid);
        ctx->n_tgt = valid;
    }
    if (sort_src) ctx->n_src = src_valid;  // (the count of finite points: what the sort will leave in src_sorted)
    hipStream_t main_stream = ctx->stream;
    // the Morton sort of the source is independent of the target's grid build: side stream
    const bool aside = sort_aside != 0 && sort_src && ctx->tune_two_streams && ctx->side_stream != nullptr;
    const bool side = aside || (sort_src && ctx->tune_two_streams && ctx->side_stream && tgt_new && max_corr > 0);
    if (side) {  // (the sort may start as soon as what is on the main stream NOW -- the packed clouds -- is done)
        WM_HIP(ctx, hipEventRecord(ctx->ev_fork, main_stream));
        WM_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    }
    int rc = WM_OK;
    if (max_corr > 0 && ctx->n_src > 0 && ctx->n_tgt > 0 && !use_brute(ctx, nn_method)) rc = ensure_levels(ctx, max_corr);
    if (sort_src && aside && sort_aside == 2) {
        ctx->sort_deferred = true;  // (enqueue_deferred_sort: the caller's own chain goes to the main stream first)
        return rc;
    }
    if (sort_src) {
        if (side) ctx->stream = ctx->side_stream;
        const int rc2 = morton_sort(ctx, ctx->src_orig.as<float4>(), ctx->n_src_input, ctx->src_bbox, src_valid,
                                    ctx->src_sorted.as<float4>());
        ctx->stream = main_stream;
        if (rc2 != WM_OK) return rc2;
        WM_TRACE(ctx, "source: sorted");
        if (side) {
            WM_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
            if (aside)
                ctx->sort_join_pending = true;
            else
                WM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    }
    return rc;
}

}  // namespace wm

using namespace wm;

// =============================================================== C ABI
extern "C" {

const char *wm_version(void) { return "wavematch-hip 0.1 (gfx950, HIP)"; }

const char *wm_strerror(int s) {
    switch (s) {
        case WM_OK: return "ok";
        case WM_NOT_CONVERGED: return "registration did not converge";
        case WM_TOO_FEW_CORRESPONDENCES: return "not enough correspondences";
        case WM_ERR_ARG: return "invalid argument";
        case WM_ERR_HIP: return "HIP runtime error (see wm_last_error)";
        case WM_ERR_RCCL: return "RCCL error (see wm_last_error)";
        case WM_ERR_STATE: return "call sequence error (missing source/target cloud)";
        case WM_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char *wm_last_error(const wm_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int wm_ctx_create(wm_ctx **out, int device) {
    if (!out) return WM_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return WM_ERR_HIP;
    wm_ctx *ctx = new (std::nothrow) wm_ctx();
    if (!ctx) return WM_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&ctx->ev_a) != hipSuccess || hipEventCreate(&ctx->ev_b) != hipSuccess) {
        delete ctx;
        return WM_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (const char *e = getenv("WM_TUNE_NDT_DENSE")) ctx->tune_ndt_dense = atoi(e);
    if (const char *e = getenv("WM_TUNE_NDT_VOX_SPLIT")) ctx->tune_ndt_vox_split = atoi(e);
    if (const char *e = getenv("WM_TUNE_NDT_KEYS64")) ctx->tune_ndt_keys64 = atoi(e);
    if (const char *e = getenv("WM_TUNE_KNN_R0")) {
        const float v = (float) atof(e);
        if (v >= 0.25f && v <= 8.f) ctx->tune_knn_r0 = v;
    }
    if (const char *e = getenv("WM_TUNE_SPIN_US")) ctx->tune_spin_us = atoi(e);
    if (const char *e = getenv("WM_TUNE_XCD_REVERSE")) ctx->tune_xcd_reverse = atoi(e);
    if (const char *e = getenv("WM_TUNE_NN_WALK_FILTER")) ctx->tune_nn_walk_filter = atoi(e);
    if (const char *e = getenv("WM_TUNE_SCAN")) ctx->tune_scan = atoi(e);
    if (const char *e = getenv("WM_SHARD_FORCE")) ctx->tune_force_shard = atoi(e);
    if (const char *e = getenv("WM_TUNE_TWO_STREAMS")) ctx->tune_two_streams = atoi(e);
    if (const char *e = getenv("WM_TUNE_FUSE_STATS")) ctx->tune_fuse_stats = atoi(e);
    if (const char *e = getenv("WM_TUNE_NN_BALANCED")) ctx->tune_nn_balanced = atoi(e);
    if (const char *e = getenv("WM_TUNE_FAST_SOLVE")) ctx->tune_fast_solve = atoi(e);
    if (const char *e = getenv("WM_TUNE_CERT_FROM")) ctx->tune_cert_from = atoi(e);
    if (const char *e = getenv("WM_TUNE_CERT_NB")) ctx->tune_cert_nb = atoi(e);
    if (const char *e = getenv("WM_TUNE_CERT_RC")) ctx->tune_cert_rc = atoi(e);
    if (const char *e = getenv("WM_TUNE_CERT_DBG_SKIP")) ctx->tune_cert_dbg_skip = atoi(e);
    if (const char *e = getenv("WM_TUNE_NN_EARLY_LOADS")) ctx->tune_nn_early_loads = atoi(e);
    if (const char *e = getenv("WM_TUNE_NN_NT_STORES")) ctx->tune_nn_nt_stores = atoi(e);
    if (const char *e = getenv("WM_TUNE_CERT_DISP")) {
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_cert_disp = v;
    }
    if (const char *e = getenv("WM_TUNE_CERT_CHANGED")) {
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_cert_changed = v;
    }
    if (const char *e = getenv("WM_TUNE_CERT_UNSETTLED")) {
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_cert_unsettled = v;
    }
    if (const char *e = getenv("WM_TUNE_CERT_PAD_MUL")) {
        const float v = (float) atof(e);
        if (v >= 0) ctx->tune_cert_pad_mul = v;
    }
    if (const char *e = getenv("WM_TUNE_CERT_PAD_FRAC")) {
        const float v = (float) atof(e);
        if (v >= 0) ctx->tune_cert_pad_frac = v;
    }
    if (const char *e = getenv("WM_TUNE_XCD_CHUNK")) ctx->tune_xcd_chunk = atoi(e);
    if (const char *e = getenv("WM_TUNE_RADIX_MIN")) ctx->tune_radix_min = atoi(e);
    if (const char *e = getenv("WM_TUNE_SORT")) ctx->tune_sort = atoi(e);
    if (const char *e = getenv("WM_TUNE_PACK_BBOX")) ctx->tune_pack_bbox = atoi(e);
    if (const char *e = getenv("WM_TUNE_NDT_BLOCKS")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 4096) ctx->tune_ndt_blocks = v;
    }
    if (const char *e = getenv("WM_TUNE_GICP_BLOCKS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4096) ctx->tune_gicp_blocks = v;
    }
    if (const char *e = getenv("WM_TRACE")) ctx->trace = atoi(e) != 0;
    if (const char *e = getenv("WM_TUNE_LANE_LF")) {
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_lane_lf = v;
    }
    if (const char *e = getenv("WM_GICP_PROFILE")) ctx->gicp_profile = atoi(e) != 0;
    if (const char *e = getenv("WM_NDT_PROFILE")) ctx->ndt_profile = atoi(e) != 0;
    if (const char *e = getenv("WM_TUNE_COOP_LF")) {
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_coop_lf = v;
    }
    if (const char *e = getenv("WM_TUNE_GICP_SERVED")) ctx->tune_gicp_served = atoi(e) == 2 ? 2 : (atoi(e) != 0 ? 1 : 0);
    if (const char *e = getenv("WM_TUNE_NDT_SPEC_HESSIAN")) ctx->tune_ndt_spec_hessian = atoi(e);
    if (const char *e = getenv("WM_TUNE_NDT_FUSED_FETCH")) ctx->tune_ndt_fused_fetch = atoi(e);
    if (const char *e = getenv("WM_TUNE_LAG")) ctx->tune_lag = atoi(e);
    if (const char *e = getenv("WM_TUNE_LATE")) ctx->tune_late = atoi(e);
    if (const char *e = getenv("WM_TUNE_BINS")) ctx->tune_bins = atoi(e);
    if (const char *e = getenv("WM_TUNE_GRID_VARIANT")) ctx->tune_grid_variant = atoi(e);
    if (const char *e = getenv("WM_TUNE_EARLY_SOURCE")) ctx->tune_early_source = atoi(e);
    if (const char *e = getenv("WM_TUNE_COV_DBG")) ctx->tune_cov_dbg = atoi(e) & 768;
    if (const char *e = getenv("WM_TUNE_R0")) {
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_r0 = v;
    }
    if (const char *e = getenv("WM_TUNE_R_LIGHT")) {  // developer tuning knob
        const float v = (float) atof(e);
        if (v > 0) ctx->tune_r_light = v;
    }
    *out = ctx;
    return WM_OK;
}

int wm_ctx_set_stream(wm_ctx *ctx, void *hip_stream, int external) {
    if (!ctx) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // NB: a NULL handle with external != 0 is the (legacy) default stream -- which is
    // what torch.cuda.current_stream().cuda_stream returns unless a side stream is active
    ctx->stream = external ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return WM_OK;
}

void wm_ctx_destroy(wm_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->src_sorted, &ctx->tgt_orig, &ctx->staging, &ctx->staging2, &ctx->cell_of, &ctx->counts,
                      &ctx->block_sums, &ctx->bbox_buf, &ctx->cloud_bbox, &ctx->keys, &ctx->keys_bak, &ctx->match_pt, &ctx->match_pt_bak, &ctx->d_levels, &ctx->ndt_keys, &ctx->ndt_keys2,
                      &ctx->ndt_vox, &ctx->ndt_vkey, &ctx->ndt_hkeys, &ctx->ndt_hvals, &ctx->ndt_dense, &ctx->ndt_meanf, &ctx->src_orig,
                      &ctx->gicp_c1, &ctx->gicp_c2, &ctx->gicp_mahal, &ctx->gicp_mailbox, &ctx->src_grid.pts,
                      &ctx->src_grid.cell_start, &ctx->vg_idx, &ctx->vg_idx2, &ctx->vg_perm,
                      &ctx->vg_perm2, &ctx->vg_tmp, &ctx->vg_seg, &ctx->io_a, &ctx->io_b, &ctx->ds_ref,
                      &ctx->ds_tgt, &ctx->match_ref, &ctx->match_tgt,
                      &ctx->partials, &ctx->partials2, &ctx->bins, &ctx->nn_bound, &ctx->late_ctl, &ctx->cert_count, &ctx->cert_prof, &ctx->cost_log, &ctx->phase_log, &ctx->shard_ref, &ctx->shard_tgt,
                      &ctx->shard_ref_band, &ctx->shard_tgt_band, &ctx->shard_misc, &ctx->shard_flags, &ctx->shard_pos_t,
                      &ctx->shard_pos_s, &ctx->shard_stats, &ctx->ndt_sum_dev, &ctx->corr_tmp_idx, &ctx->corr_tmp_d2, &ctx->d_state};
    for (DevBuf *b : bufs) b->release();
    small_batch_release(ctx);
    gicp_small_release(ctx);
    ndt_small_release(ctx);
    batch_voxel_release(ctx);
    for (auto &l : ctx->levels) {
        l.pts.release();
        l.cell_start.release();
    }
    if (ctx->h_state) (void) hipHostFree(ctx->h_state);
    if (ctx->h_gicp) (void) hipHostFree(ctx->h_gicp);
    if (ctx->h_gicp_slots) (void) hipHostFree(ctx->h_gicp_slots);
    if (ctx->h_ndt) (void) hipHostFree(ctx->h_ndt);
    if (ctx->h_sig) (void) hipHostFree(ctx->h_sig);
    if (ctx->h_pub) (void) hipHostFree(ctx->h_pub);
    if (ctx->h_late) (void) hipHostFree(ctx->h_late);
    if (ctx->ev_block) (void) hipEventDestroy(ctx->ev_block);
    if (ctx->h_scratch) (void) hipHostFree(ctx->h_scratch);
    for (hipEvent_t e : ctx->ev_pool) (void) hipEventDestroy(e);
    if (ctx->ev_a) (void) hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void) hipEventDestroy(ctx->ev_b);
    if (ctx->ev_fork) (void) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void) hipEventDestroy(ctx->ev_join);
    if (ctx->side_stream) (void) hipStreamDestroy(ctx->side_stream);
    if (ctx->own_stream) (void) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int wm_set_grid_cell(wm_ctx *ctx, float grid_cell) {
    if (!ctx || !(grid_cell >= 0)) return WM_ERR_ARG;
    ctx->grid_cell_override = grid_cell;
    for (auto &l : ctx->levels) l.built = false;
    ctx->n_levels = 0;
    ctx->levels_max_corr = -1;
    return WM_OK;
}

int wm_cloud_sizes(const wm_ctx *ctx, size_t *n_source, size_t *n_target) {
    if (!ctx) return WM_ERR_ARG;
    if (ctx->src_pending || ctx->tgt_pending) {  // counts of finite points: the pending reductions' results
        wm_ctx *c = const_cast<wm_ctx *>(ctx);
        WM_HIP(c, hipSetDevice(c->device));
        WM_TRY(finalize_clouds(c));
    }
    if (n_source) *n_source = ctx->n_src;
    if (n_target) *n_target = ctx->n_tgt;
    return WM_OK;
}

int wm_set_source(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem) {
    if (!ctx || (n > 0 && !pts) || stride < 12 || (stride & 3) || n > 0x7FFFFFF0u) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    ctx->sort_deferred = false;  // (a sort of the cloud that is replaced here, never enqueued: dropped)
    WM_TRY(join_source_sort(ctx));  // (a sort left running aside by a call that ended early: it reads what is replaced here)
    ctx->have_corr = false;
    ctx->n_src_input = n;
    ctx->n_src = 0;
    ctx->src_pending = false;
    if (n == 0) return WM_OK;
    // pack (caller order, kept for GICP's k-NN covariances) and launch the bounding-box reduction;
    // the Morton order is produced by finalize_clouds once the box has been fetched
    ctx->gicp_cov_src_valid = false;
    WM_HIP(ctx, ctx->src_orig.reserve(n * sizeof(float4)));
    WM_HIP(ctx, ctx->src_sorted.reserve(n * sizeof(float4)));
    WM_HIP(ctx, ctx->cloud_bbox.reserve(2 * 8 * sizeof(float) * kBboxBlocks));
    if (ctx->trace) fprintf(stderr, "[wm] set_source: n=%zu stride=%zu mem=%d ptr=%p\n", n, stride, mem, pts);
    WM_TRY(pack_cloud(ctx, pts, n, stride, mem, ctx->src_orig.as<float4>(), 0, false, ctx->cloud_bbox.as<float>(), &ctx->src_bbox_blocks));
    WM_TRACE(ctx, "set_source: packed");
    ctx->src_pending = true;
    return WM_OK;
}

int wm_set_target(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem) {
    if (!ctx || (n > 0 && !pts) || stride < 12 || (stride & 3) || n > 0x7FFFFFF0u) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    ctx->have_corr = false;
    ctx->ndt_built = false;
    ctx->gicp_cov_tgt_valid = false;
    ctx->n_tgt_input = n;
    ctx->n_tgt = 0;
    ctx->tgt_pending = false;
    for (auto &l : ctx->levels) l.built = false;
    ctx->n_levels = 0;
    ctx->levels_max_corr = -1;
    if (n == 0) return WM_OK;
    WM_HIP(ctx, ctx->tgt_orig.reserve(n * sizeof(float4)));
    WM_HIP(ctx, ctx->cloud_bbox.reserve(2 * 8 * sizeof(float) * kBboxBlocks));
    int slot = 0;
    bool staged = false;
    if (mem == WM_MEM_HOST && ctx->src_pending && ctx->tune_early_source) {
        // (pinned caller memory: the upload STARTS here, on a copy engine, and runs under the source's round trip and the
        // ~150 us this thread needs to enqueue the source's sort -- a blocking copy behind those, round 3's order, left the
        // device idle for the 0.2 ms of the copy: the sort's launches are issued faster than it could start)
        if (ctx->tune_early_source >= 2) staged = upload_begin_async(ctx, pts, n * stride);
    }
    // (whatever ends this call early: the copy engine has finished with the caller's memory before it returns)
    struct DrainCopy {
        wm_ctx *c;
        bool on;
        ~DrainCopy() {
            if (on) (void) hipStreamSynchronize(c->side_stream);
        }
    } drain_copy{ctx, staged};
    if (mem == WM_MEM_HOST && ctx->src_pending && ctx->tune_early_source) {
        // A HOST target right behind a new source: this cloud is about to spend ~0.25 ms per 16 MB on PCIe with
        // the device idle.  Everything the source still needs -- its bounding box (one short round trip), its
        // Morton sort and gather -- is put on the stream first and runs under the copy (own staging buffer, no
        // drain of the stream: pack_cloud slot 1).
        float *res = (float *) pinned_scratch(ctx, 2 * 8 * sizeof(float) * kBboxBlocks);
        if (!res) return WM_ERR_HIP;
        WM_TRY(fast_fetch(ctx, res, ctx->cloud_bbox.p, 8 * sizeof(float) * ctx->src_bbox_blocks));
        size_t src_valid = 0;
        finish_bbox(res, ctx->src_bbox_blocks, &ctx->src_bbox, &src_valid);
        ctx->src_pending = false;
        ctx->n_src = src_valid;
        WM_TRY(morton_sort(ctx, ctx->src_orig.as<float4>(), ctx->n_src_input, ctx->src_bbox, src_valid,
                           ctx->src_sorted.as<float4>()));
        slot = 1;
    }
    WM_TRY(pack_cloud(ctx, pts, n, stride, mem, ctx->tgt_orig.as<float4>(), slot, staged, ctx->cloud_bbox.as<float>() + 8 * kBboxBlocks,
                      &ctx->tgt_bbox_blocks));
    drain_copy.on = false;  // (pack_cloud waited for it)
    ctx->tgt_pending = true;
    // the search grid is built by the first caller that searches (finalize_clouds / ensure_levels in
    // the ICP / GICP / search entry points): an NDT registration never needs it
    return WM_OK;
}

void wm_icp_default_params(wm_icp_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->max_corr = 3;    // icp.hpp:35
    p->max_iter = 100;  // icp.hpp:37
    p->t_eps = 1e-8;    // icp.hpp:41
    p->fit_eps = 1e-2;  // icp.hpp:43
    p->mode = WM_ICP_SVD;
    p->nn_method = WM_NN_AUTO;
    p->carry_state = 1;
}

int wm_icp_align(wm_ctx *ctx, const wm_icp_params *p, double T_out[16], wm_icp_stats *stats) {
    if (!ctx || !p || !T_out) return WM_ERR_ARG;
    if (!(p->max_corr > 0) || (p->mode != WM_ICP_SVD && p->mode != WM_ICP_GN6)) return WM_ERR_ARG;
    if (p->force_iterations <= 0 && p->max_iter <= 0) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) {
        // PCL: empty input -> "Not enough correspondences"; match() returns false
        if (stats) stats->state = WM_CONV_NO_CORRESPONDENCES;
        return ctx->n_tgt_input == 0 && ctx->n_src_input == 0 ? WM_ERR_STATE
                                                               : WM_TOO_FEW_CORRESPONDENCES;
    }
    WM_TRY(finalize_clouds(ctx, p->max_corr, p->nn_method));
    if (ctx->n_src == 0 || ctx->n_tgt == 0) {
        if (stats) stats->state = WM_CONV_NO_CORRESPONDENCES;
        return WM_TOO_FEW_CORRESPONDENCES;
    }
    WM_TRY(prepare_work(ctx));
    const bool brute = use_brute(ctx, p->nn_method);
    if (!brute) WM_TRY(ensure_levels(ctx, p->max_corr));
    const float thr = threshold_d2(p->max_corr);
    double I[16];
    mat4_identity(I);
    const double prev = (p->carry_state && ctx->prev_mse >= 0) ? ctx->prev_mse : DBL_MAX;
    init_state(ctx->h_state, I, p, prev);
    ctx->h_state->svd_warm = ctx->tune_fast_solve ? 1 : 0;
    set_step_scale(ctx);
    WM_TRY(upload_state(ctx));

    return icp_run_loop(ctx, p, brute, thr, nullptr, nullptr, T_out, stats);
}

}  // extern "C"

namespace wm {

// The iteration loop of one registration, from an uploaded state to the fetched result: shared by
// wm_icp_align (blk == nullptr) and the sharded registration (wm_shard.hip: blk = the WM_STATS_LEN
// doubles in HBM that are all-reduced over `comm` between a rank's sums and the solve).
int icp_run_loop(wm_ctx *ctx, const wm_icp_params *p, bool brute, float thr, wm_comm *comm, double *blk,
                 double T_out[16], wm_icp_stats *stats) {
    const int max_it = p->force_iterations > 0 ? p->force_iterations : p->max_iter;
    const int nb = stat_blocks(ctx->n_src);
    ctx->iter_nn_ms.clear();
    WM_HIP(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    // The host runs AHEAD of the device, never more than kLag iterations (2: one iteration in flight, one
    // queued behind it -- 4 decided two iterations later when to certify, 3.72 vs 3.69 ms; 1 drains the
    // queue between iterations, 4.04 ms): every solve kernel publishes
    // (done, iterations finished, the size of its step) in one word of pinned memory, and before
    // enqueueing iteration `it` the host waits until iteration it - kLag has been published.  The
    // device always has work queued (no pipeline drain, round 2: one every 8 iterations), iterations
    // enqueued behind a `done` are no-ops, and the host picks the search kernel of iteration `it` from
    // the step size iteration it - kLag recorded (its own record, so the choice does not depend on
    // timing and a registration stays bit-reproducible): the full search (k_nn_grid) while the clouds
    // still move, the certificate kernel (k_nn_cert) once a step is a small fraction of a grid cell.
    // The choice changes the work, never the correspondences.
    const int kLag = ctx->tune_lag >= 1 && ctx->tune_lag <= 16 ? ctx->tune_lag : 2;
    if (ctx->h_pub_slots < max_it + 1) {
        if (ctx->h_pub) (void) hipHostFree(ctx->h_pub);
        ctx->h_pub = nullptr;
        ctx->h_pub_slots = 0;
        WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_pub, sizeof(unsigned long long) * (size_t) (max_it + 2),
                                  hipHostMallocDefault));
        ctx->h_pub_slots = max_it + 1;
    }
    // (nothing of an earlier align is in flight: each ends with a fetch of the state)
    memset(ctx->h_pub, 0, sizeof(unsigned long long) * (size_t) (max_it + 1));
    const bool can_cert = !brute && ctx->tune_fuse_stats && ctx->tune_nn_balanced && ctx->tune_cert_from >= -1 &&
                          ctx->n_tgt_input < (1u << 26) - 8u && !ctx->cost_log.p;
    volatile unsigned long long *pub = ctx->h_pub;
    bool cert_on = false, bounds_valid = false, seen_done = false;
    const float cert_thr = brute ? 0.f : ctx->tune_cert_disp * ctx->levels[0].d.h;
    size_t ev_used = 0;
    size_t ev_ar = (size_t) 5 * (size_t) max_it;  // the all-reduce's event pairs sit behind the iterations' slots
    const bool slab = ctx->h_state->slab_on != 0;
    XchgDev xchg{nullptr, nullptr, 0, 0, 0u};
    const bool in_kernel_exchange = blk && comm_exchange_args(comm, &xchg) == WM_OK;
    ctx->cert_launches = 0;
    // the grid path adds its sums into bins (wm_bins.hpp) and solves from them -- k_bins_solve, or, sharded, the
    // k_reduce_solve that carries the exchange: no k_reduce_rows, no rows of partial sums
    const bool use_bins = !brute && ctx->tune_fuse_stats && ctx->tune_bins != 0 && !ctx->cost_log.p;
    if (use_bins) {
        WM_TRY(bins_ready(ctx));  // (zeroes them if the last loop left them dirty)
        ctx->bins_dirty = true;   // (until this loop has ended normally)
    }
    std::vector<unsigned char> was_cert;
    std::vector<unsigned char> kind((size_t) max_it, 0);  // which search kernel iteration k got (1: certificate, 2: its first launch)
    std::vector<int> ev_slot((size_t) max_it, -1);        // profile: the iteration's first event in the pool
    // the resident kernel (k_nn_cert<.., LATE>): once the certificate policy is on, the remaining iterations run
    // inside ONE launch, the solve included, until the registration is done or the same policy says leave
    bool late_ok = can_cert && ctx->tune_late && !blk && !slab && !ctx->cert_count.p && !ctx->cert_prof.p;
    int cert_hold = 0;  // iterations for which the policy stays off after the resident kernel left by it
    ctx->late_iters = ctx->late_launches = 0;
    ctx->late_ms = 0.f;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> late_events;
    for (int it = 0; it < max_it; ++it) {
        float seen_disp = -1.f;  // what iteration it - kLag recorded (its own record): step size, ...
        float seen_changed = 1.f, seen_unsettled = 0.f;  // ... fraction of changed matches, of searched queries
        auto parse = [&](unsigned long long w) {
            seen_disp = __builtin_bit_cast(float, (unsigned) ((w >> 32) & 0xFFFFu) << 16);
            seen_changed = (float) ((w >> 16) & 0xFFFFu) / 65535.f;
            seen_unsettled = (float) (w & 0xFFFFu) / 65535.f;
        };
        if (it >= kLag) {  // wait for it (3 stages as in wait_flag)
            const unsigned need = (unsigned) (it - kLag + 1);  // iterations finished by then
            const auto t0 = std::chrono::steady_clock::now();
            bool yielding = false;
            for (unsigned spins = 1;; ++spins) {
                const unsigned long long w = pub[need];
                if ((unsigned) (w >> 48) == (need & 0xFFFFu) && w != 0ull) {
                    parse(w);
                    break;
                }
                // done -- and the record waited for is not one the device wrote before it stopped: nothing more
                // will come.  (The last record and the done word are two relaxed stores of one kernel: seeing
                // `done` first must not end the loop one iteration early -- in the sharded loop every enqueued
                // iteration carries a collective, and all ranks have to issue the same number of them: exactly
                // iterations-finished + kLag.)
                const unsigned long long p0 = pub[0];
                if ((p0 & 1ull) != 0ull && need > (unsigned) (p0 >> 1)) {
                    seen_done = true;
                    break;
                }
                if (yielding)
                    std::this_thread::yield();
                else
                    cpu_relax();
                if ((spins & 63u) == 0 || yielding) {
                    const auto waited = std::chrono::steady_clock::now() - t0;
                    if (waited > std::chrono::milliseconds(20)) {
                        // a long wait (huge clouds, a shared device): let the runtime block until everything
                        // enqueued has run -- the record is there then, unless a kernel failed
                        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        const unsigned long long w2 = pub[need];
                        if ((unsigned) (w2 >> 48) == (need & 0xFFFFu) && w2 != 0ull) parse(w2);
                        else seen_done = true;
                        break;
                    }
                    yielding = waited > std::chrono::microseconds(ctx->tune_spin_us);
                }
            }
            if (seen_done) break;
        }
        if (can_cert && cert_hold > 0) {
            --cert_hold;
            cert_on = false;
        } else if (can_cert) {
            if (ctx->tune_cert_from >= 0) {
                cert_on = it >= ctx->tune_cert_from;
            } else if (seen_disp >= 0.f) {
                // certify once a step is small AND few matches still change (on a scan whose density varies
                // by orders of magnitude the dense part keeps changing partners long after the step has
                // become small against the grid cell); back to full searches when a certificate launch had
                // to search a large share after all
                // (the record of a certificate launch that had no bounds to go by -- the first after full
                // searches -- says nothing: it searched everything)
                const unsigned char rec = (size_t) (it - kLag) < kind.size() ? kind[(size_t) (it - kLag)] : 0;
                if (!cert_on) {
                    if (seen_disp < cert_thr && seen_changed < ctx->tune_cert_changed && rec == 0) cert_on = true;
                } else if (rec == 1 && seen_unsettled > ctx->tune_cert_unsettled) {
                    cert_on = false;
                } else if (seen_disp > 3.f * cert_thr) {
                    cert_on = false;
                }
            }
        }
        unsigned late_blocks = 0;
        if (cert_on && late_ok && late_possible(ctx, p->mode, &late_blocks)) {
            const int share = resident_admit(ctx->device, (int) late_blocks, ctx->late_capacity);
            if (share > 0) {
                hipEvent_t l0 = nullptr, l1 = nullptr;
                if (p->profile) {
                    l0 = get_event(ctx, ev_used++);
                    l1 = get_event(ctx, ev_used++);
                    WM_HIP(ctx, hipEventRecord(l0, ctx->stream));
                }
                const unsigned seq = ++ctx->late_seq;
                // (a forced choice -- tune_cert_from >= 0 -- stays inside whatever the searched share)
                const bool forced_choice = ctx->tune_cert_from >= 0;
                int rc = launch_nn_late(ctx, thr, p->mode, late_blocks, bounds_valid, seq,
                                        forced_choice ? 2.f : ctx->tune_cert_unsettled, forced_choice ? 3.0e38f : 3.f * cert_thr,
                                        max_it - it);
                if (rc == WM_OK && l1) rc = hipEventRecord(l1, ctx->stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
                if (rc == WM_OK) {  // the host has nothing to decide until it leaves: wait for its word
                    volatile unsigned long long *hx = ctx->h_late;
                    const auto t0 = std::chrono::steady_clock::now();
                    bool yielding = false;
                    for (unsigned spins = 1; (unsigned) (*hx >> 32) != seq; ++spins) {
                        if (yielding) std::this_thread::yield();
                        else cpu_relax();
                        if ((spins & 63u) == 0 || yielding) {
                            const auto waited = std::chrono::steady_clock::now() - t0;
                            if (waited > std::chrono::milliseconds(20)) {
                                if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = WM_ERR_HIP;
                                break;
                            }
                            yielding = waited > std::chrono::microseconds(ctx->tune_spin_us);
                        }
                    }
                }
                resident_release(ctx->device, share);
                if (rc != WM_OK) {
                    if (rc == WM_ERR_HIP) ctx->last_error = "resident ICP kernel: launch or wait failed";
                    return rc;
                }
                const unsigned long long w = *ctx->h_late;
                if ((unsigned) (w >> 32) != seq) {
                    ctx->last_error = "resident ICP kernel: finished without its exit word";
                    return WM_ERR_STATE;
                }
                const int reason = (int) ((w >> 24) & 0xFFu), inside = (int) (w & 0xFFFFFFu);
                if (getenv("WM_LATE_DEBUG")) {  // developer: the solver's stamps (100 MHz wall clock)
                    unsigned long long d[64 * 4];
                    (void) hipStreamSynchronize(ctx->stream);
                    if (hipMemcpy(d, (char *) ctx->late_ctl.p + late_ctl_bytes(), sizeof(d), hipMemcpyDeviceToHost) == hipSuccess) {
                        fprintf(stderr, "[wm] late kernel: %d iterations, reason %d\n", inside, reason);
                        for (int k = 0; k < inside && k < 64; ++k)
                            fprintf(stderr, "  it %2d: workers (hand-out -> all rows in) %6.2f us | rows added %5.2f | solve %5.2f | hand-out %5.2f\n", k,
                                    k ? ((long long) d[k * 4] - (long long) d[(k - 1) * 4 + 3]) * 0.01 : 0.0,
                                    (d[k * 4 + 1] - d[k * 4]) * 0.01, (d[k * 4 + 2] - d[k * 4 + 1]) * 0.01,
                                    (d[k * 4 + 3] - d[k * 4 + 2]) * 0.01);
                        // the workers' stamps of iteration WM_LATE_DEBUG, relative to the solver's hand-out before it
                        const int li = atoi(getenv("WM_LATE_DEBUG"));
                        std::vector<unsigned long long> wst((size_t) late_blocks * 8);
                        if (li >= 1 && li < inside && li < 64 && ctx->cert_prof.p &&
                            hipMemcpy(wst.data(), ctx->cert_prof.p, wst.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                            const unsigned long long t0 = d[(li - 1) * 4 + 3];
                            const char *names[6] = {"pose in", "phase 1 done", "wave 0 searched", "all searched + stored", "row stored", "ticket drawn"};
                            for (int k = 0; k < 6; ++k) {
                                double mx = 0, mn = 1e30;
                                std::vector<double> v;
                                for (unsigned b = 0; b < late_blocks; ++b) {
                                    const double x = ((long long) wst[(size_t) b * 8 + k] - (long long) t0) * 0.01;
                                    v.push_back(x);
                                    mx = x > mx ? x : mx;
                                    mn = x < mn ? x : mn;
                                }
                                std::sort(v.begin(), v.end());
                                fprintf(stderr, "  it %d workers, %-22s: min %6.2f  median %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f us after the hand-out\n", li,
                                        names[k], mn, v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], mx);
                            }
                            std::vector<std::pair<double, unsigned>> slow;
                            for (unsigned b = 0; b < late_blocks; ++b)
                                slow.emplace_back(((long long) wst[(size_t) b * 8 + 3] - (long long) wst[(size_t) b * 8 + 1]) * 0.01, (unsigned) wst[(size_t) b * 8 + 6]);
                            std::sort(slow.begin(), slow.end());
                            fprintf(stderr, "  searches (phase 1 done -> all stored), slowest five [us, searched]:");
                            for (size_t k = slow.size() >= 5 ? slow.size() - 5 : 0; k < slow.size(); ++k) fprintf(stderr, " %.2f/%u", slow[k].first, slow[k].second);
                            double su = 0;
                            for (auto &x : slow) su += x.second;
                            fprintf(stderr, "; median %.2f/%u; searched per workgroup: mean %.1f\n", slow[slow.size() / 2].first, slow[slow.size() / 2].second, su / slow.size());
                        }
                    }
                }
                if (l0) late_events.emplace_back(l0, l1);
                ctx->late_launches++;
                ctx->late_iters += inside;
                ctx->cert_launches += inside;
                for (int k = 0; k < inside && (size_t) (it + k) < kind.size(); ++k) kind[(size_t) (it + k)] = (k == 0 && !bounds_valid) ? 2 : 1;
                if (inside > 0) bounds_valid = true;
                if (reason == 3) late_ok = false;        // (a wait gave up: launched iterations from here on)
                if (reason == 2) {                       // the policy: full searches again, and let the records catch up
                    cert_on = false;
                    cert_hold = kLag;
                }
                if (reason == 1) break;  // done (also: it was queued behind a `done` and ran nothing)
                if (inside <= 0 && reason != 3) {  // (cannot happen: it left without a reason to)
                    ctx->last_error = "resident ICP kernel: left without running an iteration";
                    return WM_ERR_STATE;
                }
                it += inside - 1;  // (the loop's own ++it: on to the first iteration it did not run)
                continue;
            }
        }
        hipEvent_t e0 = nullptr, e1 = nullptr, e1b = nullptr, e2 = nullptr, e3 = nullptr;
        if (p->profile) {  // 5 pool slots per iteration; level 1 only fills the first two
            ev_slot[(size_t) it] = (int) ev_used;
            e0 = get_event(ctx, ev_used++);
            e1 = get_event(ctx, ev_used++);
            if (p->profile >= 2) {
                e1b = get_event(ctx, ev_used++);
                e2 = get_event(ctx, ev_used++);
                e3 = get_event(ctx, ev_used++);
            } else {
                ev_used += 3;
                (void) get_event(ctx, ev_used - 1);
            }
        }
        unsigned rows = (unsigned) nb;
        if (brute) {
            WM_TRY(launch_nn_brute(ctx, thr, e0, e1));
            if (e1b) WM_HIP(ctx, hipEventRecord(e1b, ctx->stream));
            WM_TRY(launch_stats(ctx, p->mode));
        } else if (cert_on) {
            if (slab && !bounds_valid) {
                // a rank only ever writes the bounds of the queries it owns at the time: nothing stale
                // may survive a stretch of full searches (or the start)
                WM_HIP(ctx, ctx->nn_bound.reserve(((size_t) ctx->n_src + 64) * sizeof(float4)));
                WM_HIP(ctx, hipMemsetAsync(ctx->nn_bound.p, 0, ((size_t) ctx->n_src + 64) * sizeof(float4), ctx->stream));
            }
            WM_TRY(launch_nn_cert(ctx, thr, e0, e1, e1b, p->mode, &rows, bounds_valid || slab, use_bins));
            kind[(size_t) it] = bounds_valid ? 1 : 2;
            bounds_valid = true;
            ctx->cert_launches++;
            if (p->profile) {
                was_cert.resize((size_t) it + 1, 0);
                was_cert[(size_t) it] = 1;
            }
        } else {
            WM_TRY(launch_search_and_stats(ctx, thr, p->mode, e0, e1, e1b, &rows, use_bins));
            bounds_valid = false;
        }
        if (e2) WM_HIP(ctx, hipEventRecord(e2, ctx->stream));
        if (blk && in_kernel_exchange) {
            // sharded, mailboxes: this rank's sums, their exchange with the other ranks over xGMI and the same
            // solve on every rank in ONE launch
            WM_TRY(launch_reduce_solve<7>(ctx, rows, blk, ctx->h_pub, ctx->h_pub_slots, 1, &xchg, use_bins ? ctx->bins.as<long long>() : nullptr));
        } else if (blk) {
            // sharded: this rank's sums -> all-reduce of the block over the ranks (RCCL on this stream) ->
            // the same solve on every rank
            WM_TRY(launch_reduce_solve<1>(ctx, rows, blk, nullptr, 0, 1, nullptr, use_bins ? ctx->bins.as<long long>() : nullptr));
            hipEvent_t ea = nullptr, eb = nullptr;
            if (p->profile) {
                ea = get_event(ctx, ev_ar++);
                eb = get_event(ctx, ev_ar++);
                WM_HIP(ctx, hipEventRecord(ea, ctx->stream));
            }
            WM_TRY(comm_allreduce(ctx, comm, blk, kBlkLen));
            if (eb) WM_HIP(ctx, hipEventRecord(eb, ctx->stream));
            WM_TRY(launch_reduce_solve<2>(ctx, 0, blk, ctx->h_pub, ctx->h_pub_slots, 1));
        } else if (use_bins) {
            WM_TRY(launch_bins_solve(ctx, ctx->h_pub, ctx->h_pub_slots));
        } else {
            WM_TRY(launch_reduce_solve<3>(ctx, rows, nullptr, ctx->h_pub, ctx->h_pub_slots));
        }
        if (e3) WM_HIP(ctx, hipEventRecord(e3, ctx->stream));
        WM_HIP(ctx, hipGetLastError());
    }
    if (in_kernel_exchange) {  // every rank, whatever it saw: the ranks agree on how the exchange went (k_xchg_commit)
        hipLaunchKernelGGL(k_xchg_commit, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->d_state.as<IcpDevState>(), xchg);
        WM_HIP(ctx, hipGetLastError());
    }
    if (ctx->cert_launches > 0) WM_TRY(launch_fix_keys(ctx, thr));
    WM_TRY(download_state(ctx));
    WM_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    WM_HIP(ctx, hipEventSynchronize(ctx->ev_b));
    const IcpDevState &s = *ctx->h_state;
    if (s.xchg_failed) {
        ctx->xchg_timed_out = true;
        ctx->last_error = "sharded registration: a rank's block did not arrive in a mailbox in time, on this rank or -- as "
                          "the commit round told -- on a peer (a rank failed or fell behind by more than the exchange's "
                          "time limit); every rank of the group fails this registration alike";
        return WM_ERR_RCCL;
    }
    if (use_bins) ctx->bins_dirty = false;
    ctx->prev_mse = s.prev_mse;
    ctx->have_corr = true;
    ctx->last_align_valid = true;
    ctx->last_align_converged = s.converged != 0;
    ctx->last_align_sharded = blk != nullptr;
    memcpy(ctx->corr_T, s.T, sizeof(s.T));
    if (stats) {
        stats->converged = s.converged;
        stats->iterations = s.iter;
        stats->state = s.state;
        stats->n_corr = s.n_corr;
        stats->mse = s.mse;
        stats->prev_mse = s.prev_mse;
        stats->nn_levels = brute ? 0 : ctx->n_levels;
        stats->grid_cell = brute ? 0.f : ctx->levels[0].d.h;
        stats->deferred = s.deferred_total;
        stats->cert_launches = ctx->cert_launches;
        stats->late_iterations = ctx->late_iters;
        stats->late_launches = ctx->late_launches;
        for (auto &ev : late_events) {
            float a = 0;
            if (ev.first && ev.second && hipEventElapsedTime(&a, ev.first, ev.second) == hipSuccess) ctx->late_ms += a;
        }
        stats->late_ms = ctx->late_ms;
        stats->exchange_in_kernel = in_kernel_exchange ? 1 : 0;
        stats->owned_violations = s.owned_violations;
        (void) hipEventElapsedTime(&stats->align_ms, ctx->ev_a, ctx->ev_b);
        if (p->profile) {
            // iterations that ran (the rest of the last batch were no-ops)
            const int ran = s.iter + (s.state == WM_CONV_NO_CORRESPONDENCES ? 1 : 0);
            for (int it = 0; it < ran && (size_t) it < ev_slot.size(); ++it) {
                if (ev_slot[(size_t) it] < 0 || (size_t) (ev_slot[(size_t) it] + 4) >= ev_used) {
                    ctx->iter_nn_ms.push_back(-1.f);  // (ran inside the resident kernel: no launch of its own)
                    continue;
                }
                float a = 0, a2 = 0, b = 0, c = 0;
                hipEvent_t *e = &ctx->ev_pool[(size_t) ev_slot[(size_t) it]];
                (void) hipEventElapsedTime(&a, e[0], e[1]);
                if (p->profile >= 2) {
                    (void) hipEventElapsedTime(&a2, e[1], e[2]);
                    (void) hipEventElapsedTime(&b, e[2], e[3]);
                    (void) hipEventElapsedTime(&c, e[3], e[4]);
                }
                ctx->iter_nn_ms.push_back(a);
                stats->nn_ms += a;
                if ((size_t) it < was_cert.size() && was_cert[(size_t) it]) stats->nn_cert_ms += a;
                stats->coarse_ms += a2;
                stats->stats_ms += b;
                stats->solve_ms += c;
                stats->nn_launches += 1;
            }
        }
    }
    if (stats && blk && p->profile) {
        const int ran = s.iter + (s.state == WM_CONV_NO_CORRESPONDENCES ? 1 : 0);
        for (int it = 0; it < ran; ++it) {
            const size_t k = (size_t) 5 * (size_t) max_it + 2 * (size_t) it;
            if (k + 1 >= ctx->ev_pool.size() || k + 1 >= ev_ar) break;
            float a = 0;
            if (hipEventElapsedTime(&a, ctx->ev_pool[k], ctx->ev_pool[k + 1]) == hipSuccess) stats->allreduce_ms += a;
        }
    }
    if (s.state == WM_CONV_NO_CORRESPONDENCES) return WM_TOO_FEW_CORRESPONDENCES;
    if (!s.converged) return WM_NOT_CONVERGED;
    memcpy(T_out, s.T, sizeof(s.T));
    return WM_OK;
}

}  // namespace wm

extern "C" {

static int unpack_correspondences(wm_ctx *ctx, int32_t *match_idx, float *d2, size_t cap) {
    const size_t n_in = ctx->n_src_input;
    if (cap < n_in) return WM_ERR_ARG;
    if (n_in == 0) return WM_OK;
    WM_HIP(ctx, ctx->corr_tmp_idx.reserve(n_in * sizeof(int)));
    WM_HIP(ctx, ctx->corr_tmp_d2.reserve(n_in * sizeof(float)));
    // dropped (non-finite) source points keep "no match"
    WM_HIP(ctx, hipMemsetAsync(ctx->corr_tmp_idx.p, 0xFF, n_in * sizeof(int), ctx->stream));
    WM_HIP(ctx, hipMemsetAsync(ctx->corr_tmp_d2.p, 0, n_in * sizeof(float), ctx->stream));
    const unsigned n = (unsigned) ctx->n_src;
    if (n > 0) {
        hipLaunchKernelGGL(k_unpack_corr, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                           ctx->stream, ctx->src_sorted.as<float4>(), n,
                           ctx->keys.as<unsigned long long>(), ctx->corr_tmp_idx.as<int>(),
                           ctx->corr_tmp_d2.as<float>());
        WM_HIP(ctx, hipGetLastError());
    }
    if (match_idx) WM_TRY(copy_to_caller(ctx, match_idx, ctx->corr_tmp_idx.p, n_in * sizeof(int)));
    if (d2) WM_TRY(copy_to_caller(ctx, d2, ctx->corr_tmp_d2.p, n_in * sizeof(float)));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WM_OK;
}

int wm_icp_match(wm_ctx *ctx, const void *ref, size_t n_ref, const void *target, size_t n_target,
                 size_t stride, int mem, const wm_icp_params *p, float res, int multiscale_steps,
                 double T_out[16], wm_icp_stats *stats) {
    if (!ctx || !p || !T_out || (n_ref > 0 && !ref) || (n_target > 0 && !target) || stride < 12 ||
        (stride & 3) || n_ref > 0x7FFFFFF0u || n_target > 0x7FFFFFF0u)
        return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!(res > 0)) {  // icp.cpp:123-131
        WM_TRY(wm_set_source(ctx, ref, n_ref, stride, mem));
        WM_TRY(wm_set_target(ctx, target, n_target, stride, mem));
        return wm_icp_align(ctx, p, T_out, stats);
    }
    const size_t cap_r = n_ref > 0 ? n_ref : 1, cap_t = n_target > 0 ? n_target : 1;
    WM_HIP(ctx, ctx->match_ref.reserve(cap_r * sizeof(float4)));
    WM_HIP(ctx, ctx->match_tgt.reserve(cap_t * sizeof(float4)));
    WM_HIP(ctx, ctx->ds_ref.reserve(cap_r * sizeof(float4)));
    WM_HIP(ctx, ctx->ds_tgt.reserve(cap_t * sizeof(float4)));
    float4 *d_ref = ctx->match_ref.as<float4>(), *d_tgt = ctx->match_tgt.as<float4>();
    float4 *ds_ref = ctx->ds_ref.as<float4>(), *ds_tgt = ctx->ds_tgt.as<float4>();
    WM_TRACE(ctx, "match: begin");
    WM_TRY(pack_cloud(ctx, ref, n_ref, stride, mem, d_ref));
    WM_TRACE(ctx, "match: packed ref");
    WM_TRY(pack_cloud(ctx, target, n_target, stride, mem, d_tgt));
    WM_TRACE(ctx, "match: packed target");
    wm_icp_params prm = *p;
    wm_icp_stats last, total;
    memset(&total, 0, sizeof(total));
    double running[16];
    mat4_identity(running);
    const int steps = multiscale_steps > 0 ? multiscale_steps : 0;
    // both clouds are filtered once per scale: their bounding boxes are found once
    VgKnown kr{}, kt{};
    if (steps > 0) {
        if (n_ref > 0) WM_TRY(compute_bbox(ctx, d_ref, n_ref, &kr.bb, &kr.valid));
        if (n_target > 0) WM_TRY(compute_bbox(ctx, d_tgt, n_target, &kt.bb, &kt.valid));
    }
    for (int i = steps; i >= 0; --i) {
        const float leaf = (float) (pow(2, i) * res);  // icp.cpp:80
        size_t nr = 0, nt = 0;
        WM_TRY(voxel_downsample_dev(ctx, d_ref, n_ref, leaf, ds_ref, &nr, steps > 0 && n_ref > 0 ? &kr : nullptr));
        WM_TRACE(ctx, "match: voxel ref");
        WM_TRY(voxel_downsample_dev(ctx, d_tgt, n_target, leaf, ds_tgt, &nt, steps > 0 && n_target > 0 ? &kt : nullptr));
        WM_TRACE(ctx, "match: voxel target");
        if (ctx->trace) fprintf(stderr, "[wm] match: leaf=%g nr=%zu nt=%zu\n", leaf, nr, nt);
        if (steps > 0) {
            WM_TRY(transform_cloud_dev(ctx, ds_ref, nr, running, ds_ref));  // icp.cpp:84-86
            WM_TRACE(ctx, "match: transformed");
            prm.max_corr = pow(2, i) * p->max_corr;                          // icp.cpp:93-94
        }
        WM_TRY(wm_set_source(ctx, ds_ref, nr, sizeof(float4), WM_MEM_DEVICE));
        WM_TRACE(ctx, "match: set_source");
        WM_TRY(wm_set_target(ctx, ds_tgt, nt, sizeof(float4), WM_MEM_DEVICE));
        WM_TRACE(ctx, "match: set_target");
        double Ti[16];
        const int rc = wm_icp_align(ctx, &prm, Ti, &last);
        WM_TRACE(ctx, "match: align");
        total.align_ms += last.align_ms;
        total.nn_ms += last.nn_ms;
        total.nn_launches += last.nn_launches;
        if (stats) {
            const float a = total.align_ms, b = total.nn_ms;
            const int c = total.nn_launches;
            *stats = last;
            stats->align_ms = a;
            stats->nn_ms = b;
            stats->nn_launches = c;
        }
        if (rc != WM_OK) return rc;  // icp.cpp:96-98: fail fast, result untouched
        mat4_mul(Ti, running, running);  // icp.cpp:99-101
    }
    memcpy(T_out, running, sizeof(running));
    return WM_OK;
}

// ------------------------------------------------ sharded (multi-GPU) stepping
}  // extern "C"

namespace wm {

static void set_step_scale(wm_ctx *ctx) {  // centre and half diagonal of the (local) source cloud: IcpDevState::step_disp
    const Bbox &b = ctx->src_bbox;
    double d2 = 0;
    for (int k = 0; k < 3; ++k) {
        ctx->h_state->src_centre[k] = 0.5f * (b.lo[k] + b.hi[k]);
        d2 += 0.25 * ((double) b.hi[k] - b.lo[k]) * ((double) b.hi[k] - b.lo[k]);
    }
    ctx->h_state->src_radius = (float) sqrt(d2);
    if (ctx->n_src == 0 || !(ctx->h_state->src_radius == ctx->h_state->src_radius)) {
        ctx->h_state->src_radius = 0.f;
        ctx->h_state->src_centre[0] = ctx->h_state->src_centre[1] = ctx->h_state->src_centre[2] = 0.f;
    }
}

// expect < 0: the cloud's count of finite source points arrives in the all-reduced block (the sum of
// the ranks' stripe_finite); see IcpDevState::expect_owned
int shard_begin(wm_ctx *ctx, const wm_icp_params *p, double x_lo, double x_hi, double expect, double stripe_finite,
                bool *brute_out, float *thr_out, double prev_mse0) {
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx, p->max_corr, p->nn_method));
    WM_TRY(prepare_work(ctx));
    ctx->shard_brute = use_brute(ctx, p->nn_method) || ctx->n_tgt == 0;
    if (!ctx->shard_brute) WM_TRY(ensure_levels(ctx, p->max_corr));
    ctx->shard_thr = threshold_d2(p->max_corr);
    ctx->shard_params = *p;
    double I[16];
    mat4_identity(I);
    init_state(ctx->h_state, I, p, prev_mse0);
    ctx->h_state->svd_warm = ctx->tune_fast_solve ? 1 : 0;
    ctx->h_state->slab_on = 1;
    ctx->h_state->slab_lo = x_lo < -3.0e38 ? -INFINITY : (float) x_lo;
    ctx->h_state->slab_hi = x_hi > 3.0e38 ? INFINITY : (float) x_hi;
    ctx->h_state->expect_owned = expect;
    ctx->h_state->stripe_finite = stripe_finite;
    set_step_scale(ctx);
    // keys / matches double as next iteration's candidates: start from "nothing known" (a rank only ever
    // writes the entries of the queries it owns at the time)
    const size_t n1 = ctx->n_src > 0 ? ctx->n_src : 1;
    WM_HIP(ctx, hipMemsetAsync(ctx->keys.p, 0xFF, n1 * sizeof(unsigned long long), ctx->stream));
    WM_HIP(ctx, hipMemsetAsync(ctx->match_pt.p, 0xFF, n1 * sizeof(float4), ctx->stream));
    WM_TRY(upload_state(ctx));
    ctx->iter_nn_ms.clear();
    ctx->shard_active = true;
    if (brute_out) *brute_out = ctx->shard_brute;
    if (thr_out) *thr_out = ctx->shard_thr;
    return WM_OK;
}

}  // namespace wm

extern "C" {

int wm_icp_shard_begin(wm_ctx *ctx, const wm_icp_params *p, double x_lo, double x_hi,
                       size_t expect_owned_total) {
    // (an empty slab, x_lo == x_hi, and an empty band of the source are legitimate for a rank of a
    // sharded registration: it contributes zeros)
    if (!ctx || !p || !(p->max_corr > 0) || !(x_lo <= x_hi)) return WM_ERR_ARG;
    if (ctx->n_src_input == 0 && expect_owned_total == 0) return WM_ERR_STATE;
    WM_TRY(shard_begin(ctx, p, x_lo, x_hi, (double) expect_owned_total, 0.0, nullptr, nullptr, DBL_MAX));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WM_OK;
}

int wm_icp_shard_local_stats(wm_ctx *ctx, void *stats_dev) {
    if (!ctx || !stats_dev) return WM_ERR_ARG;
    if (!ctx->shard_active) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->shard_params.profile) {
        const size_t k = ctx->iter_nn_ms.size();
        e0 = get_event(ctx, 2 * k);
        e1 = get_event(ctx, 2 * k + 1);
        ctx->iter_nn_ms.push_back(-1.f);
    }
    unsigned rows = 0;
    if (ctx->n_src > 0) {
        if (ctx->shard_brute) {
            WM_TRY(launch_nn_brute(ctx, ctx->shard_thr, e0, e1));
            WM_TRY(launch_stats(ctx, ctx->shard_params.mode));
            rows = (unsigned) stat_blocks(ctx->n_src);
        } else {
            WM_TRY(launch_search_and_stats(ctx, ctx->shard_thr, ctx->shard_params.mode, e0, e1, nullptr, &rows));
        }
    }
    return launch_reduce_solve<1>(ctx, rows, static_cast<double *>(stats_dev));
}

int wm_icp_shard_apply(wm_ctx *ctx, const void *stats_dev) {
    if (!ctx || !stats_dev) return WM_ERR_ARG;
    if (!ctx->shard_active) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    return launch_reduce_solve<2>(ctx, 0, const_cast<double *>(static_cast<const double *>(stats_dev)));
}

int wm_icp_shard_poll(wm_ctx *ctx, int *done, double T_out[16], wm_icp_stats *stats) {
    if (!ctx) return WM_ERR_ARG;
    if (!ctx->shard_active) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(download_state(ctx));
    const IcpDevState &s = *ctx->h_state;
    if (done) *done = s.done;
    if (T_out) memcpy(T_out, s.T, sizeof(s.T));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->converged = s.converged;
        stats->iterations = s.iter;
        stats->state = s.state;
        stats->n_corr = s.n_corr;
        stats->mse = s.mse;
        stats->prev_mse = s.prev_mse;
        stats->deferred = s.deferred_total;
        stats->owned_violations = s.owned_violations;
        stats->nn_levels = ctx->shard_brute ? 0 : ctx->n_levels;
        stats->grid_cell = ctx->shard_brute ? 0.f : ctx->levels[0].d.h;
        if (ctx->shard_params.profile) {
            for (size_t k = 0; k < ctx->iter_nn_ms.size(); ++k) {
                float ms = 0;
                if (ctx->n_src > 0 &&
                    hipEventElapsedTime(&ms, ctx->ev_pool[2 * k], ctx->ev_pool[2 * k + 1]) == hipSuccess) {
                    ctx->iter_nn_ms[k] = ms;
                    stats->nn_ms += ms;
                    stats->nn_launches += 1;
                }
            }
        }
    }
    if (s.done) {
        ctx->have_corr = true;
        ctx->last_align_valid = true;
        ctx->last_align_converged = s.converged != 0;
        memcpy(ctx->corr_T, s.T, sizeof(s.T));
    }
    if (s.state == WM_CONV_NO_CORRESPONDENCES) return WM_TOO_FEW_CORRESPONDENCES;
    if (s.done && !s.converged) return WM_NOT_CONVERGED;
    return WM_OK;
}

// ------------------------------------------------ host-only ICP state machine
struct wm_host_icp {
    IcpDevState st;
};

int wm_host_icp_create(wm_host_icp **out, const wm_icp_params *p, size_t expect_owned_total) {
    if (!out || !p) return WM_ERR_ARG;
    wm_host_icp *h = new (std::nothrow) wm_host_icp();
    if (!h) return WM_ERR_NOMEM;
    double I[16];
    mat4_identity(I);
    init_state(&h->st, I, p, DBL_MAX);
    h->st.svd_warm = 1;
    h->st.expect_owned = (double) expect_owned_total;
    *out = h;
    return WM_OK;
}

void wm_host_icp_destroy(wm_host_icp *h) { delete h; }

int wm_host_icp_apply(wm_host_icp *h, const double stats[WM_STATS_LEN]) {
    if (!h || !stats) return WM_ERR_ARG;
    if (h->st.done) return WM_OK;
    icp_apply_stats(&h->st, stats);
    return WM_OK;
}

int wm_host_icp_get(const wm_host_icp *h, int *done, double T_out[16], wm_icp_stats *stats) {
    if (!h) return WM_ERR_ARG;
    if (done) *done = h->st.done;
    if (T_out) memcpy(T_out, h->st.T, sizeof(h->st.T));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->converged = h->st.converged;
        stats->iterations = h->st.iter;
        stats->state = h->st.state;
        stats->n_corr = h->st.n_corr;
        stats->mse = h->st.mse;
        stats->prev_mse = h->st.prev_mse;
        stats->owned_violations = h->st.owned_violations;
    }
    return WM_OK;
}

int wm_debug_cost_log(wm_ctx *ctx, int iterations, unsigned *out, size_t cap) {
    if (!ctx || iterations < 0) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (!out) {  // arm: the next align records the search cost of its first `iterations` iterations
        WM_TRY(finalize_clouds(ctx));
        if (iterations == 0) {
            WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->cost_log.release();
            ctx->phase_log.release();
            ctx->cost_log_cap = 0;
            return WM_OK;
        }
        WM_HIP(ctx, ctx->cost_log.reserve((size_t) iterations * (ctx->n_src > 0 ? ctx->n_src : 1) * 4));
        WM_HIP(ctx, ctx->phase_log.reserve((size_t) iterations * 8 * sizeof(unsigned long long)));
        WM_HIP(ctx, hipMemsetAsync(ctx->phase_log.p, 0, (size_t) iterations * 8 * sizeof(unsigned long long), ctx->stream));
        ctx->cost_log_iter = 0;
        ctx->cost_log_cap = iterations;
        return WM_OK;
    }
    const size_t need = (size_t) ctx->cost_log_iter * ctx->n_src;
    if (cap < need) return WM_ERR_ARG;
    WM_TRY(copy_to_caller(ctx, out, ctx->cost_log.p, need * 4));
    return ctx->cost_log_iter;
}

int wm_debug_phase_log(wm_ctx *ctx, unsigned long long *out, int iterations) {
    if (!ctx || !out || iterations < 0 || !ctx->phase_log.p) return WM_ERR_ARG;
    if (iterations > ctx->cost_log_iter) iterations = ctx->cost_log_iter;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(copy_to_caller(ctx, out, ctx->phase_log.p, (size_t) iterations * 8 * sizeof(unsigned long long)));
    return iterations;
}

int wm_debug_solve_cycles(wm_ctx *ctx, unsigned long long out[8]) {
    if (!ctx || !out || !ctx->h_state) return WM_ERR_ARG;
    for (int k = 0; k < 8; ++k) out[k] = ctx->h_state->dbg[k];
    return WM_OK;
}

int wm_debug_bins_sum(const double *x, size_t n, const unsigned *perm, double *out, long long limbs_out[3]) {
    if (!out || (n > 0 && !x)) return WM_ERR_ARG;
    std::vector<long long> bins(kBinWords, 0ll);
    for (size_t k = 0; k < n; ++k) {
        const size_t i = perm ? perm[k] : k;
        if (i >= n) return WM_ERR_ARG;
        const double v = x[i];
        if (!(fabs(v) < 4611686018427387904.0)) return WM_ERR_ARG;
        long long l[kBinLimbs];
        bins_split(v, l);
        const size_t bin = k % (size_t) kBinCount;  // (any assignment of addends to bins gives the same totals)
        for (int j = 0; j < kBinLimbs; ++j) bins[(bin * kBinLimbs + (size_t) j) * kBinStride] += l[j];
    }
    long long L[kBinLimbs] = {0, 0, 0};
    for (int b = 0; b < kBinCount; ++b)
        for (int j = 0; j < kBinLimbs; ++j) L[j] += bins[((size_t) b * kBinLimbs + (size_t) j) * kBinStride];
    *out = bins_value(L[0], L[1], L[2]);
    if (limbs_out)
        for (int j = 0; j < kBinLimbs; ++j) limbs_out[j] = L[j];
    return WM_OK;
}

int wm_debug_copy_bandwidth(wm_ctx *ctx, size_t bytes, int reps, double *gb_per_s) {
    if (!ctx || !gb_per_s || bytes < (1u << 20) || reps < 1) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = bytes / sizeof(float4);
    float4 *a = nullptr, *b = nullptr;
    WM_HIP(ctx, hipMalloc((void **) &a, n * sizeof(float4)));
    if (hipMalloc((void **) &b, n * sizeof(float4)) != hipSuccess) {
        (void) hipFree(a);
        ctx->last_error = "wm_debug_copy_bandwidth: hipMalloc failed";
        return WM_ERR_NOMEM;
    }
    (void) hipMemsetAsync(a, 0x3c, n * sizeof(float4), ctx->stream);
    const copy_f4v *ca = reinterpret_cast<const copy_f4v *>(a);
    copy_f4v *cb = reinterpret_cast<copy_f4v *>(b);
    float ms = 0;
    hipError_t e = hipSuccess;
    for (int shape = 0; shape < 3; ++shape) {
        const unsigned blocks = shape == 0 ? 1024u : (shape == 1 ? 65536u : 16384u);
        for (int r = -2; r < reps; ++r) {  // two warm-up launches
            if (r == 0) (void) hipEventRecord(ctx->ev_a, ctx->stream);
            if (shape == 2) hipLaunchKernelGGL(k_copy_f4<true>, dim3(blocks), dim3(256), 0, ctx->stream, ca, cb, n);
            else hipLaunchKernelGGL(k_copy_f4<false>, dim3(blocks), dim3(256), 0, ctx->stream, ca, cb, n);
        }
        (void) hipEventRecord(ctx->ev_b, ctx->stream);
        e = hipEventSynchronize(ctx->ev_b);
        float t = 0;
        (void) hipEventElapsedTime(&t, ctx->ev_a, ctx->ev_b);
        if (e != hipSuccess) break;
        if (shape == 0 || (t > 0 && t < ms)) ms = t;
    }
    (void) hipFree(a);
    (void) hipFree(b);
    WM_HIP(ctx, e);
    *gb_per_s = ms > 0 ? 2.0 * (double) (n * sizeof(float4)) * reps / (ms * 1e-3) / 1e9 : 0.0;
    return WM_OK;
}

int wm_set_option(wm_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return WM_ERR_ARG;
    const std::string k(name);
    if (k == "cert_from") ctx->tune_cert_from = (int) value;
    else if (k == "cert_nb") ctx->tune_cert_nb = (int) value;
    else if (k == "cert_rc") ctx->tune_cert_rc = (int) value;
    else if (k == "cert_disp" && value > 0) ctx->tune_cert_disp = (float) value;
    else if (k == "cert_changed" && value > 0) ctx->tune_cert_changed = (float) value;
    else if (k == "cert_unsettled" && value > 0) ctx->tune_cert_unsettled = (float) value;
    else if (k == "cert_pad_mul" && value >= 0) ctx->tune_cert_pad_mul = (float) value;
    else if (k == "cert_pad_frac" && value >= 0) ctx->tune_cert_pad_frac = (float) value;
    else if (k == "late") ctx->tune_late = value != 0 ? 1 : 0;
    else if (k == "bins") ctx->tune_bins = (int) value;
    else if (k == "ndt_keys64") {
        ctx->tune_ndt_keys64 = value != 0 ? 1 : 0;
        ctx->ndt_built = false;
    } else if (k == "ndt_vox_split") {  // developer: who forms a voxel's sums (wm_ndt.hip); the model is rebuilt
        ctx->tune_ndt_vox_split = (int) value;
        ctx->ndt_built = false;
    }
    else if (k == "gicp_served") ctx->tune_gicp_served = value == 2 ? 2 : (value != 0 ? 1 : 0);
    else if (k == "gicp_serve_test_stall_ms") ctx->gicp_serve_test_stall_ms = (int) value;
    else return WM_ERR_ARG;
    return WM_OK;
}

int wm_debug_cert_log(wm_ctx *ctx, int iterations, unsigned *out, int cap) {
    if (!ctx || iterations < 0) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (!out) {  // arm: the next aligns count the queries k_nn_cert had to search, launch by launch
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->cert_log_iter = 0;
        ctx->cert_log_cap = 0;
        if (iterations == 0) {
            ctx->cert_count.release();
            return WM_OK;
        }
        WM_HIP(ctx, ctx->cert_count.reserve((size_t) iterations * 64 * sizeof(unsigned)));
        WM_HIP(ctx, hipMemsetAsync(ctx->cert_count.p, 0, (size_t) iterations * 64 * sizeof(unsigned), ctx->stream));
        if (getenv("WM_CERT_PROF")) {
            WM_HIP(ctx, ctx->cert_prof.reserve((size_t) iterations * 64 * sizeof(unsigned long long)));
            WM_HIP(ctx, hipMemsetAsync(ctx->cert_prof.p, 0, (size_t) iterations * 64 * sizeof(unsigned long long), ctx->stream));
        }
        ctx->cert_log_cap = iterations;
        return WM_OK;
    }
    const int n = ctx->cert_log_iter < cap ? ctx->cert_log_iter : cap;
    std::vector<unsigned> tmp((size_t) (n > 0 ? n : 1) * 64);
    if (n > 0) WM_TRY(copy_to_caller(ctx, tmp.data(), ctx->cert_count.p, (size_t) n * 64 * sizeof(unsigned)));
    for (int i = 0; i < n; ++i) {
        unsigned t = 0;
        for (int k = 0; k < 64; ++k) t += tmp[(size_t) i * 64 + k];
        out[i] = t;
    }
    return n;
}

int wm_debug_pub_log(wm_ctx *ctx, unsigned long long *out, int cap) {
    if (!ctx || !out || !ctx->h_pub) return WM_ERR_ARG;
    const int n = ctx->h_pub_slots < cap ? ctx->h_pub_slots : cap;
    for (int k = 0; k < n; ++k) out[k] = ctx->h_pub[k];
    return n;
}

int wm_debug_cert_prof(wm_ctx *ctx, unsigned long long *out, int cap) {
    if (!ctx || !out || !ctx->cert_prof.p) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    const int n = ctx->cert_log_iter < cap ? ctx->cert_log_iter : cap;
    if (n > 0) WM_TRY(copy_to_caller(ctx, out, ctx->cert_prof.p, (size_t) n * 64 * sizeof(unsigned long long)));
    return n;
}

int wm_get_iteration_times(wm_ctx *ctx, float *nn_ms, int cap) {
    if (!ctx || !nn_ms || cap < 0) return 0;
    int n = (int) ctx->iter_nn_ms.size();
    if (n > cap) n = cap;
    for (int i = 0; i < n; ++i) nn_ms[i] = ctx->iter_nn_ms[i];
    return n;
}

int wm_get_correspondences(wm_ctx *ctx, int32_t *match_idx, float *d2, size_t cap) {
    if (!ctx) return WM_ERR_ARG;
    if (!ctx->have_corr) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    return unpack_correspondences(ctx, match_idx, d2, cap);
}

int wm_nn_search(wm_ctx *ctx, const double T[16], double max_corr, int nn_method,
                 int32_t *match_idx, float *d2, size_t cap, float *kernel_ms) {
    if (!ctx || !T || !(max_corr > 0)) return WM_ERR_ARG;
    if (ctx->n_src_input == 0 || ctx->n_tgt_input == 0) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(finalize_clouds(ctx, max_corr, nn_method & ~WM_NN_WARM));
    WM_TRY(prepare_work(ctx));
    const bool warm = (nn_method & WM_NN_WARM) && ctx->have_corr;
    nn_method &= ~WM_NN_WARM;
    const bool brute = use_brute(ctx, nn_method) || ctx->n_tgt == 0;
    if (!brute) WM_TRY(ensure_levels(ctx, max_corr));
    init_state(ctx->h_state, T, nullptr, DBL_MAX);
    ctx->h_state->have_prev = warm ? 1 : 0;
    WM_TRY(upload_state(ctx));
    const float thr = threshold_d2(max_corr);
    hipEvent_t e0 = kernel_ms ? ctx->ev_a : nullptr, e1 = kernel_ms ? ctx->ev_b : nullptr;
    if (brute)
        WM_TRY(launch_nn_brute(ctx, thr, e0, e1));
    else
        WM_TRY(launch_nn_grid(ctx, thr, e0, e1, nullptr));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (kernel_ms) (void) hipEventElapsedTime(kernel_ms, ctx->ev_a, ctx->ev_b);
    ctx->have_corr = true;
    ctx->last_align_valid = false;
    memcpy(ctx->corr_T, T, sizeof(double) * 16);
    if (match_idx || d2) return unpack_correspondences(ctx, match_idx, d2, cap);
    return WM_OK;
}

int wm_icp_stats_for(wm_ctx *ctx, const double T[16], int mode, double stats[WM_STATS_LEN]) {
    if (!ctx || !T || !stats || (mode != WM_ICP_SVD && mode != WM_ICP_GN6)) return WM_ERR_ARG;
    if (!ctx->have_corr) return WM_ERR_STATE;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_TRY(prepare_work(ctx));
    wm_icp_params p;
    wm_icp_default_params(&p);
    p.mode = mode;
    init_state(ctx->h_state, T, &p, DBL_MAX);
    WM_TRY(upload_state(ctx));
    WM_TRY(launch_stats(ctx, mode));
    WM_TRY(launch_reduce_solve<1>(ctx, (unsigned) stat_blocks(ctx->n_src), nullptr));
    WM_TRY(download_state(ctx));
    memcpy(stats, ctx->h_state->stats, sizeof(double) * WM_STATS_LEN);
    return WM_OK;
}

int wm_umeyama_from_stats(const double stats[WM_STATS_LEN], double Tk_out[16]) {
    if (!stats || !Tk_out) return WM_ERR_ARG;
    if (!(stats[0] >= 3.0)) return WM_TOO_FEW_CORRESPONDENCES;
    umeyama_from_stats(stats, Tk_out);
    return WM_OK;
}

int wm_gn6_from_stats(const double stats[WM_STATS_LEN], double Tk_out[16]) {
    if (!stats || !Tk_out) return WM_ERR_ARG;
    if (!(stats[0] >= 3.0)) return WM_TOO_FEW_CORRESPONDENCES;
    return gn6_from_stats(stats, Tk_out) ? WM_OK : WM_NOT_CONVERGED;
}

}  // extern "C"
