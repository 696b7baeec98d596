// wm_shard.hip -- one ICP registration spread over the GPUs of a node, driven from C++:
// slab planning on the device, the iteration loop on the host side of this library, and the
// per-iteration exchange as ONE ncclAllReduce of WM_STATS_LEN doubles on the context's stream
// (RCCL over xGMI; librccl is linked directly).  No Python and no torch in this path.
//
// The reference has nothing to port here: its only parallelism is one matcher per thread
// (wave_matching/include/wave/matching/multi_matcher.hpp:32).  north_star: "partition the target
// cloud across the 8 GPUs of one node with RCCL all-reduce over xGMI of the normal equations only".
//
// Partition (exact, no per-point exchange): rank r indexes the target points of one
// x-slab [lo_r, hi_r) widened by a max_corr halo, and handles the source points whose TRANSFORMED x
// lies in the slab -- so every source point is handled by exactly one rank, and its true neighbour
// within max_corr is in that rank's subset.  Slab edges are equal-count quantiles of the target's
// x coordinates, found from a 64 Ki-bin histogram: integer counts, so every rank computes the
// same edges from the same cloud without talking to the others.  A rank keeps only the band of
// source points within max_corr of its slab; the all-reduced count of handled points is checked
// every iteration (wm_icp_shard_begin), and if a point ever left all bands the registration is
// redone with full source clouds -- the same verdict on every rank, it comes from the reduced block.
#include <cstring>  // before rocprim (its headers use memcpy unqualified)

#include <rocprim/rocprim.hpp>

#include "wm_internal.hpp"

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <float.h>
#include <math.h>

// ---------------------------------------------------------------- communicators
// The exchange step behind one interface: RCCL (one rank per process or per thread, one GPU each),
// or an in-process stand-in whose ranks share ONE GPU and add their blocks on the host in rank
// order at a barrier (what the single-GPU tests run; it shows what the exchange has to guarantee:
// bit-identical sums on every rank).
struct wm_local_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long generation = 0;
    std::vector<std::vector<double>> slots;
    std::vector<double> sum;
};

struct wm_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;
    wm_local_group *local = nullptr;  // shared by the ranks of an emulated group; freed by rank 0's destroy
    std::string last_error;
};

namespace wm {

int comm_allreduce(wm_ctx *ctx, wm_comm *comm, double *dev, int n) {
    if (!comm) return WM_OK;
    if (comm->nccl) {  // (also for a world of one: the plumbing check of bench.py / the tests)
        const ncclResult_t r = ncclAllReduce(dev, dev, (size_t) n, ncclDouble, ncclSum, comm->nccl, ctx->stream);
        if (r != ncclSuccess) {
            ctx->last_error = std::string("ncclAllReduce: ") + ncclGetErrorString(r);
            return WM_ERR_RCCL;
        }
        return WM_OK;
    }
    if (comm->world == 1) return WM_OK;
    wm_local_group *g = comm->local;
    if (!g) return WM_ERR_ARG;
    // emulation: device -> host, barrier, rank-ordered sum, host -> device
    std::vector<double> mine((size_t) n);
    WM_HIP(ctx, hipMemcpyAsync(mine.data(), dev, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->slots[(size_t) comm->rank] = mine;
        const unsigned long gen = g->generation;
        if (++g->arrived == g->world) {
            g->sum.assign((size_t) n, 0.0);
            for (int r = 0; r < g->world; ++r)
                for (int k = 0; k < n; ++k) g->sum[(size_t) k] += g->slots[(size_t) r][(size_t) k];
            g->arrived = 0;
            ++g->generation;
            g->cv.notify_all();
        } else if (!g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->generation != gen; })) {
            --g->arrived;  // a rank went missing (it failed): give up instead of hanging
            ctx->last_error = "emulated all-reduce: a rank did not arrive";
            return WM_ERR_STATE;
        }
        mine = g->sum;  // (the next generation cannot overwrite it before every rank has left
                        //  this one: a rank re-enters only through the next all-reduce, which
                        //  completes only when all have arrived again)
    }
    WM_HIP(ctx, hipMemcpyAsync(dev, mine.data(), (size_t) n * 8, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WM_OK;
}

// ------------------------------------------------------------- slab planning
// Every rank has (or receives) both clouds as the caller laid them out -- `stride` bytes per point,
// x y z first -- and plans WITHOUT talking to the others, from quantities that are the same on
// every rank by construction:
//   * the slab edges are equal-count quantiles of the target's x, from a 64 Ki-bin histogram of a
//     FIXED sub-sample (every world-th point: the same points on every rank, an eighth of the work at
//     eight ranks, and a million samples place a quantile to a few parts in a thousand);
//   * the rank's target slab + halo and its band of the source are selected straight out of the raw
//     clouds in ONE pass each (rocPRIM's single-pass select over a loading iterator: 12-16 B read per
//     point, the band written; round 2 packed, boxed and flag-scan-moved both full clouds: ~10x the
//     traffic and a dozen launches);
//   * the number of finite source points -- what the per-iteration ownership check compares with --
//     is counted per rank over its 1 / world stripe of the source and summed by the all-reduce that
//     runs anyway (a spare slot of the block).
constexpr int kHistBins = 1 << 16;
constexpr int kRangeBlocks = 256;

__device__ __forceinline__ float raw_x(const unsigned char *raw, size_t stride, size_t i) {
    return *reinterpret_cast<const float *>(raw + i * stride);
}

// per-block min / max of the finite x of the sub-sample (points 0, step, 2 step, ...)
__global__ void __launch_bounds__(kBlock)
    k_xrange(const unsigned char *__restrict__ raw, size_t stride, unsigned n, unsigned step, float *__restrict__ part) {
    float lo = INFINITY, hi = -INFINITY;
    const unsigned m = (n + step - 1) / step;
    for (unsigned k = blockIdx.x * kBlock + threadIdx.x; k < m; k += gridDim.x * kBlock) {
        const float x = raw_x(raw, stride, (size_t) k * step);
        if (x - x == 0.f) {  // finite
            lo = fminf(lo, x);
            hi = fmaxf(hi, x);
        }
    }
    __shared__ float s_lo[kBlock], s_hi[kBlock];
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if ((int) threadIdx.x < w) {
            s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + w]);
            s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s_lo[0];
        part[2 * blockIdx.x + 1] = s_hi[0];
    }
}

// (lo, bin width) of the histogram from the range partials: every block of the kernels below forms
// them again from the kRangeBlocks partials -- cheaper than a launch and a round trip for two numbers
__device__ __forceinline__ void range_of(const float *__restrict__ part, float *lo_out, float *w_out) {
    __shared__ float r_lo[kBlock], r_hi[kBlock];
    float lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < kRangeBlocks; k += kBlock) {
        lo = fminf(lo, part[2 * k]);
        hi = fmaxf(hi, part[2 * k + 1]);
    }
    r_lo[threadIdx.x] = lo;
    r_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if ((int) threadIdx.x < w) {
            r_lo[threadIdx.x] = fminf(r_lo[threadIdx.x], r_lo[threadIdx.x + w]);
            r_hi[threadIdx.x] = fmaxf(r_hi[threadIdx.x], r_hi[threadIdx.x + w]);
        }
        __syncthreads();
    }
    lo = r_lo[0];
    hi = r_hi[0];
    if (!(lo <= hi)) lo = hi = 0.f;  // no finite point at all
    *lo_out = lo;
    *w_out = fmaxf((hi - lo) / (float) kHistBins, 1e-30f);
}

// histogram of the sub-sample's finite x (LDS is too small for 64 Ki bins, and a cloud's points
// arrive in no particular x order, so the atomics spread over the whole table)
__global__ void __launch_bounds__(kBlock)
    k_xhist(const unsigned char *__restrict__ raw, size_t stride, unsigned n, unsigned step,
            const float *__restrict__ part, unsigned *__restrict__ hist) {
    float lo, w;
    range_of(part, &lo, &w);
    const float inv_w = 1.0f / w;
    const unsigned m = (n + step - 1) / step;
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= m) return;
    const float x = raw_x(raw, stride, (size_t) k * step);
    if (!(x - x == 0.f)) return;
    int b = (int) ((x - lo) * inv_w);
    b = min(max(b, 0), kHistBins - 1);
    atomicAdd(&hist[b], 1u);
}

// edges[0] = -inf, edges[world] = +inf, edges[k] = upper boundary of the first bin at which the
// running count reaches k / world of the points (one workgroup; 64 Ki bins = 256 per thread)
__global__ void __launch_bounds__(kBlock)
    k_plan_edges(const unsigned *__restrict__ hist, const float *__restrict__ part, int world, float *__restrict__ edges) {
    float lo, w;
    range_of(part, &lo, &w);
    __shared__ unsigned long long s_pre[kBlock + 1];
    constexpr int kPer = kHistBins / kBlock;
    unsigned long long mine = 0;
    for (int k = 0; k < kPer; ++k) mine += hist[threadIdx.x * kPer + k];
    s_pre[threadIdx.x + 1] = mine;
    if (threadIdx.x == 0) s_pre[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int t = 1; t <= kBlock; ++t) s_pre[t] += s_pre[t - 1];
    __syncthreads();
    const unsigned long long total = s_pre[kBlock];
    if (threadIdx.x == 0) {
        edges[0] = -INFINITY;
        edges[world] = INFINITY;
    }
    // thread t owns bins [t kPer, (t + 1) kPer): it emits every edge whose target count falls there
    unsigned long long run = s_pre[threadIdx.x];
    for (int k = 0; k < kPer; ++k) {
        const unsigned long long before = run;
        run += hist[threadIdx.x * kPer + k];
        for (int e = 1; e < world; ++e) {
            const unsigned long long want = (total * (unsigned long long) e + (unsigned long long) world - 1ull) /
                                            (unsigned long long) world;  // ceil(total e / world)
            if (want > before && want <= run) edges[e] = lo + (float) (threadIdx.x * kPer + k + 1) * w;
        }
    }
    if (total == 0ull && threadIdx.x == 0)
        for (int e = 1; e < world; ++e) edges[e] = INFINITY;
}

// finite points of the stripe [i0, i1) of a raw cloud, one partial count per block
__global__ void __launch_bounds__(kBlock)
    k_count_finite(const unsigned char *__restrict__ raw, size_t stride, unsigned i0, unsigned i1,
                   unsigned *__restrict__ part) {
    unsigned c = 0;
    for (unsigned i = i0 + blockIdx.x * kBlock + threadIdx.x; i < i1; i += gridDim.x * kBlock) {
        const float *q = reinterpret_cast<const float *>(raw + (size_t) i * stride);
        const float x = q[0], y = q[1], z = q[2];
        c += (x - x == 0.f && y - y == 0.f && z - z == 0.f) ? 1u : 0u;
    }
    __shared__ unsigned s_c[kBlock];
    s_c[threadIdx.x] = c;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if ((int) threadIdx.x < w) s_c[threadIdx.x] += s_c[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s_c[0];
}

// a raw cloud as a sequence of float4 (x, y, z, index bits), and the band test on it
struct RawLoad {
    const unsigned char *raw;
    size_t stride;
    __device__ float4 operator()(unsigned i) const {
        const float *q = reinterpret_cast<const float *>(raw + (size_t) i * stride);
        return make_float4(q[0], q[1], q[2], __uint_as_float(i));
    }
};
struct InBand {  // x within [edges[rank] - widen, edges[rank + 1] + widen] (a NaN never passes)
    const float *edges;
    int rank;
    float widen;
    __device__ bool operator()(const float4 &p) const {
        return p.x >= edges[rank] - widen && p.x <= edges[rank + 1] + widen;
    }
};

struct ShardPlan {
    float edges[2];      // this rank's [lo, hi)
    unsigned n_tgt_local, n_src_local, stripe_finite, pad;
};
__global__ void k_plan_pack(const float *__restrict__ edges, int rank, const unsigned *__restrict__ n_t,
                            const unsigned *__restrict__ n_s, const unsigned *__restrict__ fin_part, unsigned fin_blocks,
                            ShardPlan *out) {
    if (threadIdx.x != 0) return;
    out->edges[0] = edges[rank];
    out->edges[1] = edges[rank + 1];
    out->n_tgt_local = *n_t;
    out->n_src_local = n_s ? *n_s : 0u;
    unsigned c = 0;
    for (unsigned b = 0; b < fin_blocks; ++b) c += fin_part[b];
    out->stripe_finite = c;
    out->pad = 0;
}

// stable selection of a raw cloud's points inside the rank's band: out <- (x, y, z, index), *count <- how many
static int select_band(wm_ctx *ctx, const unsigned char *raw, size_t stride, unsigned n, const float *edges_dev,
                       int rank, float widen, float4 *out, unsigned *count_dev) {
    if (n == 0) {
        WM_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(unsigned), ctx->stream));
        return WM_OK;
    }
    using In = rocprim::transform_iterator<rocprim::counting_iterator<unsigned>, RawLoad, float4>;
    const In in(rocprim::counting_iterator<unsigned>(0u), RawLoad{raw, stride});
    const InBand pred{edges_dev, rank, widen};
    size_t bytes = 0;
    WM_HIP(ctx, rocprim::select(nullptr, bytes, in, out, count_dev, (size_t) n, pred, ctx->stream));
    WM_HIP(ctx, ctx->shard_flags.reserve(bytes + 64));
    WM_HIP(ctx, rocprim::select(ctx->shard_flags.p, bytes, in, out, count_dev, (size_t) n, pred, ctx->stream));
    return WM_OK;
}

}  // namespace wm

using namespace wm;

extern "C" {

int wm_comm_get_unique_id(void *id_out) {
    if (!id_out) return WM_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) <= WM_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return WM_ERR_RCCL;
    memset(id_out, 0, WM_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return WM_OK;
}

int wm_comm_init_rank(wm_comm **out, int device, const void *id_bytes, int rank, int world) {
    if (!out || !id_bytes || world < 1 || rank < 0 || rank >= world) return WM_ERR_ARG;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return WM_ERR_HIP;
    wm_comm *c = new (std::nothrow) wm_comm();
    if (!c) return WM_ERR_NOMEM;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    if (ncclCommInitRank(&c->nccl, world, id, rank) != ncclSuccess) {
        delete c;
        return WM_ERR_RCCL;
    }
    *out = c;
    return WM_OK;
}

int wm_comm_init_all(wm_comm **comms, const int *devices, int n) {
    if (!comms || !devices || n < 1) return WM_ERR_ARG;
    std::vector<ncclComm_t> nc((size_t) n);
    if (ncclCommInitAll(nc.data(), n, devices) != ncclSuccess) return WM_ERR_RCCL;
    for (int r = 0; r < n; ++r) comms[r] = nullptr;
    for (int r = 0; r < n; ++r) {
        wm_comm *c = new (std::nothrow) wm_comm();
        if (!c) {  // give everything back: the handles not yet wrapped, and the wrappers made so far
            for (int k = r; k < n; ++k) (void) ncclCommDestroy(nc[(size_t) k]);
            for (int k = 0; k < r; ++k) {
                wm_comm_destroy(comms[k]);
                comms[k] = nullptr;
            }
            return WM_ERR_NOMEM;
        }
        c->rank = r;
        c->world = n;
        c->device = devices[r];
        c->nccl = nc[(size_t) r];
        comms[r] = c;
    }
    return WM_OK;
}

int wm_comm_init_local(wm_comm **comms, int n, int device) {
    if (!comms || n < 1) return WM_ERR_ARG;
    wm_local_group *g = new (std::nothrow) wm_local_group();
    if (!g) return WM_ERR_NOMEM;
    g->world = n;
    g->slots.resize((size_t) n);
    for (int r = 0; r < n; ++r) comms[r] = nullptr;
    for (int r = 0; r < n; ++r) {
        wm_comm *c = new (std::nothrow) wm_comm();
        if (!c) {
            for (int k = 0; k < r; ++k) {
                comms[k]->local = nullptr;  // (the group is freed below, once)
                wm_comm_destroy(comms[k]);
                comms[k] = nullptr;
            }
            delete g;
            return WM_ERR_NOMEM;
        }
        c->rank = r;
        c->world = n;
        c->device = device;
        c->local = g;
        comms[r] = c;
    }
    return WM_OK;
}

void wm_comm_destroy(wm_comm *c) {
    if (!c) return;
    if (c->nccl) (void) ncclCommDestroy(c->nccl);
    if (c->local && c->rank == 0) delete c->local;
    delete c;
}

// how long one all-reduce of WM_STATS_LEN doubles takes on this communicator, back to back on the
// context's stream (us per all-reduce; collective: every rank calls it)
int wm_comm_allreduce_probe(wm_ctx *ctx, wm_comm *comm, int reps, double *us_out) {
    if (!ctx || !comm || !us_out || reps < 1) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_HIP(ctx, ctx->shard_stats.reserve(WM_STATS_LEN * sizeof(double)));
    double *blk = ctx->shard_stats.as<double>();
    WM_HIP(ctx, hipMemsetAsync(blk, 0, WM_STATS_LEN * sizeof(double), ctx->stream));
    for (int r = 0; r < 3; ++r) WM_TRY(wm::comm_allreduce(ctx, comm, blk, WM_STATS_LEN));
    WM_HIP(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    for (int r = 0; r < reps; ++r) WM_TRY(wm::comm_allreduce(ctx, comm, blk, WM_STATS_LEN));
    WM_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    WM_HIP(ctx, hipEventSynchronize(ctx->ev_b));
    float ms = 0;
    (void) hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
    *us_out = (double) ms * 1e3 / reps;
    return WM_OK;
}

int wm_comm_rank(const wm_comm *c) { return c ? c->rank : -1; }
int wm_comm_world(const wm_comm *c) { return c ? c->world : 0; }

// One registration, sharded.  Collective: every rank calls it with the same two clouds and the
// same parameters.  Everything that depends on the clouds -- slab edges, the rank's slab + halo of
// the target, its band of the source, the index over them -- is (re)computed inside the call.
// Host clouds: with an RCCL communicator rank 0 uploads them once and they travel to the other
// ranks over xGMI (ncclBroadcast); otherwise every rank uploads for itself.
static int align_sharded_impl(wm_ctx *ctx, wm_comm *comm, const void *ref, size_t n_ref, const void *target,
                              size_t n_target, size_t stride, int mem, const wm_icp_params *p, double T_out[16],
                              wm_icp_stats *stats) {
    if (!ctx || !p || !T_out || (n_ref > 0 && !ref) || (n_target > 0 && !target) || stride < 12 || (stride & 3) ||
        n_ref > 0x7FFFFFF0u || n_target > 0x7FFFFFF0u || !(p->max_corr > 0))
        return WM_ERR_ARG;
    if (p->mode != WM_ICP_SVD && p->mode != WM_ICP_GN6) return WM_ERR_ARG;
    if (p->force_iterations <= 0 && p->max_iter <= 0) return WM_ERR_ARG;
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (world == 1 && !(comm && comm->nccl && ctx->tune_force_shard)) {  // nothing to shard
        WM_TRY(wm_set_source(ctx, ref, n_ref, stride, mem));
        WM_TRY(wm_set_target(ctx, target, n_target, stride, mem));
        return wm_icp_align(ctx, p, T_out, stats);
    }
    if (world > 500) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_ref == 0 || n_target == 0) {
        if (stats) stats->state = WM_CONV_NO_CORRESPONDENCES;
        return n_ref == 0 && n_target == 0 ? WM_ERR_STATE : WM_TOO_FEW_CORRESPONDENCES;
    }
    const unsigned nr = (unsigned) n_ref, nt = (unsigned) n_target;
    wm_icp_stats st;
    memset(&st, 0, sizeof(st));
    if (comm && comm->nccl) {
        int cnt = 0;
        if (ncclCommCount(comm->nccl, &cnt) == ncclSuccess) st.rccl_ranks = cnt;
    }
    hipEvent_t e_a = ctx->ev_a, e_b = ctx->ev_b;
    WM_HIP(ctx, hipEventRecord(e_a, ctx->stream));
    // ---- the clouds, as laid out by the caller, in device memory
    const unsigned char *raw_ref = static_cast<const unsigned char *>(ref);
    const unsigned char *raw_tgt = static_cast<const unsigned char *>(target);
    if (mem == WM_MEM_HOST) {
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the staging buffers may still feed an earlier call)
        WM_HIP(ctx, ctx->shard_ref.reserve(n_ref * stride));
        WM_HIP(ctx, ctx->shard_tgt.reserve(n_target * stride));
        const bool bcast = comm && comm->nccl && world > 1;
        if (!bcast || rank == 0) {
            WM_HIP(ctx, hipMemcpy(ctx->shard_ref.p, ref, n_ref * stride, hipMemcpyHostToDevice));
            WM_HIP(ctx, hipMemcpy(ctx->shard_tgt.p, target, n_target * stride, hipMemcpyHostToDevice));
        }
        if (bcast) {  // one trip over PCIe, the rest over xGMI
            ncclResult_t r1 = ncclBroadcast(ctx->shard_ref.p, ctx->shard_ref.p, n_ref * stride, ncclUint8, 0, comm->nccl,
                                            ctx->stream);
            ncclResult_t r2 = r1 == ncclSuccess ? ncclBroadcast(ctx->shard_tgt.p, ctx->shard_tgt.p, n_target * stride,
                                                                ncclUint8, 0, comm->nccl, ctx->stream)
                                                : r1;
            if (r2 != ncclSuccess) {
                ctx->last_error = std::string("ncclBroadcast: ") + ncclGetErrorString(r2);
                return WM_ERR_RCCL;
            }
        }
        raw_ref = ctx->shard_ref.as<unsigned char>();
        raw_tgt = ctx->shard_tgt.as<unsigned char>();
    }
    // ---- the plan (see "slab planning" above): all on the device, one fetch of 24 bytes at the end
    WM_HIP(ctx, ctx->shard_ref_band.reserve((size_t) nr * sizeof(float4)));
    WM_HIP(ctx, ctx->shard_tgt_band.reserve((size_t) nt * sizeof(float4)));
    WM_HIP(ctx, ctx->shard_misc.reserve((size_t) kHistBins * 4 + 16384));
    unsigned *hist = ctx->shard_misc.as<unsigned>();
    float *edges = reinterpret_cast<float *>(hist + kHistBins);               // [world + 1] (<= 501)
    float *range_part = edges + 512;                                          // [kRangeBlocks][2]
    unsigned *fin_part = reinterpret_cast<unsigned *>(range_part + 2 * kRangeBlocks);  // [256]
    unsigned *counts = fin_part + 256;                                        // [0] target band, [1] source band
    ShardPlan *plan_dev = reinterpret_cast<ShardPlan *>(counts + 8);
    const unsigned step = (unsigned) world;  // the sub-sample every rank histograms
    const unsigned m_sub = (nt + step - 1) / step;
    WM_HIP(ctx, hipMemsetAsync(hist, 0, (size_t) kHistBins * 4, ctx->stream));
    hipLaunchKernelGGL(k_xrange, dim3(kRangeBlocks), dim3(kBlock), 0, ctx->stream, raw_tgt, stride, nt, step, range_part);
    hipLaunchKernelGGL(k_xhist, dim3((m_sub + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, raw_tgt, stride, nt,
                       step, range_part, hist);
    hipLaunchKernelGGL(k_plan_edges, dim3(1), dim3(kBlock), 0, ctx->stream, hist, range_part, world, edges);
    // finite source points of this rank's stripe
    const unsigned s0 = (unsigned) ((unsigned long long) nr * (unsigned) rank / (unsigned) world);
    const unsigned s1 = (unsigned) ((unsigned long long) nr * (unsigned) (rank + 1) / (unsigned) world);
    hipLaunchKernelGGL(k_count_finite, dim3(256), dim3(kBlock), 0, ctx->stream, raw_ref, stride, s0, s1, fin_part);
    WM_HIP(ctx, hipGetLastError());
    WM_HIP(ctx, hipEventRecord(e_b, ctx->stream));
    // this rank's target slab + halo (a float32-safe halo: max_corr plus a hair for the rounding of x)
    // and source band (points that start within max_corr of the slab)
    const float halo = (float) (p->max_corr * (1.0 + 1e-6) + 1e-4);
    const float pad = (float) p->max_corr;
    hipEvent_t e_c = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool full_source = attempt == 1;
        WM_TRY(select_band(ctx, raw_tgt, stride, nt, edges, rank, halo, ctx->shard_tgt_band.as<float4>(), counts));
        if (full_source)
            WM_TRY(select_band(ctx, raw_ref, stride, nr, edges, 0, INFINITY, ctx->shard_ref_band.as<float4>(), counts + 1));
        else
            WM_TRY(select_band(ctx, raw_ref, stride, nr, edges, rank, pad, ctx->shard_ref_band.as<float4>(), counts + 1));
        hipLaunchKernelGGL(k_plan_pack, dim3(1), dim3(64), 0, ctx->stream, edges, rank, counts, counts + 1, fin_part, 256u,
                           plan_dev);
        WM_HIP(ctx, hipGetLastError());
        if (!e_c) {
            while (ctx->ev_pool.size() < 1) {
                hipEvent_t e;
                WM_HIP(ctx, hipEventCreate(&e));
                ctx->ev_pool.push_back(e);
            }
            e_c = ctx->ev_pool[0];
        }
        WM_HIP(ctx, hipEventRecord(e_c, ctx->stream));
        ShardPlan *plan = (ShardPlan *) pinned_scratch(ctx, sizeof(ShardPlan));
        if (!plan) return WM_ERR_HIP;
        WM_TRY(fast_fetch(ctx, plan, plan_dev, sizeof(ShardPlan)));
        const ShardPlan pl = *plan;
        const auto t_host0 = std::chrono::steady_clock::now();
        if (attempt == 0) {
            (void) hipEventSynchronize(e_c);  // (done long ago: the fetch behind it has arrived; this settles the runtime's view)
            (void) hipEventElapsedTime(&st.plan_ms, e_a, e_b);
            if (hipEventElapsedTime(&st.compact_ms, e_b, e_c) != hipSuccess) st.compact_ms = -1.f;
        }
        st.n_tgt_local = pl.n_tgt_local;
        st.n_src_local = pl.n_src_local;
        st.shard_attempts = attempt + 1;
        // ---- the local clouds and their index
        WM_TRY(wm_set_source(ctx, ctx->shard_ref_band.p, pl.n_src_local, sizeof(float4), WM_MEM_DEVICE));
        WM_TRY(wm_set_target(ctx, ctx->shard_tgt_band.p, pl.n_tgt_local, sizeof(float4), WM_MEM_DEVICE));
        ctx->shard_lo = pl.edges[0];
        ctx->shard_hi = pl.edges[1];
        bool brute = false;
        float thr = 0.f;
        WM_TRY(shard_begin(ctx, p, (double) pl.edges[0], (double) pl.edges[1], -1.0, (double) pl.stripe_finite, &brute, &thr,
                           (p->carry_state && ctx->prev_mse >= 0) ? ctx->prev_mse : DBL_MAX));
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const auto t_host1 = std::chrono::steady_clock::now();
        // ---- the iteration loop: search + local sums -> all-reduce of the block -> solve (wm_icp.hip)
        WM_HIP(ctx, ctx->shard_stats.reserve(kBlkLen * sizeof(double)));
        double T[16];
        wm_icp_stats it_st;
        memset(&it_st, 0, sizeof(it_st));  // (the loop ADDS its event times into the block)
        const int rc = icp_run_loop(ctx, p, brute, thr, comm, ctx->shard_stats.as<double>(), T, &it_st);
        const auto t_host2 = std::chrono::steady_clock::now();
        ctx->shard_active = false;
        if (rc < 0) return rc;
        // (plan / compact are this call's own; the loop's statistics take the rest of the block)
        const wm_icp_stats keep = st;
        st = it_st;
        st.plan_ms = keep.plan_ms;
        st.compact_ms = keep.compact_ms;
        st.n_tgt_local = keep.n_tgt_local;
        st.n_src_local = keep.n_src_local;
        st.rccl_ranks = keep.rccl_ranks;
        st.shard_attempts = keep.shard_attempts;
        st.index_ms = keep.index_ms + std::chrono::duration<float, std::milli>(t_host1 - t_host0).count();
        st.iter_ms = keep.iter_ms + std::chrono::duration<float, std::milli>(t_host2 - t_host1).count();
        if (stats) *stats = st;
        if (st.owned_violations > 0 && !full_source) continue;  // (same verdict on every rank: it comes from the reduced block)
        if (rc == WM_OK) memcpy(T_out, T, sizeof(T));
        return rc;
    }
    return WM_ERR_STATE;
}

int wm_icp_align_sharded(wm_ctx *ctx, wm_comm *comm, const void *ref, size_t n_ref, const void *target,
                         size_t n_target, size_t stride, int mem, const wm_icp_params *p, double T_out[16],
                         wm_icp_stats *stats) {
    const int rc = align_sharded_impl(ctx, comm, ref, n_ref, target, n_target, stride, mem, p, T_out, stats);
    // A rank that fails inside the collective part (a HIP or RCCL error: out of memory, a lost device)
    // would leave its peers waiting in ncclAllReduce for ever: abort the communicator, which fails the
    // peers' pending collectives too.  The communicator is finished after that; argument errors are
    // returned before anything collective has started and abort nothing.
    if ((rc == WM_ERR_HIP || rc == WM_ERR_RCCL || rc == WM_ERR_NOMEM) && comm && comm->nccl && comm->world > 1) {
        (void) ncclCommAbort(comm->nccl);
        comm->nccl = nullptr;
        comm->last_error = "aborted after a failed sharded registration";
    }
    return rc;
}

// ICPMatcher::match() with a voxel filter (icp.cpp:77-122), sharded: every rank filters both clouds
// itself (pcl::VoxelGrid of a cloud is one sort -- deterministic, so all ranks hold the same filtered
// clouds; it is the registration of each scale, the part that grows with the iteration count, that
// is spread over the ranks), then one sharded align per scale on the filtered, device-resident clouds.
int wm_icp_match_sharded(wm_ctx *ctx, wm_comm *comm, const void *ref, size_t n_ref, const void *target,
                         size_t n_target, size_t stride, int mem, const wm_icp_params *p, float res,
                         int multiscale_steps, double T_out[16], wm_icp_stats *stats) {
    if (!ctx || !p || !T_out || (n_ref > 0 && !ref) || (n_target > 0 && !target) || stride < 12 || (stride & 3) ||
        n_ref > 0x7FFFFFF0u || n_target > 0x7FFFFFF0u)
        return WM_ERR_ARG;
    if (!(res > 0)) return wm_icp_align_sharded(ctx, comm, ref, n_ref, target, n_target, stride, mem, p, T_out, stats);
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    const size_t cap_r = n_ref > 0 ? n_ref : 1, cap_t = n_target > 0 ? n_target : 1;
    WM_HIP(ctx, ctx->match_ref.reserve(cap_r * sizeof(float4)));
    WM_HIP(ctx, ctx->match_tgt.reserve(cap_t * sizeof(float4)));
    WM_HIP(ctx, ctx->ds_ref.reserve(cap_r * sizeof(float4)));
    WM_HIP(ctx, ctx->ds_tgt.reserve(cap_t * sizeof(float4)));
    float4 *d_ref = ctx->match_ref.as<float4>(), *d_tgt = ctx->match_tgt.as<float4>();
    float4 *ds_ref = ctx->ds_ref.as<float4>(), *ds_tgt = ctx->ds_tgt.as<float4>();
    WM_TRY(pack_cloud(ctx, ref, n_ref, stride, mem, d_ref));
    WM_TRY(pack_cloud(ctx, target, n_target, stride, mem, d_tgt));
    wm_icp_params prm = *p;
    wm_icp_stats last, total;
    memset(&total, 0, sizeof(total));
    double running[16];
    mat4_identity(running);
    const int steps = multiscale_steps > 0 ? multiscale_steps : 0;
    VgKnown kr{}, kt{};
    if (steps > 0) {
        if (n_ref > 0) WM_TRY(compute_bbox(ctx, d_ref, n_ref, &kr.bb, &kr.valid));
        if (n_target > 0) WM_TRY(compute_bbox(ctx, d_tgt, n_target, &kt.bb, &kt.valid));
    }
    for (int i = steps; i >= 0; --i) {
        const float leaf = (float) (pow(2, i) * res);  // icp.cpp:80
        size_t nr = 0, nt = 0;
        WM_TRY(voxel_downsample_dev(ctx, d_ref, n_ref, leaf, ds_ref, &nr, steps > 0 && n_ref > 0 ? &kr : nullptr));
        WM_TRY(voxel_downsample_dev(ctx, d_tgt, n_target, leaf, ds_tgt, &nt, steps > 0 && n_target > 0 ? &kt : nullptr));
        if (steps > 0) {
            WM_TRY(transform_cloud_dev(ctx, ds_ref, nr, running, ds_ref));  // icp.cpp:84-86
            prm.max_corr = pow(2, i) * p->max_corr;                          // icp.cpp:93-94
        }
        double Ti[16];
        const int rc = wm_icp_align_sharded(ctx, comm, ds_ref, nr, ds_tgt, nt, sizeof(float4), WM_MEM_DEVICE, &prm, Ti, &last);
        total.align_ms += last.align_ms;
        total.nn_ms += last.nn_ms;
        total.nn_launches += last.nn_launches;
        if (stats) {
            *stats = last;
            stats->align_ms = total.align_ms;
            stats->nn_ms = total.nn_ms;
            stats->nn_launches = total.nn_launches;
        }
        if (rc != WM_OK) return rc;  // icp.cpp:96-98: fail fast, result untouched (the same verdict on every rank)
        mat4_mul(Ti, running, running);  // icp.cpp:99-101
    }
    memcpy(T_out, running, sizeof(running));
    return WM_OK;
}

}  // extern "C"

// ------------------------------------------------------------ all ranks in one process
struct wm_multi {
    std::vector<wm_ctx *> ctx;
    std::vector<wm_comm *> comm;
};

extern "C" {

int wm_multi_create(wm_multi **out, const int *devices, int n, int emulate) {
    if (!out || !devices || n < 1) return WM_ERR_ARG;
    *out = nullptr;
    wm_multi *m = new (std::nothrow) wm_multi();
    if (!m) return WM_ERR_NOMEM;
    m->ctx.assign((size_t) n, nullptr);
    m->comm.assign((size_t) n, nullptr);
    int rc = WM_OK;
    for (int r = 0; r < n && rc == WM_OK; ++r) rc = wm_ctx_create(&m->ctx[(size_t) r], emulate ? devices[0] : devices[r]);
    if (rc == WM_OK && n > 1)
        rc = emulate ? wm_comm_init_local(m->comm.data(), n, devices[0]) : wm_comm_init_all(m->comm.data(), devices, n);
    if (rc != WM_OK) {
        wm_multi_destroy(m);
        return rc;
    }
    *out = m;
    return WM_OK;
}

void wm_multi_destroy(wm_multi *m) {
    if (!m) return;
    for (size_t r = m->comm.size(); r-- > 0;) wm_comm_destroy(m->comm[r]);  // rank 0 (owner of a local group) last
    for (wm_ctx *c : m->ctx) wm_ctx_destroy(c);
    delete m;
}

int wm_multi_size(const wm_multi *m) { return m ? (int) m->ctx.size() : 0; }

int wm_multi_icp_align(wm_multi *m, const void *ref, size_t n_ref, const void *target, size_t n_target,
                       size_t stride, const wm_icp_params *p, double T_out[16], wm_icp_stats *stats) {
    return wm_multi_icp_match(m, ref, n_ref, target, n_target, stride, p, -1.f, 0, T_out, stats);
}

int wm_multi_icp_match(wm_multi *m, const void *ref, size_t n_ref, const void *target, size_t n_target,
                       size_t stride, const wm_icp_params *p, float res, int multiscale_steps, double T_out[16],
                       wm_icp_stats *stats) {
    if (!m || !p || !T_out) return WM_ERR_ARG;
    const int n = (int) m->ctx.size();
    if (n == 1)
        return wm_icp_match(m->ctx[0], ref, n_ref, target, n_target, stride, WM_MEM_HOST, p, res, multiscale_steps, T_out,
                            stats);
    std::vector<int> rcs((size_t) n, WM_ERR_STATE);
    std::vector<wm_icp_stats> sts((size_t) n);
    std::vector<double> Ts((size_t) n * 16, 0.0);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            rcs[(size_t) r] = wm_icp_match_sharded(m->ctx[(size_t) r], m->comm[(size_t) r], ref, n_ref, target, n_target,
                                                   stride, WM_MEM_HOST, p, res, multiscale_steps, &Ts[(size_t) r * 16],
                                                   &sts[(size_t) r]);
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (rcs[(size_t) r] < 0) return rcs[(size_t) r];
    if (stats) *stats = sts[0];
    if (rcs[0] == WM_OK) memcpy(T_out, Ts.data(), 16 * sizeof(double));
    return rcs[0];
}

// ICPMatcher::estimateInfo() after a registration over the group (wm_multi_icp_match): every rank adds up
// its own pairs, the sums are exchanged, every rank finishes the same 6x6 (wm_icp_info_sharded)
int wm_multi_icp_info(wm_multi *m, int method, const double T_result[16], double lin_covar, double ang_covar,
                      double max_corr, double info[36], int *degenerate) {
    if (!m || !info) return WM_ERR_ARG;
    const int n = (int) m->ctx.size();
    if (n == 1) return wm_icp_info(m->ctx[0], method, T_result, lin_covar, ang_covar, max_corr, info, degenerate);
    std::vector<int> rcs((size_t) n, WM_ERR_STATE), deg((size_t) n, 0);
    std::vector<double> infos((size_t) n * 36, 0.0);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            rcs[(size_t) r] = wm_icp_info_sharded(m->ctx[(size_t) r], m->comm[(size_t) r], method, T_result, lin_covar,
                                                  ang_covar, max_corr, &infos[(size_t) r * 36], &deg[(size_t) r]);
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (rcs[(size_t) r] < 0) return rcs[(size_t) r];
    if (rcs[0] == WM_OK) {
        memcpy(info, infos.data(), 36 * sizeof(double));
        if (degenerate) *degenerate = deg[0];
    }
    return rcs[0];
}

}  // extern "C"
