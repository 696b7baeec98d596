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
constexpr int kHistBins = 1 << 16;

// histogram of the finite points' x over [lo, hi]: LDS is too small for 64 Ki bins, and a cloud's
// points arrive in no particular x order, so the atomics spread over the whole table
__global__ void __launch_bounds__(kBlock)
    k_xhist(const float4 *__restrict__ pts, unsigned n, float lo, float inv_w, unsigned *__restrict__ hist) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float x = pts[i].x;
    if (!(x == x)) return;
    int b = (int) ((x - lo) * inv_w);
    b = min(max(b, 0), kHistBins - 1);
    atomicAdd(&hist[b], 1u);
}

// edges[0] = -inf, edges[world] = +inf, edges[k] = upper boundary of the first bin at which the
// running count reaches k / world of the points (one workgroup; 64 Ki bins = 256 per thread)
__global__ void __launch_bounds__(kBlock)
    k_plan_edges(const unsigned *__restrict__ hist, float lo, float w, int world, float *__restrict__ edges) {
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

// stable compaction of the points with x in [lo, hi] (NaN points never pass)
__global__ void __launch_bounds__(kBlock)
    k_band_flags(const float4 *__restrict__ pts, unsigned n, const float *__restrict__ edges, int rank,
                 float widen, unsigned *__restrict__ flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float x = pts[i].x;
    const float lo = edges[rank] - widen, hi = edges[rank + 1] + widen;
    flags[i] = (x >= lo && x <= hi) ? 1u : 0u;
}
__global__ void __launch_bounds__(kBlock)
    k_band_move(const float4 *__restrict__ pts, unsigned n, const unsigned *__restrict__ flags,
                const unsigned *__restrict__ pos, float4 *__restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) out[pos[i]] = pts[i];
}

struct ShardPlan {
    float edges[2];      // this rank's [lo, hi)
    unsigned n_tgt_local, n_src_local, n_src_finite, pad;
};
__global__ void k_plan_pack(const float *__restrict__ edges, int rank, const unsigned *__restrict__ pos_t,
                            unsigned n_t, const unsigned *__restrict__ pos_s, unsigned n_s,
                            const float *__restrict__ bbox_partials, unsigned bbox_blocks, ShardPlan *out) {
    if (threadIdx.x != 0) return;
    out->edges[0] = edges[rank];
    out->edges[1] = edges[rank + 1];
    out->n_tgt_local = n_t ? pos_t[n_t] : 0u;
    out->n_src_local = n_s ? pos_s[n_s] : 0u;
    unsigned c = 0;
    for (unsigned b = 0; b < bbox_blocks; ++b) c += __float_as_uint(bbox_partials[8 * b + 6]);
    out->n_src_finite = c;
    out->pad = 0;
}

static int compact_band(wm_ctx *ctx, const float4 *pts, unsigned n, const float *edges_dev, int rank, float widen,
                        DevBuf &flags, DevBuf &pos, float4 *out) {
    if (n == 0) return WM_OK;
    WM_HIP(ctx, flags.reserve(((size_t) n + 1) * 4));
    WM_HIP(ctx, pos.reserve(((size_t) n + 1) * 4));
    const unsigned blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_band_flags, dim3(blocks), dim3(kBlock), 0, ctx->stream, pts, n, edges_dev, rank, widen,
                       flags.as<unsigned>());
    WM_TRY(exclusive_scan(ctx, flags.as<unsigned>(), n, pos.as<unsigned>()));  // pos[n] = count
    hipLaunchKernelGGL(k_band_move, dim3(blocks), dim3(kBlock), 0, ctx->stream, pts, n, flags.as<unsigned>(),
                       pos.as<unsigned>(), out);
    WM_HIP(ctx, hipGetLastError());
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
    for (int r = 0; r < n; ++r) {
        wm_comm *c = new (std::nothrow) wm_comm();
        if (!c) return WM_ERR_NOMEM;
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
    for (int r = 0; r < n; ++r) {
        wm_comm *c = new (std::nothrow) wm_comm();
        if (!c) return WM_ERR_NOMEM;
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

int wm_comm_rank(const wm_comm *c) { return c ? c->rank : -1; }
int wm_comm_world(const wm_comm *c) { return c ? c->world : 0; }

// One registration, sharded.  Collective: every rank calls it with the same two clouds and the
// same parameters.  Everything that depends on the clouds -- slab edges, the rank's slab + halo of
// the target, its band of the source, the index over them -- is (re)computed inside the call.
int wm_icp_align_sharded(wm_ctx *ctx, wm_comm *comm, const void *ref, size_t n_ref, const void *target,
                         size_t n_target, size_t stride, int mem, const wm_icp_params *p, double T_out[16],
                         wm_icp_stats *stats) {
    if (!ctx || !p || !T_out || (n_ref > 0 && !ref) || (n_target > 0 && !target) || stride < 12 || (stride & 3) ||
        n_ref > 0x7FFFFFF0u || n_target > 0x7FFFFFF0u || !(p->max_corr > 0))
        return WM_ERR_ARG;
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (world == 1 && !(comm && comm->nccl && ctx->tune_force_shard)) {  // nothing to shard
        WM_TRY(wm_set_source(ctx, ref, n_ref, stride, mem));
        WM_TRY(wm_set_target(ctx, target, n_target, stride, mem));
        return wm_icp_align(ctx, p, T_out, stats);
    }
    WM_HIP(ctx, hipSetDevice(ctx->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_ref == 0 || n_target == 0) {
        if (stats) stats->state = WM_CONV_NO_CORRESPONDENCES;
        return n_ref == 0 && n_target == 0 ? WM_ERR_STATE : WM_TOO_FEW_CORRESPONDENCES;
    }
    const unsigned nr = (unsigned) n_ref, nt = (unsigned) n_target;
    // full clouds, packed
    WM_HIP(ctx, ctx->shard_ref.reserve((size_t) nr * sizeof(float4)));
    WM_HIP(ctx, ctx->shard_tgt.reserve((size_t) nt * sizeof(float4)));
    WM_HIP(ctx, ctx->shard_ref_band.reserve((size_t) nr * sizeof(float4)));
    WM_HIP(ctx, ctx->shard_tgt_band.reserve((size_t) nt * sizeof(float4)));
    WM_HIP(ctx, ctx->shard_misc.reserve((size_t) kHistBins * 4 + 4096 + 2 * 8 * sizeof(float) * kBboxBlocks));
    float4 *d_ref = ctx->shard_ref.as<float4>(), *d_tgt = ctx->shard_tgt.as<float4>();
    unsigned *hist = ctx->shard_misc.as<unsigned>();
    float *edges = reinterpret_cast<float *>(hist + kHistBins);               // [world + 1] (<= 1024 ranks)
    ShardPlan *plan_dev = reinterpret_cast<ShardPlan *>(edges + 512);
    float *bb_t = reinterpret_cast<float *>(hist + kHistBins + 1024);         // target bbox partials
    float *bb_s = bb_t + 8 * kBboxBlocks;                                     // source bbox partials
    if (world > 500) return WM_ERR_ARG;
    WM_TRY(pack_cloud(ctx, ref, n_ref, stride, mem, d_ref));
    WM_TRY(pack_cloud(ctx, target, n_target, stride, mem, d_tgt));
    // x range of the target (the host needs it for the histogram's scale: one round trip), and the
    // number of finite source points (stays on the device until the plan is fetched)
    unsigned bt = 0, bs = 0;
    WM_TRY(launch_bbox(ctx, d_tgt, nt, bb_t, &bt));
    WM_TRY(launch_bbox(ctx, d_ref, nr, bb_s, &bs));
    float *h = (float *) pinned_scratch(ctx, 8 * sizeof(float) * kBboxBlocks);
    if (!h) return WM_ERR_HIP;
    WM_TRY(fast_fetch(ctx, h, bb_t, 8 * sizeof(float) * bt));
    Bbox tb;
    size_t t_valid = 0;
    finish_bbox(h, bt, &tb, &t_valid);
    const float xlo = tb.lo[0], xw = fmaxf((tb.hi[0] - tb.lo[0]) / (float) kHistBins, 1e-30f);
    WM_HIP(ctx, hipMemsetAsync(hist, 0, (size_t) kHistBins * 4, ctx->stream));
    hipLaunchKernelGGL(k_xhist, dim3((nt + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, d_tgt, nt, xlo,
                       1.0f / xw, hist);
    hipLaunchKernelGGL(k_plan_edges, dim3(1), dim3(kBlock), 0, ctx->stream, hist, xlo, xw, world, edges);
    WM_HIP(ctx, hipGetLastError());
    // this rank's target slab + halo (a float32-safe halo: max_corr plus a hair for the rounding of x)
    // and source band (points that start within max_corr of the slab)
    const float halo = (float) (p->max_corr * (1.0 + 1e-6) + 1e-4);
    const float pad = (float) p->max_corr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool full_source = attempt == 1;
        WM_TRY(compact_band(ctx, d_tgt, nt, edges, rank, halo, ctx->shard_flags, ctx->shard_pos_t,
                            ctx->shard_tgt_band.as<float4>()));
        if (!full_source)
            WM_TRY(compact_band(ctx, d_ref, nr, edges, rank, pad, ctx->shard_flags, ctx->shard_pos_s,
                                ctx->shard_ref_band.as<float4>()));
        hipLaunchKernelGGL(k_plan_pack, dim3(1), dim3(64), 0, ctx->stream, edges, rank, ctx->shard_pos_t.as<unsigned>(),
                           nt, ctx->shard_pos_s.as<unsigned>(), full_source ? 0u : nr, bb_s, bs, plan_dev);
        WM_HIP(ctx, hipGetLastError());
        ShardPlan *plan = (ShardPlan *) pinned_scratch(ctx, sizeof(ShardPlan));
        if (!plan) return WM_ERR_HIP;
        WM_TRY(fast_fetch(ctx, plan, plan_dev, sizeof(ShardPlan)));
        const ShardPlan pl = *plan;
        if (full_source)
            WM_TRY(wm_set_source(ctx, d_ref, nr, sizeof(float4), WM_MEM_DEVICE));
        else
            WM_TRY(wm_set_source(ctx, ctx->shard_ref_band.p, pl.n_src_local, sizeof(float4), WM_MEM_DEVICE));
        WM_TRY(wm_set_target(ctx, ctx->shard_tgt_band.p, pl.n_tgt_local, sizeof(float4), WM_MEM_DEVICE));
        ctx->shard_lo = pl.edges[0];
        ctx->shard_hi = pl.edges[1];
        // the iteration loop: search + local sums -> all-reduce of the block -> solve, all enqueued on
        // the context's stream; the host looks at the state once per batch
        WM_TRY(wm_icp_shard_begin(ctx, p, (double) pl.edges[0], (double) pl.edges[1], pl.n_src_finite));
        WM_HIP(ctx, ctx->shard_stats.reserve(WM_STATS_LEN * sizeof(double)));
        double *blk = ctx->shard_stats.as<double>();
        const bool forced = p->force_iterations > 0;
        const int max_it = forced ? p->force_iterations : p->max_iter;
        int it = 0, done = 0, rc = WM_OK;
        double T[16];
        wm_icp_stats st;
        memset(&st, 0, sizeof(st));
        hipEvent_t e0 = ctx->ev_a, e1 = ctx->ev_b;
        WM_HIP(ctx, hipEventRecord(e0, ctx->stream));
        while (it < max_it && !done) {
            const int nb = forced ? max_it : (max_it - it < 8 ? max_it - it : 8);
            for (int k = 0; k < nb; ++k) {
                WM_TRY(wm_icp_shard_local_stats(ctx, blk));
                WM_TRY(comm_allreduce(ctx, comm, blk, WM_STATS_LEN));
                WM_TRY(wm_icp_shard_apply(ctx, blk));
            }
            it += nb;
            rc = wm_icp_shard_poll(ctx, &done, T, &st);
            if (rc < 0) return rc;
        }
        WM_HIP(ctx, hipEventRecord(e1, ctx->stream));
        WM_HIP(ctx, hipEventSynchronize(e1));
        (void) hipEventElapsedTime(&st.align_ms, e0, e1);
        if (stats) *stats = st;
        if (st.owned_violations > 0 && !full_source) continue;  // (same verdict on every rank)
        if (rc == WM_OK) memcpy(T_out, T, sizeof(T));
        return rc;
    }
    return WM_ERR_STATE;
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
    if (!m || !p || !T_out) return WM_ERR_ARG;
    const int n = (int) m->ctx.size();
    if (n == 1) return wm_icp_align_sharded(m->ctx[0], nullptr, ref, n_ref, target, n_target, stride, WM_MEM_HOST, p, T_out, stats);
    std::vector<int> rcs((size_t) n, WM_ERR_STATE);
    std::vector<wm_icp_stats> sts((size_t) n);
    std::vector<double> Ts((size_t) n * 16, 0.0);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            rcs[(size_t) r] = wm_icp_align_sharded(m->ctx[(size_t) r], m->comm[(size_t) r], ref, n_ref, target, n_target,
                                                   stride, WM_MEM_HOST, p, &Ts[(size_t) r * 16], &sts[(size_t) r]);
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (rcs[(size_t) r] < 0) return rcs[(size_t) r];
    if (stats) *stats = sts[0];
    if (rcs[0] == WM_OK) memcpy(T_out, Ts.data(), 16 * sizeof(double));
    return rcs[0];
}

}  // extern "C"
