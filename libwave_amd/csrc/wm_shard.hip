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
#include "wm_xchg.hpp"

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
    // mailboxes for the in-kernel exchange of the ICP loop's block (wm_xchg.hpp); ncclAllReduce where they are absent
    unsigned long long *mail = nullptr;        // this rank's mailbox: fine-grained device memory
    unsigned long long **peer_dev = nullptr;   // [world] in device memory: every rank's mailbox as this device sees it
    std::vector<void *> ipc_opened;            // peers' mailboxes mapped through IPC handles (other processes)
    bool p2p = false;
    unsigned p2p_timeout_ms = 5000;
};

namespace wm {

// ---------------------------------------------------------------- mailboxes
// (default on; WM_COMM_P2P=0 keeps ncclAllReduce as the loop's exchange)
static bool mailboxes_wanted() {
    const char *e = getenv("WM_COMM_P2P");
    return !(e && atoi(e) == 0);
}

static int mailbox_alloc(wm_comm *c) {
    if (c->world > kXMaxWorld) return WM_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return WM_ERR_HIP;
    const size_t bytes = xchg_mailbox_words(c->world) * sizeof(unsigned long long);
    void *p = nullptr;
    // remote stores must be seen by a kernel that is already polling: uncached (or at least fine-grained) memory
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess || !p) {
        (void) hipGetLastError();
        p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) {
            (void) hipGetLastError();
            return WM_ERR_NOMEM;
        }
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void) hipFree(p);
        return WM_ERR_HIP;
    }
    c->mail = static_cast<unsigned long long *>(p);
    if (const char *e = getenv("WM_COMM_P2P_TIMEOUT_MS")) {
        const int v = atoi(e);
        if (v >= 0) c->p2p_timeout_ms = (unsigned) v;  // (0: a block that is not there at the first look is "late" -- tests)
    }
    return WM_OK;
}

static int mailbox_set_peers(wm_comm *c, const std::vector<unsigned long long *> &peers) {
    if (hipSetDevice(c->device) != hipSuccess) return WM_ERR_HIP;
    void *d = nullptr;
    if (hipMalloc(&d, peers.size() * sizeof(void *)) != hipSuccess) return WM_ERR_NOMEM;
    if (hipMemcpy(d, peers.data(), peers.size() * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess) {
        (void) hipFree(d);
        return WM_ERR_HIP;
    }
    c->peer_dev = static_cast<unsigned long long **>(d);
    return WM_OK;
}

static void mailbox_free(wm_comm *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    for (void *q : c->ipc_opened) (void) hipIpcCloseMemHandle(q);
    c->ipc_opened.clear();
    if (c->peer_dev) (void) hipFree(c->peer_dev);
    if (c->mail) (void) hipFree(c->mail);
    c->peer_dev = nullptr;
    c->mail = nullptr;
    c->p2p = false;
}

static XchgDev mailbox_args(const wm_comm *c, unsigned timeout_ms) {
    return XchgDev{c->peer_dev, c->mail, c->world, c->rank, timeout_ms};
}

// one exchange on its own: block[k] = (k + 1) * (rank + 1) in, the sum over the ranks out; ok = it arrived
__global__ void __launch_bounds__(256) k_xchg_probe(XchgDev x, double *out, int *ok) {
    __shared__ double s_in[kBlkLen], s_out[kBlkLen];
    __shared__ unsigned s_half[kXMaxWorld * kXWords], s_ctl[2];
    if (threadIdx.x < kBlkLen) s_in[threadIdx.x] = (double) (threadIdx.x + 1) * (double) (x.rank + 1);
    __syncthreads();
    const bool arrived = xchg_allreduce<256>(x, s_in, s_out, s_half, s_ctl);
    if (threadIdx.x < kBlkLen && out) out[threadIdx.x] = s_out[threadIdx.x];
    if (threadIdx.x == 0 && ok) *ok = arrived ? 1 : 0;
}

// launch one probe exchange on `stream` (every rank of the group must, at about the same time)
static int mailbox_probe_launch(wm_comm *c, hipStream_t stream, double *out_dev, int *ok_dev, unsigned timeout_ms) {
    if (hipSetDevice(c->device) != hipSuccess) return WM_ERR_HIP;
    hipLaunchKernelGGL(k_xchg_probe, dim3(1), dim3(256), 0, stream, mailbox_args(c, timeout_ms), out_dev, ok_dev);
    return hipGetLastError() == hipSuccess ? WM_OK : WM_ERR_HIP;
}

// did the probe deliver the sum it must?  (host side, after the stream has been synchronised)
static bool mailbox_probe_good(const wm_comm *c, const double *out_host, int ok_host) {
    if (!ok_host) return false;
    const double ranks = 0.5 * (double) c->world * (double) (c->world + 1);  // sum of (rank + 1)
    for (int k = 0; k < kBlkLen; ++k)
        if (out_host[k] != (double) (k + 1) * ranks) return false;
    return true;
}

int comm_exchange_args(wm_comm *comm, XchgDev *out) {
    if (!comm || !comm->p2p || !comm->mail || !comm->peer_dev || !out) return WM_ERR_STATE;
    *out = mailbox_args(comm, comm->p2p_timeout_ms);
    return WM_OK;
}

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

__device__ __forceinline__ float raw_x(const unsigned char *raw, size_t stride, size_t i) {
    return *reinterpret_cast<const float *>(raw + i * stride);
}

// The sub-sample the slab edges are planned from: kPlanSamples points at a fixed stride through the target (the same
// points on every rank), their x gathered into one compact array (NaN for a non-finite point) with per-block min / max.
// (Round 2-4 histogrammed every world-th point into 64 Ki bins in HBM: 1M global atomics and a one-workgroup scan of
// the table, 110 us of the 130 a plan took.  Sixteen thousand samples place a quantile to 0.4 % of the cloud.)
constexpr unsigned kPlanSamples = 16384;
constexpr int kSampleBlocks = kPlanSamples / kBlock;
__global__ void __launch_bounds__(kBlock)
    k_xsample(const unsigned char *__restrict__ raw, size_t stride, unsigned n, unsigned step, float *__restrict__ xs,
              float *__restrict__ part) {
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    const unsigned m = (n + step - 1) / step;
    float x = NAN;
    if (k < m) x = raw_x(raw, stride, (size_t) k * step);
    if (!(x - x == 0.f)) x = NAN;
    xs[k] = x;
    float lo = x == x ? x : INFINITY, hi = x == x ? x : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    __shared__ float s_lo[kBlock / 64], s_hi[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            lo = fminf(lo, s_lo[w]);
            hi = fmaxf(hi, s_hi[w]);
        }
        part[2 * blockIdx.x] = lo;
        part[2 * blockIdx.x + 1] = hi;
    }
}

// edges[0] = -inf, edges[world] = +inf, edges[k] = upper boundary of the first of kHistBins equal bins of the samples'
// x range at which the running count reaches k / world of the finite samples.  ONE workgroup, the whole histogram in
// LDS: two 16-bit counters per word (a bin holds at most kPlanSamples < 65 536 samples).
constexpr int kPlanThreads = 1024;
// where word W of the histogram lives: a thread owns 32 consecutive words, and 64 threads reading their k-th word at
// once would all hit one LDS bank (64 waves-cycles per read, 16 waves: most of this kernel's time) -- rotated by the
// owner's number within its 32 words, they spread over the banks
__device__ __forceinline__ unsigned plan_word(unsigned W) { return (W & ~31u) | ((W + (W >> 5)) & 31u); }
__global__ void __launch_bounds__(kPlanThreads)
    k_plan_edges(const float *__restrict__ xs, const float *__restrict__ part, int world, float *__restrict__ edges) {
    __shared__ unsigned s_hist[kHistBins / 2];  // 128 KB
    __shared__ float s_lo[kPlanThreads / 64], s_hi[kPlanThreads / 64];
    __shared__ unsigned s_wave[kPlanThreads / 64];
    const unsigned t = threadIdx.x;
    for (unsigned w = t; w < kHistBins / 2; w += kPlanThreads) s_hist[w] = 0u;
    float lo = INFINITY, hi = -INFINITY;
    if (t < (unsigned) kSampleBlocks) {
        lo = part[2 * t];
        hi = part[2 * t + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((t & 63u) == 0u) {
        s_lo[t >> 6] = lo;
        s_hi[t >> 6] = hi;
    }
    __syncthreads();
    lo = s_lo[0];
    hi = s_hi[0];
    for (int w = 1; w < kSampleBlocks / 64 + (kSampleBlocks % 64 ? 1 : 0); ++w) {
        lo = fminf(lo, s_lo[w]);
        hi = fmaxf(hi, s_hi[w]);
    }
    if (!(lo <= hi)) lo = hi = 0.f;  // no finite sample at all
    const float width = fmaxf((hi - lo) / (float) kHistBins, 1e-30f);
    const float inv_w = 1.0f / width;
    for (unsigned k = t; k < kPlanSamples; k += kPlanThreads) {
        const float x = xs[k];
        if (x == x) {
            int b = (int) ((x - lo) * inv_w);
            b = min(max(b, 0), kHistBins - 1);
            atomicAdd(&s_hist[plan_word((unsigned) b >> 1)], (b & 1) ? 0x10000u : 1u);
        }
    }
    __syncthreads();
    // thread t owns bins [t kPer, (t + 1) kPer)
    constexpr int kPer = kHistBins / kPlanThreads;  // 64 bins = 32 words
    unsigned mine = 0;
#pragma unroll 8
    for (int w = 0; w < kPer / 2; ++w) {
        const unsigned v = s_hist[plan_word(t * (kPer / 2) + w)];
        mine += (v & 0xFFFFu) + (v >> 16);
    }
    unsigned incl = mine;  // inclusive scan across the workgroup: in the wave, then over the waves' totals
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = (unsigned) __shfl_up((int) incl, off);
        if ((int) (t & 63u) >= off) incl += up;
    }
    if ((t & 63u) == 63u) s_wave[t >> 6] = incl;
    __syncthreads();
    unsigned before = incl - mine, total = 0;
    for (unsigned w = 0; w < kPlanThreads / 64; ++w) {
        const unsigned v = s_wave[w];
        if (w < (t >> 6)) before += v;
        total += v;
    }
    if (t == 0) {
        edges[0] = -INFINITY;
        edges[world] = INFINITY;
        if (total == 0u)
            for (int e = 1; e < world; ++e) edges[e] = INFINITY;
    }
    if (total == 0u || mine == 0u) return;
    // the edges whose target count falls among this thread's bins (counts are small: 32-bit products do not overflow
    // below 262 144 ranks)
    unsigned run = before;
    for (int k = 0; k < kPer; ++k) {
        const unsigned v = s_hist[plan_word(t * (kPer / 2) + (k >> 1))];
        const unsigned c = (k & 1) ? (v >> 16) : (v & 0xFFFFu);
        if (c == 0u) continue;
        const unsigned prev = run;
        run += c;
        for (int e = 1; e < world; ++e) {
            const unsigned want = (unsigned) (((unsigned long long) total * (unsigned) e + (unsigned) world - 1u) / (unsigned) world);
            if (want > prev && want <= run) edges[e] = lo + (float) (t * kPer + k + 1) * width;
        }
    }
}

struct ShardPlan {
    float edges[2];      // this rank's [lo, hi)
    unsigned n_tgt_local, n_src_local, stripe_finite, pad;
};

// ---- this rank's bands of both clouds, selected in input order (a STABLE compaction: the local clouds' order, and
// with it every later tie-break and summation order, is a function of the clouds alone).  Three launches for both
// clouds together: per-block counts -> one workgroup's scan of them (+ the plan, packed for the host's one fetch) ->
// the write.  (Rounds 2-4: rocprim::select per cloud -- a memset, a look-back initialisation, the partition and a
// count kernel each, 83 us for the two at 1M points; and a separate pass over the source for its finite count.)
constexpr unsigned kBandRows = 4;
constexpr unsigned kBandPoints = kBandRows * kBlock;  // points per workgroup
struct BandArgs {
    const unsigned char *raw_t, *raw_s;
    size_t stride;
    unsigned nt, ns, blocks_t, blocks_s;
    const float *edges;
    int rank;
    float widen_t, widen_s;  // (INFINITY: everything finite in x passes)
    int all_s;               // the source unselected (second attempt: every rank takes the whole source)
    unsigned s0, s1;         // the stripe of the source whose finite points this rank counts
};

__device__ __forceinline__ bool band_test(const BandArgs &a, bool source, unsigned i, float *xyz) {
    const unsigned char *raw = source ? a.raw_s : a.raw_t;
    const float *q = reinterpret_cast<const float *>(raw + (size_t) i * a.stride);
    xyz[0] = q[0];
    xyz[1] = q[1];
    xyz[2] = q[2];
    const int r = (source && a.all_s) ? 0 : a.rank;
    const float w = source ? (a.all_s ? INFINITY : a.widen_s) : a.widen_t;
    return xyz[0] >= a.edges[r] - w && xyz[0] <= a.edges[r + 1] + w;  // (a NaN never passes)
}

__global__ void __launch_bounds__(kBlock) k_band_count(BandArgs a, unsigned *__restrict__ blk_cnt, unsigned *__restrict__ blk_fin) {
    const bool source = blockIdx.x >= a.blocks_t;
    const unsigned blk = source ? blockIdx.x - a.blocks_t : blockIdx.x;
    const unsigned n = source ? a.ns : a.nt;
    unsigned cnt = 0, fin = 0;
#pragma unroll
    for (unsigned j = 0; j < kBandRows; ++j) {
        const unsigned i = blk * kBandPoints + j * kBlock + threadIdx.x;
        float v[3];
        bool in = false;
        if (i < n) {
            in = band_test(a, source, i, v);
            if (source && i >= a.s0 && i < a.s1) fin += (v[0] - v[0] == 0.f && v[1] - v[1] == 0.f && v[2] - v[2] == 0.f) ? 1u : 0u;
        }
        cnt += in ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += (unsigned) __shfl_xor((int) cnt, off);
        fin += (unsigned) __shfl_xor((int) fin, off);
    }
    __shared__ unsigned s_c[kBlock / 64], s_f[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        s_c[threadIdx.x >> 6] = cnt;
        s_f[threadIdx.x >> 6] = fin;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned c = 0, f = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            c += s_c[w];
            f += s_f[w];
        }
        blk_cnt[blockIdx.x] = c;
        if (source) blk_fin[blk] = f;
    }
}

// exclusive scan of cnt[0, n) in place by ONE workgroup; returns the total (to every thread)
template <int THREADS>
__device__ unsigned scan_in_place(unsigned *__restrict__ cnt, unsigned n, unsigned *s_wave /* [THREADS / 64 + 1] */) {
    const unsigned t = threadIdx.x;
    const unsigned per = (n + THREADS - 1) / THREADS;
    const unsigned b0 = min(t * per, n), b1 = min(b0 + per, n);
    unsigned mine = 0;
    for (unsigned b = b0; b < b1; ++b) mine += cnt[b];
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = (unsigned) __shfl_up((int) incl, off);
        if ((int) (t & 63u) >= off) incl += up;
    }
    __syncthreads();  // (s_wave may still be read from an earlier call)
    if ((t & 63u) == 63u) s_wave[t >> 6] = incl;
    __syncthreads();
    unsigned before = incl - mine, total = 0;
    for (unsigned w = 0; w < THREADS / 64; ++w) {
        const unsigned v = s_wave[w];
        if (w < (t >> 6)) before += v;
        total += v;
    }
    unsigned run = before;
    for (unsigned b = b0; b < b1; ++b) {
        const unsigned c = cnt[b];
        cnt[b] = run;
        run += c;
    }
    return total;
}

__global__ void __launch_bounds__(1024)
    k_band_scan(unsigned *__restrict__ blk_cnt, const unsigned *__restrict__ blk_fin, unsigned blocks_t, unsigned blocks_s,
                const float *__restrict__ edges, int rank, ShardPlan *__restrict__ out) {
    __shared__ unsigned s_wave[1024 / 64 + 1];
    const unsigned n_t = scan_in_place<1024>(blk_cnt, blocks_t, s_wave);
    const unsigned n_s = scan_in_place<1024>(blk_cnt + blocks_t, blocks_s, s_wave);
    unsigned f = 0;
    for (unsigned b = threadIdx.x; b < blocks_s; b += 1024) f += blk_fin[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) f += (unsigned) __shfl_xor((int) f, off);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) s_wave[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned fin = 0;
        for (int w = 0; w < 1024 / 64; ++w) fin += s_wave[w];
        out->edges[0] = edges[rank];
        out->edges[1] = edges[rank + 1];
        out->n_tgt_local = n_t;
        out->n_src_local = n_s;
        out->stripe_finite = fin;
        out->pad = 0;
    }
}

__global__ void __launch_bounds__(kBlock)
    k_band_write(BandArgs a, const unsigned *__restrict__ blk_off, float4 *__restrict__ out_t, float4 *__restrict__ out_s) {
    const bool source = blockIdx.x >= a.blocks_t;
    const unsigned blk = source ? blockIdx.x - a.blocks_t : blockIdx.x;
    const unsigned n = source ? a.ns : a.nt;
    float4 *out = source ? out_s : out_t;
    __shared__ unsigned s_cnt[kBandRows][kBlock / 64];
    float v[kBandRows][3];
    bool in[kBandRows];
    unsigned below[kBandRows];  // passing points of this wave's row before this lane
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (unsigned j = 0; j < kBandRows; ++j) {
        const unsigned i = blk * kBandPoints + j * kBlock + threadIdx.x;
        in[j] = i < n && band_test(a, source, i, v[j]);
        const unsigned long long m = __ballot(in[j]);
        below[j] = (unsigned) __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[j][wave] = (unsigned) __popcll(m);
    }
    __syncthreads();
    unsigned base = blk_off[blockIdx.x];
#pragma unroll
    for (unsigned j = 0; j < kBandRows; ++j) {
#pragma unroll
        for (unsigned w = 0; w < kBlock / 64; ++w) {
            const unsigned c = s_cnt[j][w];
            if (w == wave && in[j]) {
                const unsigned i = blk * kBandPoints + j * kBlock + threadIdx.x;
                out[base + below[j]] = make_float4(v[j][0], v[j][1], v[j][2], __uint_as_float(i));
            }
            base += c;
        }
    }
}

}  // namespace wm

using namespace wm;

// One rank per process (or thread) under RCCL: mailboxes reach the other ranks as IPC handles, all-gathered over the
// communicator itself; a probe exchange over them and an all-reduced verdict decide, alike on every rank, whether the
// loop uses them.  Returns an error only when a COLLECTIVE failed (the communicator is unusable then).
static int setup_mailboxes_rccl(wm_comm *c) {
    struct Record {
        int ok, pad[15];
        hipIpcMemHandle_t h;
    };
    static_assert(sizeof(Record) == 128, "one 128-byte record per rank");
    const int world = c->world;
    bool ok = mailboxes_wanted() && world <= kXMaxWorld && mailbox_alloc(c) == WM_OK;
    if (world == 1) {
        if (ok) ok = mailbox_set_peers(c, {c->mail}) == WM_OK;
    }
    hipStream_t stream = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess)
        return WM_ERR_HIP;
    void *scratch = nullptr;  // [world] records | probe sums | ok word | verdict
    const size_t rec_bytes = (size_t) world * sizeof(Record);
    int rc = WM_OK;
    if (hipMalloc(&scratch, rec_bytes + 512) != hipSuccess) {
        (void) hipStreamDestroy(stream);
        return WM_ERR_NOMEM;
    }
    double *probe_out = reinterpret_cast<double *>(static_cast<char *>(scratch) + rec_bytes);
    int *probe_ok = reinterpret_cast<int *>(probe_out + kBlkLen + 2);
    int *verdict = probe_ok + 2;
    std::vector<Record> recs((size_t) world);
    if (world > 1) {
        Record mine;
        memset(&mine, 0, sizeof(mine));
        if (ok) ok = hipIpcGetMemHandle(&mine.h, c->mail) == hipSuccess;
        if (!ok) (void) hipGetLastError();
        mine.ok = ok ? 1 : 0;
        if (hipMemcpy(static_cast<Record *>(scratch) + c->rank, &mine, sizeof(mine), hipMemcpyHostToDevice) != hipSuccess)
            rc = WM_ERR_HIP;
        if (rc == WM_OK && ncclAllGather(static_cast<Record *>(scratch) + c->rank, scratch, sizeof(Record), ncclUint8, c->nccl,
                                         stream) != ncclSuccess)
            rc = WM_ERR_RCCL;
        if (rc == WM_OK && hipStreamSynchronize(stream) != hipSuccess) rc = WM_ERR_HIP;
        if (rc == WM_OK && hipMemcpy(recs.data(), scratch, rec_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = WM_ERR_HIP;
        bool all = rc == WM_OK;
        for (int r = 0; r < world && all; ++r) all = recs[(size_t) r].ok != 0;
        if (rc == WM_OK && all) {  // (every rank sees the same records: the same branch everywhere)
            std::vector<unsigned long long *> peers((size_t) world, nullptr);
            bool opened = true;
            for (int r = 0; r < world && opened; ++r) {
                if (r == c->rank) {
                    peers[(size_t) r] = c->mail;
                    continue;
                }
                void *q = nullptr;
                if (hipIpcOpenMemHandle(&q, recs[(size_t) r].h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !q) {
                    (void) hipGetLastError();
                    opened = false;
                    break;
                }
                c->ipc_opened.push_back(q);
                peers[(size_t) r] = static_cast<unsigned long long *>(q);
            }
            if (opened) opened = mailbox_set_peers(c, peers) == WM_OK;
            // the probe: a rank that could not map its peers skips it -- the others' probes then run into their
            // time limit, and the verdict below is "no" everywhere
            int good = 0;
            if (opened && hipMemset(probe_ok, 0, 16) == hipSuccess &&
                mailbox_probe_launch(c, stream, probe_out, probe_ok, 2000u) == WM_OK && hipStreamSynchronize(stream) == hipSuccess) {
                double out_h[kBlkLen];
                int ok_h = 0;
                if (hipMemcpy(out_h, probe_out, sizeof(out_h), hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(&ok_h, probe_ok, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
                    good = mailbox_probe_good(c, out_h, ok_h) ? 1 : 0;
            }
            (void) hipGetLastError();
            if (hipMemcpy(verdict, &good, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = WM_ERR_HIP;
            if (rc == WM_OK && ncclAllReduce(verdict, verdict, 1, ncclInt32, ncclMin, c->nccl, stream) != ncclSuccess)
                rc = WM_ERR_RCCL;
            if (rc == WM_OK && hipStreamSynchronize(stream) != hipSuccess) rc = WM_ERR_HIP;
            int all_good = 0;
            if (rc == WM_OK && hipMemcpy(&all_good, verdict, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = WM_ERR_HIP;
            ok = rc == WM_OK && all_good == 1;
        } else {
            ok = false;
        }
    } else if (ok) {  // a group of one: its own mailbox, checked the same way
        int good = 0;
        if (hipMemset(probe_ok, 0, 16) == hipSuccess && mailbox_probe_launch(c, stream, probe_out, probe_ok, 2000u) == WM_OK &&
            hipStreamSynchronize(stream) == hipSuccess) {
            double out_h[kBlkLen];
            int ok_h = 0;
            if (hipMemcpy(out_h, probe_out, sizeof(out_h), hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(&ok_h, probe_ok, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
                good = mailbox_probe_good(c, out_h, ok_h) ? 1 : 0;
        }
        (void) hipGetLastError();
        ok = good == 1;
    }
    (void) hipFree(scratch);
    (void) hipStreamDestroy(stream);
    if (rc != WM_OK) return rc;
    if (ok)
        c->p2p = true;
    else
        mailbox_free(c);
    return WM_OK;
}

// All ranks in this process (ncclCommInitAll, or the one-GPU stand-in): mailboxes are plain device pointers, with peer
// access enabled between the devices.  Best effort: on any failure every rank keeps the collective exchange.
static void setup_mailboxes_in_process(wm_comm **comms, int n, bool probe) {
    if (!mailboxes_wanted() || n > kXMaxWorld) return;
    bool ok = true;
    for (int r = 0; r < n && ok; ++r) ok = mailbox_alloc(comms[r]) == WM_OK;
    for (int r = 0; r < n && ok; ++r)
        for (int q = 0; q < n && ok; ++q) {
            if (comms[r]->device == comms[q]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, comms[r]->device, comms[q]->device) != hipSuccess || !can) ok = false;
            if (ok && hipSetDevice(comms[r]->device) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(comms[q]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                (void) hipGetLastError();
            }
        }
    std::vector<unsigned long long *> peers((size_t) n, nullptr);
    for (int r = 0; r < n && ok; ++r) peers[(size_t) r] = comms[r]->mail;
    for (int r = 0; r < n && ok; ++r) ok = mailbox_set_peers(comms[r], peers) == WM_OK;
    if (ok && probe) {  // one exchange, all ranks at once (one stream per device), before anything relies on it
        std::vector<hipStream_t> st((size_t) n, nullptr);
        std::vector<void *> buf((size_t) n, nullptr);
        for (int r = 0; r < n && ok; ++r) {
            ok = hipSetDevice(comms[r]->device) == hipSuccess &&
                 hipStreamCreateWithFlags(&st[(size_t) r], hipStreamNonBlocking) == hipSuccess &&
                 hipMalloc(&buf[(size_t) r], 512) == hipSuccess && hipMemset(buf[(size_t) r], 0, 512) == hipSuccess;
        }
        for (int r = 0; r < n && ok; ++r)
            ok = mailbox_probe_launch(comms[r], st[(size_t) r], static_cast<double *>(buf[(size_t) r]),
                                      reinterpret_cast<int *>(static_cast<double *>(buf[(size_t) r]) + kBlkLen + 2), 2000u) == WM_OK;
        for (int r = 0; r < n; ++r) {
            if (!st[(size_t) r]) continue;
            (void) hipSetDevice(comms[r]->device);
            if (hipStreamSynchronize(st[(size_t) r]) != hipSuccess) ok = false;
            if (ok && buf[(size_t) r]) {
                double out_h[kBlkLen + 3];
                if (hipMemcpy(out_h, buf[(size_t) r], sizeof(out_h), hipMemcpyDeviceToHost) != hipSuccess)
                    ok = false;
                else
                    ok = mailbox_probe_good(comms[r], out_h, *reinterpret_cast<const int *>(out_h + kBlkLen + 2));
            }
            (void) hipStreamDestroy(st[(size_t) r]);
            if (buf[(size_t) r]) (void) hipFree(buf[(size_t) r]);
        }
        (void) hipGetLastError();
    }
    for (int r = 0; r < n; ++r) {
        if (ok)
            comms[r]->p2p = true;
        else
            mailbox_free(comms[r]);
    }
}

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
    // mailboxes: best effort, and the SAME verdict on every rank (a rank that failed any step says so through the
    // collectives every rank runs regardless) -- ncclAllReduce stays the exchange if they cannot be had
    const int rc_mail = setup_mailboxes_rccl(c);
    if (rc_mail != WM_OK) {
        wm_comm_destroy(c);
        return rc_mail;
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
    setup_mailboxes_in_process(comms, n, true);
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
    // the stand-in's ranks share ONE GPU: their solve kernels would poll each other's mailboxes while queued behind
    // one another on the same hardware queues, so the host-side sum stays the default here; WM_COMM_P2P_LOCAL=1
    // runs the mailbox protocol anyway (tests: two ranks, two streams)
    if (const char *e = getenv("WM_COMM_P2P_LOCAL"))
        if (atoi(e) == 1) setup_mailboxes_in_process(comms, n, false);
    return WM_OK;
}

void wm_comm_destroy(wm_comm *c) {
    if (!c) return;
    mailbox_free(c);
    if (c->nccl) (void) ncclCommDestroy(c->nccl);
    if (c->local && c->rank == 0) delete c->local;
    delete c;
}

// how long one all-reduce of WM_STATS_LEN doubles takes on this communicator, back to back on the
// context's stream (us per all-reduce; collective: every rank calls it)
int wm_comm_allreduce_probe(wm_ctx *ctx, wm_comm *comm, int reps, double *us_out) {
    if (!ctx || !comm || !us_out || reps < 1) return WM_ERR_ARG;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_HIP(ctx, ctx->shard_stats.reserve(kBlkLen * sizeof(double)));
    double *blk = ctx->shard_stats.as<double>();
    WM_HIP(ctx, hipMemsetAsync(blk, 0, WM_STATS_LEN * sizeof(double), ctx->stream));
    // (with mailboxes: the exchange as the loop runs it, in a kernel of its own here)
    auto once = [&]() -> int {
        if (!comm->p2p) return wm::comm_allreduce(ctx, comm, blk, WM_STATS_LEN);
        hipLaunchKernelGGL(k_xchg_probe, dim3(1), dim3(256), 0, ctx->stream, mailbox_args(comm, comm->p2p_timeout_ms), blk,
                           (int *) nullptr);
        WM_HIP(ctx, hipGetLastError());
        return WM_OK;
    };
    for (int r = 0; r < 3; ++r) WM_TRY(once());
    WM_HIP(ctx, hipEventRecord(ctx->ev_a, ctx->stream));
    for (int r = 0; r < reps; ++r) WM_TRY(once());
    WM_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
    WM_HIP(ctx, hipEventSynchronize(ctx->ev_b));
    float ms = 0;
    (void) hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
    *us_out = (double) ms * 1e3 / reps;
    return WM_OK;
}

int wm_comm_rank(const wm_comm *c) { return c ? c->rank : -1; }
int wm_comm_world(const wm_comm *c) { return c ? c->world : 0; }
int wm_comm_mailboxes(const wm_comm *c) { return c && c->p2p ? 1 : 0; }
int wm_comm_set_exchange_timeout_ms(wm_comm *c, int ms) {
    if (!c || ms < 0) return WM_ERR_ARG;
    c->p2p_timeout_ms = (unsigned) ms;
    return WM_OK;
}

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
    WM_HIP(ctx, ctx->shard_misc.reserve((size_t) kPlanSamples * 4 + 16384));
    float *xs = ctx->shard_misc.as<float>();                                  // [kPlanSamples]
    float *edges = xs + kPlanSamples;                                         // [world + 1] (<= 501)
    float *range_part = edges + 512;                                          // [kSampleBlocks][2]
    ShardPlan *plan_dev = reinterpret_cast<ShardPlan *>(range_part + 512);
    // the sub-sample every rank plans from: kPlanSamples points at a fixed stride through the target
    const unsigned step = nt > kPlanSamples ? (nt + kPlanSamples - 1) / kPlanSamples : 1u;
    hipLaunchKernelGGL(k_xsample, dim3(kSampleBlocks), dim3(kBlock), 0, ctx->stream, raw_tgt, stride, nt, step, xs, range_part);
    hipLaunchKernelGGL(k_plan_edges, dim3(1), dim3(kPlanThreads), 0, ctx->stream, xs, range_part, world, edges);
    WM_HIP(ctx, hipGetLastError());
    WM_HIP(ctx, hipEventRecord(e_b, ctx->stream));
    // this rank's target slab + halo (a float32-safe halo: max_corr plus a hair for the rounding of x)
    // and source band (points that start within max_corr of the slab)
    BandArgs ba;
    ba.raw_t = raw_tgt;
    ba.raw_s = raw_ref;
    ba.stride = stride;
    ba.nt = nt;
    ba.ns = nr;
    ba.blocks_t = (nt + kBandPoints - 1) / kBandPoints;
    ba.blocks_s = (nr + kBandPoints - 1) / kBandPoints;
    ba.edges = edges;
    ba.rank = rank;
    ba.widen_t = (float) (p->max_corr * (1.0 + 1e-6) + 1e-4);
    ba.widen_s = (float) p->max_corr;
    ba.all_s = 0;
    // (the finite source points of this rank's 1 / world stripe: counted by the same pass)
    ba.s0 = (unsigned) ((unsigned long long) nr * (unsigned) rank / (unsigned) world);
    ba.s1 = (unsigned) ((unsigned long long) nr * (unsigned) (rank + 1) / (unsigned) world);
    const unsigned band_blocks = ba.blocks_t + ba.blocks_s;
    WM_HIP(ctx, ctx->shard_flags.reserve(((size_t) band_blocks + ba.blocks_s + 64) * sizeof(unsigned)));
    unsigned *blk_cnt = ctx->shard_flags.as<unsigned>(), *blk_fin = blk_cnt + band_blocks;
    hipEvent_t e_c = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool full_source = attempt == 1;
        ba.all_s = full_source ? 1 : 0;
        hipLaunchKernelGGL(k_band_count, dim3(band_blocks), dim3(kBlock), 0, ctx->stream, ba, blk_cnt, blk_fin);
        hipLaunchKernelGGL(k_band_scan, dim3(1), dim3(1024), 0, ctx->stream, blk_cnt, blk_fin, ba.blocks_t, ba.blocks_s, edges,
                           rank, plan_dev);
        hipLaunchKernelGGL(k_band_write, dim3(band_blocks), dim3(kBlock), 0, ctx->stream, ba, blk_cnt,
                           ctx->shard_tgt_band.as<float4>(), ctx->shard_ref_band.as<float4>());
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
        // (no synchronisation here: the loop's first launches queue up behind the index build, as in wm_icp_align)
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
    if (ctx && ctx->xchg_timed_out) {
        // ... except when what failed is the exchange through the mailboxes (a block did not arrive somewhere in the
        // group): this rank stops using its mailboxes, and SO DOES EVERY OTHER RANK -- the commit round at the end of a
        // registration's loop (k_xchg_commit, wm_icp.hip) hands every rank the same verdict, so the group's next
        // registration exchanges by ncclAllReduce on all ranks (wm_multi_icp_match retries).  Whether the peer that
        // was late is still alive this rank cannot know: a peer that has died or aborted its communicator makes that
        // collective fail or block as any RCCL collective would -- callers that see WM_ERR_RCCL here should rebuild
        // communicator and contexts if it happens again (bench.py does on the first registration).
        ctx->xchg_timed_out = false;
        if (comm) {
            comm->p2p = false;
            comm->last_error = "mailbox exchange failed in a sharded registration: the group exchanges by ncclAllReduce from now on";
        }
        return rc;
    }
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
    auto run_all = [&]() -> int {
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            rcs[(size_t) r] = wm_icp_match_sharded(m->ctx[(size_t) r], m->comm[(size_t) r], ref, n_ref, target, n_target,
                                                   stride, WM_MEM_HOST, p, res, multiscale_steps, &Ts[(size_t) r * 16],
                                                   &sts[(size_t) r]);
        });
    for (auto &t : th) t.join();
    return WM_OK;
    };
    bool had_mailboxes = false;
    for (int r = 0; r < n; ++r) had_mailboxes = had_mailboxes || (m->comm[(size_t) r] && m->comm[(size_t) r]->p2p);
    run_all();
    bool rccl_error = false;
    for (int r = 0; r < n; ++r) rccl_error = rccl_error || rcs[(size_t) r] == WM_ERR_RCCL;
    if (rccl_error && had_mailboxes) {
        // the exchange through the mailboxes failed somewhere (it has never run over xGMI in the builder's container):
        // every rank of this group goes back to the collective exchange, and the registration is run once more
        bool usable = true;
        for (int r = 0; r < n; ++r) {
            if (m->comm[(size_t) r]) m->comm[(size_t) r]->p2p = false;
            usable = usable && m->comm[(size_t) r] && (m->comm[(size_t) r]->nccl || m->comm[(size_t) r]->local);
        }
        if (usable) run_all();
    }
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
