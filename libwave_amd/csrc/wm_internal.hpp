// wm_internal.hpp -- context, device-side structs and helpers shared by the
// HIP translation units of libwavematch_hip.so (gfx950 only).
#pragma once
#include <stdio.h>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/wavematch.h"
#include "wm_math.hpp"

namespace wm {

constexpr int kMaxLevels = 6;
constexpr int kBlock = 256;
constexpr unsigned kNoIdx = 0xFFFFFFFFu;
// all-reduced block of the sharded loop: the public WM_STATS_LEN slots + [kStatsLen]: queries the certificate
// kernel searched on this rank (summed: every rank steers by the same share) + one spare
constexpr int kBlkLen = 34;
constexpr int kAcc = 18;  // ICP statistics per partial row: 17 sums + the number of source points this rank handled

// ----------------------------------------------------------- error handling
#define WM_HIP(ctx, call)                                                              \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(_e);     \
            return WM_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

#define WM_TRY(expr)                \
    do {                            \
        int _s = (expr);            \
        if (_s != WM_OK) return _s; \
    } while (0)

// ----------------------------------------------------------- device buffers
// Growable device allocation cached on the context (hipMalloc is far too slow
// to sit inside match()).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

// ------------------------------------------------------------ uniform grid
// One level of the target index: points cell-sorted (x fastest) in HBM, so the
// three x-adjacent cells of a query's neighbourhood are ONE contiguous run.
struct GridDev {
    float ox, oy, oz;  // origin (min corner)
    float h, inv_h;    // cell size
    float slack;       // cell-unit safety margin for float cell assignment
    int nx, ny, nz;
    const float4 *pts;          // cell-sorted, .w = original index bits
    const uint32_t *cell_start; // nx*ny*nz + 1
};

struct LevelsDev {  // the level ladder as the search kernel reads it (lives in HBM)
    GridDev g[kMaxLevels];
    int n;
};

struct GridLevel {
    GridDev d{};
    DevBuf pts, cell_start;
    uint64_t ncells = 0;
    bool built = false;
};

// ----------------------------------------------------- ICP state in HBM
// Lives in device memory so that a whole registration runs without the host
// in the loop; mirrored to pinned host memory when the host needs to look.
struct IcpDevState {
    double T[16];   // cumulative source->target
    float Tf[12];   // float rows 0..2 of T (what the correspondence kernel applies)
    float Tf_search[12];  // ... the last correspondence search ran under (set by the solve that consumed it): what the keys' d2 refer to
    double Tk[16];  // last incremental step
    double stats[kStatsLen];
    double mse, prev_mse;
    int iter, done, converged, state, n_corr, max_iter, forced, mode;
    int have_prev;  // keys[] hold the previous iteration's result (level prediction)
    // sharded registration: this rank handles the source points whose transformed x
    // lies in [slab_lo, slab_hi)
    int slab_on;
    float slab_lo, slab_hi;
    // every rank holds only the source points near its slab; the all-reduced count of
    // handled points must equal the cloud size in EVERY iteration, else a point fell
    // outside all bands and the registration has to be redone with full source clouds
    double expect_owned;    // > 0: the count to expect; < 0: it arrives in the all-reduced block (slot 30: the sum of
                            // the ranks' stripe_finite); 0: no check
    double stripe_finite;   // finite source points in this rank's 1 / world stripe of the (full) source cloud
    int owned_violations;
    double rot_thr, trans_thr, fit_eps;
    unsigned queue_count[kMaxLevels + 1];
    unsigned long long deferred_total;
    int svd_warm;      // warm-started SVD on (tune_fast_solve)
    double svd_v[10];  // V of the last iteration's SVD (+ a valid flag): the next one starts from it
    unsigned long long dbg[8];  // developer: cycle stamps of the last solve kernel (wm_debug_solve_cycles)
    unsigned cert_unsettled[64];  // k_nn_cert: queries it had to search in this iteration (partial counts; zeroed by the solve)
    double local_handled;  // queries THIS rank handled in the last search (sharded: before the all-reduce)
    float frac_changed;    // fraction of the handled queries whose match changed in the last search
    float frac_unsettled;  // fraction the last certificate launch had to search (0 after a full search)
    float step_disp;   // upper estimate of how far the last step moved the source points (metres)
    float src_radius;  // half diagonal of the source cloud's bounding box (for step_disp)
    float src_centre[3];
    // the count of changed matches rides in the fraction of the handled-points sum (units of 2^-24, wm_nn.hip:
    // icp_terms); beyond 2^23 queries per context only every (changed_mask + 1)-th query is counted, so
    // that the fraction can never carry into the integer part, and the count is scaled back up here
    unsigned changed_mask;
    int uns_global;  // sharded: the count of searched queries came through the all-reduced block (all ranks steer alike)
    int xchg_failed;  // sharded, in-kernel exchange (wm_xchg.hpp): a peer's block never arrived; the registration ended there
    int pad_;
};

struct Bbox {
    float lo[3], hi[3];
};

}  // namespace wm

// The opaque C handle.
struct wm_ctx {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    hipStream_t side_stream = nullptr;      // the source's Morton sort runs here, beside the target's grid build
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // wm_set_source / wm_set_target only pack the cloud and LAUNCH its bounding-box reduction; the
    // host fetches both results in one round trip when the first consumer needs them (finalize_clouds)
    bool src_pending = false, tgt_pending = false;
    unsigned src_bbox_blocks = 0, tgt_bbox_blocks = 0;
    wm::DevBuf cloud_bbox;                  // [2][kBboxBlocks][8] floats: source slot, target slot
    unsigned tuned_uses = 0;                // level-0 builds that trusted the cached cell size since the last check
    std::string last_error;

    // source (wave `ref`): Morton-ordered float4, .w = caller's index
    size_t n_src = 0, n_src_input = 0;
    wm::DevBuf src_sorted;

    // target (wave `target`)
    size_t n_tgt = 0, n_tgt_input = 0;
    wm::DevBuf tgt_orig;  // float4 in caller order, .w = index; non-finite -> NaN
    wm::Bbox tgt_bbox{};
    wm::GridLevel levels[wm::kMaxLevels];
    int n_levels = 0;
    double levels_max_corr = -1;
    float grid_cell_override = 0;
    bool trace = false;
    float tune_lane_lf = 0.2f;   // lane-serial scan: finest level with cell size >= this x radius
    float tune_coop_lf = 0.5f;   // cooperative scan: finest level with cell size >= this x radius
    int tune_lag = 2;            // iterations the host may run ahead of the device (icp_run_loop)
    float tune_r0 = 0.5f;        // first radius of an unseeded search, in level-0 cells
    float tune_r_light = 16.0f;  // lane-serial vs cooperative scan threshold, in level-0 cells (12-24 within 1 %)
    double tuned_h = 0, tuned_vol = 0;  // last auto-tuned level-0 cell size and its cloud
    size_t tuned_n = 0;
    double tuned_src_h = 0, tuned_src_vol = 0;  // the same for the source grid of the GICP covariances
    unsigned tuned_src_uses = 0;               // builds that trusted it since the last occupancy check
    size_t tuned_src_n = 0;

    // scratch
    wm::DevBuf staging, staging2, cell_of, counts, block_sums, bbox_buf;
    wm::DevBuf match_pt, match_pt_bak;  // float4 per (sorted) source point: its match's xyz
    void *small_batch = nullptr;            // wm_small.hip: staging of the batched small registrations
    void *gicp_small_batch = nullptr;       // wm_gicp_small.hip: ... of the batched small GICP registrations
    void *ndt_small_batch = nullptr;        // wm_ndt_small.hip: ... of the batched small NDT registrations
    void *batch_voxel = nullptr;            // wm_batch.hip: buffers of the batched voxel filter
    wm::DevBuf phase_log;                   // developer: per-iteration phase cycle sums of the search kernel
    wm::DevBuf cost_log;                    // developer: per-query search cost of every iteration (wm_debug_cost_log)
    int cost_log_iter = 0, cost_log_cap = 0;
    wm::DevBuf keys, partials, partials2, corr_tmp_idx, corr_tmp_d2, d_levels;
    // the iteration's sums as exact integer limbs (wm_bins.hpp): all zero between iterations (the solve puts the
    // zeros back)
    wm::DevBuf bins;
    bool bins_dirty = false;     // an iteration loop is running or ended abnormally: the bins may hold sums
    int tune_grid_variant = 0;   // developer: k_nn_grid<SVD, balanced> at other register budgets (wm_nn.hip: launch_nn_grid)
    int tune_bins = 1;           // 0: rows of partial sums + k_reduce_rows + k_reduce_solve, as up to round 5
    wm::DevBuf nn_bound;                    // float4 per (sorted) source point, written by k_nn_cert's searches: where the query was (xyz) and a lower bound (w) on its distance, there, to every target point but its match
    wm::DevBuf cert_count;                  // developer: unsettled queries per launch of k_nn_cert ([launch][64] partial counts)
    wm::DevBuf cert_prof;                   // developer: phase cycle sums per launch of k_nn_cert ([launch][8][8])
    int cert_log_iter = 0, cert_log_cap = 0;
    int cert_launches = 0;                  // of the last align
    // the resident late-iteration kernel (k_nn_cert<.., LATE>)
    int late_capacity = 0;                  // workgroups of it the device holds at once (0: not asked yet, -1: unusable)
    wm::DevBuf late_ctl;                    // LateCtl
    unsigned long long *h_late = nullptr;   // pinned: its exit word
    unsigned late_seq = 0;
    int late_iters = 0, late_launches = 0;  // of the last align: iterations that ran inside it
    float late_ms = 0.f;                    // ... and its event-timed duration (profile >= 1)
    // 1: use it.  OFF by default: measured at 1M points (profiles/r04_experiments.md) an iteration inside costs
    // 33-43 us against 32-37 us for a launched certificate iteration + its solve kernel -- the workers' certificate
    // phase is bound by the vector ALU (~8 us chip-wide for the f64 sums of a million queries), the slowest
    // workgroup's searches end 10 us after the median one's, and the solver's chain (rows 3.8, solve 4.4, hand-out
    // 0.7 us) is serial behind them
    int tune_late = 0;
    int tune_early_source = 2;              // a host target's upload overlaps the source's sort (wm_set_target); 2: and, from
                                            // pinned memory, starts on a copy engine before that sort is enqueued
    int tune_cov_dbg = 0;                   // developer timing experiment in k_gicp_cov (wrong results): see there
    unsigned long long *h_pub = nullptr;    // pinned: [0] (done << 63 | iterations finished << 32 | step size bits) of the latest solve, [k] iteration k's own record
    int h_pub_slots = 0;
    wm::DevBuf vg_idx, vg_idx2, vg_perm, vg_perm2, vg_tmp, vg_seg, io_a, io_b, ds_ref, ds_tgt, match_ref, match_tgt;
    wm::DevBuf d_state;
    wm::IcpDevState *h_state = nullptr;  // pinned
    void *h_scratch = nullptr;           // pinned, device-visible scratch (reduction partials, level table)
    size_t h_scratch_bytes = 0;
    unsigned *h_sig = nullptr;           // pinned: completion flag polled by fast_fetch
    unsigned sig_seq = 0;
    // served GICP evaluations (wm_gicp.hip): the mailbox in device memory the host writes through the BAR
    wm::DevBuf gicp_mailbox;
    unsigned gicp_serve_seq = 0;
    unsigned gicp_serve_abandoned = 0;      // rounds after which the evaluator's `abandoned` word was found set (diagnostic)
    int gicp_serve_ok = 0, gicp_serve_capacity = 0, gicp_serve_cached = 0;  // 0: not tried yet, 1: usable, -1: not on this system
    int tune_gicp_served = 1;             // 0: a kernel launch per evaluation; 1: resident evaluator; 2: ... without the on-chip copy of the pairs
    int gicp_serve_test_stall_ms = 0;     // test hook: the host sleeps this long before its third served evaluation
    void *h_gicp_slots = nullptr;         // pinned: the evaluator's answers, thirteen (sum, command number) pairs
    double *h_gicp = nullptr;            // pinned, device-visible: the GICP objective's partial sums land here
    int ndt_rank = 0, ndt_world = 1;     // wm_ndt_set_shard: this context's slice of the source
    int (*ndt_reduce)(double *, int, void *) = nullptr;
    void *ndt_reduce_user = nullptr;
    double *h_ndt = nullptr;             // pinned: the NDT derivative passes' block partials
    bool ndt_profile = false;            // HIP events around every derivative pass (kernel_ms)
    bool gicp_profile = false;           // HIP events around every objective evaluation (fdf_kernel_ms)
    bool have_corr = false, last_align_valid = false, last_align_converged = false;
    bool last_align_sharded = false;     // the last align was this rank's part of a sharded registration (wm_icp_info_sharded)
    wm::DevBuf keys_bak;
    double corr_T[16];

    // carried PCL object state
    double prev_mse = -1;

    // events for profile mode
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipEvent_t ev_block = nullptr;  // the event sync_sleeping polls between short sleeps
    std::vector<float> iter_nn_ms;

    // GICP: caller-order source, its own search grid, per-point covariances
    wm::DevBuf src_orig, gicp_c1, gicp_c2, gicp_mahal;
    wm::Bbox src_bbox{};
    wm::GridLevel src_grid;
    bool gicp_cov_src_valid = false, gicp_cov_tgt_valid = false;
    int gicp_cov_k = 0;
    double gicp_cov_eps = 0;

    // NDT voxel model of the target
    wm::DevBuf ndt_keys, ndt_keys2, ndt_vox, ndt_vkey, ndt_hkeys, ndt_hvals, ndt_dense, ndt_meanf, ndt_vsum;
    bool ndt_dense_on = false;  // dense cell -> voxel-slot table built (small lattices)
    int ndt_dense_lo[3] = {0, 0, 0}, ndt_dense_dim[3] = {0, 0, 0};
    int tune_ndt_keys64 = 0;      // developer: 64-bit voxel sort keys whatever the lattice's size (WM_TUNE_NDT_KEYS64)
    int tune_ndt_vox_split = -1;  // developer: points per voxel up to which a LANE forms a voxel's sums (-1: ndt_build's choice)
    int tune_ndt_dense = 2;  // 0: hash grid; 1: dense cell -> slot table; 2: + the float4 cell lattice (wm_ndt.hip)
    bool ndt_cells4_on = false;
    wm::DevBuf ndt_cells4;
    float tune_knn_r0 = 0.f;     // first radius of the k-NN (covariance) scan in cells; 0 = by k (1.0 up to k = 12, else 1.5)
    int tune_radix_min = 256 << 10;  // sorts of more items take the radix path (wm_sort.hpp) ...
    int tune_pack_bbox = 1;          // a cloud's bounding box is formed by the launch that packs it (WM_TUNE_PACK_BBOX)
    int tune_sort = 1;               // ... 1: the library's own three-launches-per-pass sort, 0: rocPRIM's onesweep (WM_TUNE_SORT)
    int tune_xcd_chunk = 32;     // search kernel: XCDs take turns in chunks of this many workgroups (0: one eighth each)
    int tune_scan = 1;           // exclusive scans: rocPRIM look-back scan (1) or the three-kernel scan (0)
    int tune_nn_walk_filter = 1;  // balanced walk: LDS atomic only for trips that can improve the owner's best
    int tune_xcd_reverse = 0;    // search kernel: hand the workgroups out back to front (experiment)
    int tune_force_shard = 0;    // WM_SHARD_FORCE=1: a one-rank RCCL group still runs the sharded loop (plumbing check)
    int tune_two_streams = 1;    // source Morton sort on a side stream beside the target's grid build
    int tune_fuse_stats = 1;     // ICP statistics summed in the tail of the search kernel (0: separate k_icp_stats pass)
    int tune_nn_balanced = 1;    // search kernel: wave-pooled candidate trips (0: every lane walks its own)
    int tune_cert_from = -1;     // k_nn_cert from this iteration of an align on (-1: chosen from the step size, tune_cert_disp; -2: never)
    float tune_cert_disp = 0.15f;  // ... once a step moves the points by less than this many level-0 cells
    float tune_cert_changed = 0.05f;   // ... AND fewer than this fraction of the matches changed in the last full search
    float tune_cert_unsettled = 0.40f; // back to full searches when a certificate launch had to search more than this fraction
    float tune_cert_pad_mul = 8.f, tune_cert_pad_frac = 0.5f;  // runner-up room of a certified search (see k_nn_cert)
    int tune_cert_nb = 4;        // batches of 64 queries per workgroup of k_nn_cert (2, 4 or 8)
    int tune_cert_rc = 3;        // rows per step of its searches (3 or 6)
    int tune_cert_dbg_skip = 0;  // developer timing experiment (wrong results): see k_nn_cert
    int tune_nn_early_loads = 1; // k_nn_grid: the three stream loads issued before the state is looked at
    int tune_nn_nt_stores = 1;   // search kernels: non-temporal result stores (nothing left dirty in L2 at the kernel boundary)
    int tune_fast_solve = 1;     // experiment knob for the solve kernel
    int tune_spin_us = 80;       // wait_flag: busy-poll this long before polling with yields
    int tune_ndt_spec_hessian = 1;  // form the Hessian along with the first extra line-search trial (wm_ndt.hip step_length_mt)
    int tune_ndt_blocks = 0;  // workgroups (= partial rows) of one NDT derivative pass; 0: one resident round (wm_ndt.hip)
    int ndt_cus = 0;
    int tune_ndt_fused_fetch = 1;  // a pass's last workgroup adds the rows and hands the sums to the host (wm_ndt.hip)
    wm::DevBuf ndt_ticket;
    wm::DevBuf ndt_perm, ndt_perm2, ndt_flags, ndt_seg, ndt_tmp;  // ndt_build's scratch (its own: see there)
    bool xchg_timed_out = false;  // the last sharded loop ended because a peer's block did not arrive (wm_shard.hip)
    bool sort_join_pending = false;  // the source's Morton sort runs on the side stream, ev_join recorded, nobody waits yet
    bool sort_deferred = false;      // ... is still to be enqueued there (finalize_clouds mode 2: ev_fork recorded, n_src = the finite points' count)
    unsigned ndt_seq = 0;
    int tune_gicp_blocks = 256;  // workgroups (= partial rows) of one GICP objective evaluation (double-double sums: 512 / 256 / 128 / 64 -> 6.9 / 6.4 / 7.3 / 9.6 ms per 500k registration)
    bool ndt_built = false;
    int ndt_model_builds = 0;    // voxel models built so far (wm_ndt_stats.model_builds)
    double ndt_res = -1;
    unsigned ndt_nvox = 0, ndt_nvalid = 0, ndt_hmask = 0;

    // sharded (multi-GPU) stepping
    wm::DevBuf shard_ref, shard_tgt, shard_ref_band, shard_tgt_band, shard_misc, shard_flags, shard_flags2, shard_pos_t,
        shard_pos_s, shard_stats, ndt_sum_dev;
    float shard_lo = 0, shard_hi = 0;
    struct wm_comm *ndt_comm = nullptr;    // wm_ndt_set_comm: the derivative passes' sums are all-reduced on the device
    bool shard_active = false;
    wm_icp_params shard_params{};
    float shard_thr = 0;
    bool shard_brute = false;
};

namespace wm {

// ---- wm_grid.hip
// (bbox_partials / bbox_blocks: the packed cloud's bounding-box partials as launch_bbox leaves them, in the same launch)
int pack_cloud(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem, float4 *out, int slot = 0, bool staged = false,
               float *bbox_partials = nullptr, unsigned *bbox_blocks = nullptr);
bool upload_begin_async(wm_ctx *ctx, const void *pts, size_t bytes);  // (pinned host memory: the copy starts now, see wm_grid.hip)
int compute_bbox(wm_ctx *ctx, const float4 *pts, size_t n, Bbox *out, size_t *n_valid);
// the two halves of compute_bbox: enqueue the reduction into `partials_dev` (kBboxBlocks * 8 floats),
// and finish it on the host from the fetched partials
constexpr int kBboxBlocks = 512;
int launch_bbox(wm_ctx *ctx, const float4 *pts, size_t n, float *partials_dev, unsigned *blocks_out);
void finish_bbox(const float *partials_host, unsigned blocks, Bbox *out, size_t *n_valid);
// Fetch what wm_set_source / wm_set_target left pending (bounding boxes, finite-point counts), then
// Morton-sort the source on the side stream while -- when max_corr > 0 and the search will use the
// grid -- the target's level ladder is built on the main stream; both are joined before returning.
// sort_aside: the source's Morton sort goes to the side stream and the call does NOT wait for it (ctx->sort_join_pending;
// join_source_sort makes the context's stream wait) -- for a caller with work of its own to enqueue meanwhile
// sort_aside = 2: ... and the sort is not even ENQUEUED yet (ctx->sort_deferred): the caller enqueues what is on its own
// critical path first and calls enqueue_deferred_sort when it comes to its first wait for the device (wm_ndt_align)
int finalize_clouds(wm_ctx *ctx, double max_corr = -1.0, int nn_method = 0, int sort_aside = 0);
int join_source_sort(wm_ctx *ctx);
int enqueue_deferred_sort(wm_ctx *ctx);
int build_grid_level(wm_ctx *ctx, const float4 *pts, size_t n, const Bbox &bb, float h,
                     GridLevel *lvl, double *avg_occupancy);
int morton_sort(wm_ctx *ctx, const float4 *pts, size_t n, const Bbox &bb, size_t n_valid,
                float4 *out);
int ensure_levels(wm_ctx *ctx, double max_corr);
// fetch a small result from device memory into pinned host memory and wait for it (copy, fence
// and completion flag by one wavefront; see k_fetch_signal)
int fast_fetch(wm_ctx *ctx, void *dst_pinned, const void *src_dev, size_t bytes);
// column sums of a [rows][k] f64 block (k <= 32), reduced on the device, k doubles delivered
int fast_fetch_sum(wm_ctx *ctx, double *dst_pinned, const double *src_dev, unsigned rows, unsigned k);
// a caller-supplied producer kernel that ends with fast_fetch's fence + flag protocol (it gets the
// flag's address and the sequence number to write); waits for it
int fast_fetch_begin(wm_ctx *ctx, unsigned **flag, unsigned *seq);
int fast_fetch_wait(wm_ctx *ctx, unsigned seq);
// results that a kernel hands over as 16-byte slots {double value, unsigned number, 0} in pinned memory, one store each
// (no flag behind the data, hence no system-scope fence): wait until all n slots carry `seq`
int wait_slots(wm_ctx *ctx, const double *slots, int n, unsigned seq);
template <class Launch>
inline int fast_fetch_custom(wm_ctx *ctx, Launch launch) {
    unsigned *flag = nullptr, seq = 0;
    WM_TRY(fast_fetch_begin(ctx, &flag, &seq));
    launch(flag, seq);
    WM_HIP(ctx, hipGetLastError());
    return fast_fetch_wait(ctx, seq);
}
// the same column sums left in DEVICE memory (no signal): what an all-reduce then works on
int sum_to_device(wm_ctx *ctx, double *dst_dev, const double *src_dev, unsigned rows, unsigned k);
// ---- wm_shard.hip: sum `n` doubles in device memory over the ranks of `comm`, on the context's stream
int comm_allreduce(wm_ctx *ctx, struct wm_comm *comm, double *dev, int n);
struct XchgDev;
// WM_OK (and *out filled) when the communicator has mailboxes for the in-kernel exchange of the sharded loop's block
int comm_exchange_args(struct wm_comm *comm, XchgDev *out);
// developer tracing (env WM_TRACE=1): drain the stream and print a marker, so that a GPU fault can
// be pinned to the stage that was running
#define WM_TRACE(ctx, what)                                                     \
    do {                                                                        \
        if ((ctx)->trace) {                                                     \
            (void) hipStreamSynchronize((ctx)->stream);                         \
            fprintf(stderr, "[wm] %s\n", what);                                 \
            fflush(stderr);                                                     \
        }                                                                       \
    } while (0)
// pinned, device-visible host scratch of at least `bytes` (kernels write small results into it
// through fast_fetch; the host reads them once its flag has arrived)
void *pinned_scratch(wm_ctx *ctx, size_t bytes);
// device -> caller (pageable) memory: drain the stream, then a blocking copy (the caller's pages
// are pinned and unpinned by the runtime inside that one call)
int copy_to_caller(wm_ctx *ctx, void *dst, const void *src_dev, size_t bytes);
int exclusive_scan(wm_ctx *ctx, const unsigned *in, size_t n, unsigned *out);

// ---- wm_voxel.hip
// pcl::VoxelGrid on device: `in` is a packed float4 cloud (w = index, NaN = invalid);
// writes centroids (float4, w = output index) to `out` (capacity >= n), count to *n_out
// `known`: the cloud's bounding box and finite-point count if the caller already has them (the
// multiscale match filters the same cloud at four leaf sizes); nullptr = computed here
struct VgKnown {
    Bbox bb;
    size_t valid;
};
int voxel_downsample_dev(wm_ctx *ctx, const float4 *in, size_t n, float leaf, float4 *out,
                         size_t *n_out, const VgKnown *known = nullptr);
int transform_cloud_dev(wm_ctx *ctx, const float4 *in, size_t n, const double T[16], float4 *out);

// ---- wm_nn.hip
// stats_mode < 0: search only; WM_ICP_SVD / WM_ICP_GN6: the search kernel also reduces the ICP
// statistics of the iteration to *rows_out rows of kAcc doubles in ctx->partials
int launch_nn_grid(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2,
                   int stats_mode = -1, unsigned *rows_out = nullptr, bool use_bins = false);
int bins_ready(wm_ctx *ctx);  // wm_icp.hip: the iteration's bins (wm_bins.hpp) allocated and all zero
int launch_nn_brute(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1);
// the resident form of the certificate kernel (wm_nn.hip: k_nn_cert<.., LATE>): the late iterations in one launch
bool late_possible(wm_ctx *ctx, int stats_mode, unsigned *blocks_out);
size_t late_ctl_bytes();  // sizeof(LateCtl): the developer stamps sit behind it
int launch_nn_late(wm_ctx *ctx, float thr_d2, int stats_mode, unsigned blocks, bool bounds_valid, unsigned exit_seq,
                   float stop_unsettled, float stop_disp, int max_inside);
// the device's budget of resident workgroups (per process), in 1/1024ths of the device: the share taken (0: refused)
int resident_admit(int device, int nb, int capacity);
void resident_release(int device, int share);
// one turn of a host-side busy-wait (the pause hint of the host's architecture; nothing where there is none)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

// Wait for everything enqueued on the context's stream WITHOUT burning a core: the batched paths wait
// milliseconds per launch, and a crew of MultiMatcher workers that all spin through their waits
// (hipStreamSynchronize busy-polls) exhausts a container's CPU quota -- the whole process is then
// throttled, staging threads included.  The thread sleeps between looks at an event (wake-up ~0.1 ms).
int sync_sleeping(wm_ctx *ctx);
int launch_fix_keys(wm_ctx *ctx, float thr_d2);  // after certified iterations: every key's distance brought up to date
// the certificate kernel (late iterations): stats_mode as above; bounds_valid = the previous search of
// this align was launch_nn_cert too (its per-query bounds are still in ctx->nn_bound)
int launch_nn_cert(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, int stats_mode,
                   unsigned *rows_out, bool bounds_valid, bool use_bins = false);
// pinned host staging of the batched paths (wm_small.hip, wm_gicp_small.hip, wm_ndt_small.hip): grows, never shrinks
inline int pinned_reserve(wm_ctx *ctx, void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return WM_OK;
    if (*p) (void) hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    WM_HIP(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return WM_OK;
}
inline size_t align_up256(size_t v) { return (v + 255) & ~(size_t) 255; }
void small_batch_release(wm_ctx *ctx);
void gicp_small_release(wm_ctx *ctx);
void ndt_small_release(wm_ctx *ctx);   // wm_ndt_small.hip: ... of the batched small NDT registrations  // wm_gicp_small.hip: the staging of the batched small GICP registrations
// wm_batch.hip: pcl::VoxelGrid of all the clouds of a batch in one pass (see there)
int batch_voxel_filter(wm_ctx *ctx, const wm_batch_item *items, const std::vector<int> &idx, size_t stride, int mem, float leaf,
                       const float4 **filtered, std::vector<unsigned> &off, std::vector<unsigned> &n_out);
// ---- wm_small.hip / wm_batch.hip: whole registrations inside one workgroup, many per launch
struct SmallJob {
    const void *src;
    size_t n_src;
    const void *tgt;
    size_t n_tgt;
    double prev_mse0;
    int presorted;  // the source is already in a spatial order (skips the in-kernel sort)
};
struct SmallResult {
    double T[16];
    double info[36];
    double mse, prev_mse;
    int iterations, converged, state, n_corr, info_degenerate;
    float cell;
    unsigned long long cyc[4];
};
int small_run(wm_ctx *ctx, const SmallJob *jobs, int n, size_t stride, int mem, const wm_icp_params *p, int with_info,
              double info_max_corr, SmallResult *res, float *kernel_ms);
void small_fill_stats(const SmallResult &r, float kernel_ms, wm_icp_stats *s);
void batch_voxel_release(wm_ctx *ctx);
int batch_match_scaled(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride, int mem,
                       const wm_icp_params *p, float res, int multiscale_steps, int with_info, double *T_out,
                       double *info_out, wm_icp_stats *stats, int *status);
float threshold_d2(double max_corr);
float threshold_d2_strict(double max_corr);

// ---- wm_icp.hip
int shard_begin(wm_ctx *ctx, const wm_icp_params *p, double x_lo, double x_hi, double expect, double stripe_finite,
                bool *brute_out, float *thr_out, double prev_mse0);
// the iteration loop of one registration (state already uploaded); blk != nullptr: sharded (see wm_icp.hip)
int icp_run_loop(wm_ctx *ctx, const wm_icp_params *p, bool brute, float thr, struct wm_comm *comm, double *blk,
                 double T_out[16], wm_icp_stats *stats);
// one correspondence pass with transform T; `predict` lets the search start from the
// radii in the current keys
int nn_pass(wm_ctx *ctx, const double T[16], float thr_d2, double max_corr, bool predict, bool slab = false,
            float slab_lo = 0.f, float slab_hi = 0.f, bool wait = true);

}  // namespace wm
