/*
 * wavematch.h -- C ABI of libwavematch_hip.so: the MI355X (gfx950) registration
 * back end behind libwave's wave::Matcher<PCLPointCloudPtr> API.
 *
 * The reference (wavelab/libwave) has no FFI boundary on this path: its
 * ICPMatcher / GICPMatcher / NDTMatcher call PCL C++ templates directly.  Each
 * entry point below names the reference call (file:line under /root/reference)
 * whose work it replaces; the C++ shim in include/wave/matching maps the
 * reference's classes onto these calls.
 *
 * Conventions
 *   - every function returns an int status: 0 = WM_OK, >0 = algorithmic
 *     "no result" (the reference's match() == false), <0 = error.  No C++
 *     exceptions cross this boundary.
 *   - clouds are arrays of points with a caller-given byte stride whose first
 *     12 bytes are float x, y, z (pcl::PointXYZ is stride 16; packed XYZ is 12).
 *     `mem` says where the array lives: WM_MEM_HOST (copied H2D by the call) or
 *     WM_MEM_DEVICE (an HBM pointer, e.g. a torch tensor's data_ptr()).
 *     The library never retains a caller pointer past the call.
 *   - 4x4 transforms and 6x6 matrices are row-major doubles.
 *   - a wm_ctx is thread-compatible (one thread at a time); distinct contexts
 *     are fully concurrent (own HIP stream).  One ctx == one reference matcher
 *     object (it carries PCL's per-object state, e.g. the convergence
 *     criteria's previous MSE).
 */
#ifndef WAVEMATCH_H
#define WAVEMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_OK 0
#define WM_NOT_CONVERGED 1            /* pcl hasConverged() == false */
#define WM_TOO_FEW_CORRESPONDENCES 2  /* PCL: "Not enough correspondences found" */
#define WM_ERR_ARG (-1)
#define WM_ERR_HIP (-2)
#define WM_ERR_RCCL (-3)
#define WM_ERR_STATE (-4) /* e.g. align before set_source/set_target */
#define WM_ERR_NOMEM (-5)

enum { WM_MEM_HOST = 0, WM_MEM_DEVICE = 1 };

typedef struct wm_ctx wm_ctx;

/* ------------------------------------------------------------- lifecycle */
/* Replaces the PCL member objects a matcher constructs (icp.hpp:100-102,
 * gicp.hpp:61-62, ndt.hpp:72).  `device` is the HIP device ordinal. */
int wm_ctx_create(wm_ctx **out, int device);
void wm_ctx_destroy(wm_ctx *ctx);
const char *wm_strerror(int status);
/* last HIP/RCCL error text recorded on this ctx ("" if none) */
const char *wm_last_error(const wm_ctx *ctx);
/* library / build identification, e.g. "wavematch-hip 0.1 gfx950" */
const char *wm_version(void);

/* external != 0: adopt the caller's HIP stream (e.g. torch's current stream, so that
 * collectives issued by the caller order correctly against this context's kernels);
 * a NULL handle then means the default stream.  external == 0: back to the context's
 * own stream.  The external stream is not owned. */
int wm_ctx_set_stream(wm_ctx *ctx, void *hip_stream, int external);

/* ---------------------------------------------------------------- clouds */
/* wave `ref` == PCL source (the cloud that is moved / queried):
 *   icp.setInputSource  wave_matching/src/icp.cpp:87,110,125
 *   gicp.setInputSource wave_matching/src/gicp.cpp:44
 *   ndt.setInputSource  wave_matching/src/ndt.cpp:50
 * Uploads, drops non-finite points, and Morton-orders the cloud in HBM. */
int wm_set_source(wm_ctx *ctx, const void *pts, size_t n, size_t stride_bytes, int mem);
/* wave `target` == PCL target (the cloud that is indexed):
 *   icp.setInputTarget  wave_matching/src/icp.cpp:91,114,124
 *   gicp.setInputTarget wave_matching/src/gicp.cpp:54
 *   ndt.setInputTarget  wave_matching/src/ndt.cpp:55
 * Replaces the FLANN kd-tree build in pcl::Registration::initCompute with a
 * cell-sorted uniform grid in HBM.  grid_cell = 0 picks the cell size from the
 * measured point density. */
int wm_set_target(wm_ctx *ctx, const void *pts, size_t n, size_t stride_bytes, int mem);
int wm_set_grid_cell(wm_ctx *ctx, float grid_cell);
/* number of points currently held (after dropping non-finite ones) */
int wm_cloud_sizes(const wm_ctx *ctx, size_t *n_source, size_t *n_target);

/* ------------------------------------------------------------------- ICP */
enum { WM_ICP_SVD = 0, WM_ICP_GN6 = 1 };
enum { WM_NN_AUTO = 0, WM_NN_GRID = 1, WM_NN_BRUTE = 2,
       /* wm_nn_search only, OR-ed in: seed every query with the point it matched in the previous
        * search of the same clouds (what consecutive ICP iterations do); same result, less work */
       WM_NN_WARM = 0x100 };
enum { /* pcl::registration::DefaultConvergenceCriteria::ConvergenceState */
       WM_CONV_NOT_CONVERGED = 0,
       WM_CONV_ITERATIONS = 1,
       WM_CONV_TRANSFORM = 2,
       WM_CONV_ABS_MSE = 3,
       WM_CONV_REL_MSE = 4,
       WM_CONV_NO_CORRESPONDENCES = 5,
       WM_CONV_FORCED = 6 };

typedef struct {
    double max_corr;      /* ICPMatcherParams::max_corr, icp.hpp:35 -> icp.cpp:47 */
    int max_iter;         /* ICPMatcherParams::max_iter, icp.hpp:37 -> icp.cpp:48 */
    double t_eps;         /* ICPMatcherParams::t_eps,    icp.hpp:41 -> icp.cpp:49 */
    double fit_eps;       /* ICPMatcherParams::fit_eps,  icp.hpp:43 -> icp.cpp:50 */
    int force_iterations; /* >0: run exactly this many iterations (bench; no stop tests) */
    int mode;             /* WM_ICP_SVD: PCL's Umeyama step (parity default);
                             WM_ICP_GN6: 6x6 J^T J / J^T r Gauss-Newton step */
    int nn_method;        /* WM_NN_AUTO | WM_NN_GRID | WM_NN_BRUTE */
    int carry_state;      /* 1: seed the criteria's previous MSE from the ctx (PCL keeps
                             it across align() calls on one object); 0: fresh */
    int profile;          /* HIP-event timing on the ctx stream: 1 = the level-0
                             correspondence kernel only; 2 = every kernel class */
    int reserved;
} wm_icp_params;

typedef struct {
    int converged;  /* hasConverged() */
    int iterations; /* nr_iterations_ */
    int state;      /* WM_CONV_* */
    int n_corr;     /* correspondences of the last iteration */
    double mse;     /* mean d^2 of the last iteration's correspondences */
    double prev_mse;
    /* measurement (HIP events on the ctx stream; ms) */
    float align_ms;      /* whole wm_icp_align call, device side */
    float nn_ms;         /* sum over level-0 correspondence-kernel launches (profile=1) */
    float coarse_ms;     /* sum over the coarser-level launches for deferred queries */
    float stats_ms;      /* sum over statistic-reduction launches (profile=1) */
    float solve_ms;      /* sum over reduce+solve launches (profile=1) */
    int nn_launches;     /* level-0 correspondence launches timed in nn_ms */
    int nn_levels;       /* grid levels used */
    uint64_t deferred;   /* queries that needed a coarser grid level (profile=1) */
    float grid_cell;     /* level-0 cell size used */
    int owned_violations; /* sharded path: iterations in which the ranks together did not
                             handle every source point (see wm_icp_shard_begin) */
    int cert_launches;   /* iterations whose correspondences came from the certificate kernel
                            (k_nn_cert: previous match proved still nearest, search only where the
                            proof fails) instead of a full search; same correspondences either way */
    float nn_cert_ms;    /* the part of nn_ms spent in those launches (profile >= 1) */
    /* sharded registrations (wm_icp_align_sharded): where this rank's time went, so that a multi-GPU
       run explains itself.  Device times from HIP events on the context's stream, ms. */
    float plan_ms;       /* x range + 64 Ki-bin histogram of a fixed sub-sample of the target, slab edges */
    float compact_ms;    /* this rank's target slab + halo and source band selected out of the full clouds */
    float index_ms;      /* host wall time: packing, bounding boxes, source order, grid ladder of the LOCAL clouds */
    float iter_ms;       /* host wall time of the iteration loop (search, sums, all-reduce, solve) */
    float allreduce_ms;  /* sum over the iterations' all-reduces of WM_STATS_LEN doubles (profile >= 1) */
    unsigned n_tgt_local, n_src_local; /* points this rank indexed / searched */
    int rccl_ranks;      /* ncclCommCount of the communicator (0: none or the in-process stand-in) */
    int shard_attempts;  /* 2 if the registration had to be redone with full source clouds */
    /* the resident late-iteration kernel (k_nn_cert<.., LATE>: certificate, searches, sums, solve and
       stopping rules of the late iterations in ONE launch; counted in cert_launches too) */
    int late_iterations; /* iterations that ran inside it */
    int late_launches;   /* its launches (1, unless its own policy sent it back to full searches in between) */
    float late_ms;       /* their duration by HIP events (profile >= 1): search + sums + solve of those iterations */
    int exchange_in_kernel; /* sharded: 1 if the iterations' blocks were exchanged through the ranks' mailboxes inside
                               the solve kernel (one launch per iteration's tail), 0 if by ncclAllReduce between two */
} wm_icp_stats;

void wm_icp_default_params(wm_icp_params *p);

/* pcl::IterativeClosestPoint::align + hasConverged + getFinalTransformation
 * (wave_matching/src/icp.cpp:95-101,116-119,126-129).  T_out maps source->target.
 * Returns WM_OK when converged, WM_NOT_CONVERGED / WM_TOO_FEW_CORRESPONDENCES
 * otherwise (T_out then untouched, as ICPMatcher leaves `result`). */
int wm_icp_align(wm_ctx *ctx, const wm_icp_params *p, double T_out[16], wm_icp_stats *stats);

/* ICPMatcher::match() in one call (wave_matching/src/icp.cpp:75-133): with both
 * caller clouds as inputs, runs the reference's three branches entirely on device
 *   res > 0 && multiscale_steps > 0 : for i = steps..0: leaf = 2^i * res, VoxelGrid
 *        both clouds, pre-transform the filtered ref by the running transform,
 *        max_corr = 2^i * p->max_corr, align, running = T_i * running  (icp.cpp:77-104)
 *   res > 0 && multiscale_steps == 0: VoxelGrid both clouds, align    (icp.cpp:105-122)
 *   res <= 0                         : align on the full clouds       (icp.cpp:123-131)
 * Returns WM_OK and writes T_out only when every align converged (as match() does). */
int wm_icp_match(wm_ctx *ctx, const void *ref, size_t n_ref, const void *target, size_t n_target,
                 size_t stride_bytes, int mem, const wm_icp_params *p, float res,
                 int multiscale_steps, double T_out[16], wm_icp_stats *stats);

/* Many registrations per launch -- the throughput path under wave::MultiMatcher
 * (wave_matching/include/wave/matching/multi_matcher.hpp:29-96; worker loop impl/
 * multi_matcher_impl.hpp:45-53: setRef, setTarget, match, estimateInfo per queued pair).  Every item
 * is ICPMatcher::match() as wm_icp_match runs it (res, multiscale_steps: the three branches of
 * icp.cpp:75-133), with stopping criteria that start fresh, and, with with_info = 1, the estimator
 * whose result estimateInfo() always ends with (icp.cpp:135-142 -> estimateLUMold,
 * icp_pcl_functions.cpp:51-179, on the clouds of the last align).  One workgroup per item runs a
 * whole align: a target of up to WM_BATCH_LDS_TARGET_POINTS points lives, cell-sorted, in its compute
 * unit's LDS; larger ones, up to WM_BATCH_MAX_TARGET_POINTS (16-bit slots and indices), in HBM scratch
 * that the L2 / Infinity Cache keep close (same code, ~3x slower per point).
 *   res <= 0: the clouds as given; a target beyond WM_BATCH_MAX_TARGET_POINTS is WM_ERR_ARG
 *             (register such pairs one by one).
 *   res > 0 : both clouds of every item go through pcl::VoxelGrid first -- all clouds of the batch in
 *             one pass of device-wide kernels, the cloud number above the leaf index in one sort key --
 *             per scale (leaf = 2^i res, i = multiscale_steps .. 0; icp.cpp:77-104) or once
 *             (multiscale_steps == 0; icp.cpp:105-122); an item that fails at a scale stops there, as
 *             match() does.  Items whose filtered target is still too large, or whose leaf lattice
 *             overflows int32, are registered by wm_icp_match inside the call.
 * Results per item k: status[k] (what wm_icp_match would have returned), T_out + 16 k (written
 * when status[k] == WM_OK), info_out + 36 k (with_info; written whenever an align of the item ran),
 * stats[k] (of the item's last align).  T_out, info_out, stats may be NULL.  The call returns WM_OK
 * when the batch ran.  In stats[k], align_ms is the device time of the launches the item took part
 * in; nn_ms / stats_ms / solve_ms / coarse_ms carry a developer aid instead of times: shader-clock
 * kilocycles of the item's LAST iteration (query loop, row sum, solve) and of its set-up. */
#define WM_BATCH_LDS_TARGET_POINTS 10000
#define WM_BATCH_MAX_TARGET_POINTS 65535
typedef struct {
    const void *src;    /* wave `ref`    */
    size_t n_src;
    const void *target; /* wave `target` */
    size_t n_target;
} wm_batch_item;
int wm_icp_batch_match(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride_bytes,
                       int mem, const wm_icp_params *p, float res, int multiscale_steps,
                       int with_info, double *T_out, double *info_out, wm_icp_stats *stats,
                       int *status);

/* pcl::VoxelGrid<PointXYZ>::filter of MANY clouds in one pass (the filter stage of wm_icp_batch_match
 * with res > 0, exposed): cloud 2 k = items[k].src, 2 k + 1 = items[k].target; their centroids, back
 * to back in that order, as packed xyz floats in host memory out_xyz (capacity cap_points points),
 * their counts in n_out[2 n_items].  Same results, bit for bit, as wm_voxel_downsample cloud by cloud
 * (a leaf lattice beyond int32 is WM_ERR_ARG here). */
int wm_voxel_downsample_batch(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride_bytes,
                              int mem, float leaf, float *out_xyz, size_t cap_points, size_t *n_out);

/* pcl::VoxelGrid<PointXYZ>::filter on device (icp.cpp:81-90,106-113; gicp.cpp:39-40,
 * 49-50): float centroid per occupied leaf, ascending leaf index; `out` must hold
 * `cap` points of `out_stride` bytes. */
int wm_voxel_downsample(wm_ctx *ctx, const void *pts, size_t n, size_t stride_bytes, int mem,
                        float leaf, void *out, size_t out_stride, int out_mem, size_t cap,
                        size_t *n_out);

/* pcl::transformPointCloud(in, out, Eigen::Affine3d) on device (icp.cpp:84-86; every
 * reference test fixture, tests/icp_tests.cpp:31): double arithmetic, float store. */
int wm_transform_cloud(wm_ctx *ctx, const void *pts, size_t n, size_t stride_bytes, int mem,
                       const double T[16], void *out, size_t out_stride, int out_mem);

/* ICPMatcher's information-matrix estimators on the correspondences of the last
 * align (wave_matching/src/icp.cpp:135-142):
 *   WM_INFO_LUM     estimateLUM     icp_pcl_functions.cpp:182-289
 *   WM_INFO_CENSI   estimateCensi   icp.cpp:167-397   (T_result = Matcher::result,
 *                   lin/ang_covar = ICPMatcherParams::lidar_{lin,ang}_covar)
 *   WM_INFO_LUMOLD  estimateLUMold  icp_pcl_functions.cpp:51-179  (fresh exact NN of
 *                   the aligned cloud, d2 < max_corr^2)
 * info is the 6x6 row-major result.  *degenerate (may be NULL) is set when the LUM
 * residual s^2 underflows: LUM then yields identity, LUMold divides by it anyway
 * (both as the reference does).  LUM / Censi return WM_NOT_CONVERGED and leave info
 * untouched when the last align did not converge (the reference's hasConverged()
 * guard). */
enum { WM_INFO_LUM = 0, WM_INFO_CENSI = 1, WM_INFO_LUMOLD = 2 };
int wm_icp_info(wm_ctx *ctx, int method, const double T_result[16], double lin_covar,
                double ang_covar, double max_corr, double info[36], int *degenerate);

/* Device-to-device copy rate of this GPU (GB/s, read + write bytes) measured with a plain float4
 * copy kernel on the context's stream: the practical HBM ceiling to hold next to the 8 TB/s
 * specification (bench.py's roofline.peak_measured_copy). */
int wm_debug_copy_bandwidth(wm_ctx *ctx, size_t bytes, int reps, double *gb_per_s);
/* developer / tests, HOST ONLY (no device is touched): the arithmetic the ICP loop's sums go through on the device
 * (libwave_amd/csrc/wm_bins.hpp) -- every x[i] cut into three signed 40-bit limbs of trunc(x * 2^56), the limbs added as
 * 64-bit integers into 64 bins in the order perm[] names (NULL: as given), the totals turned back into one double.
 * *out = that double; limbs_out (may be NULL): the three limb totals.  WM_ERR_ARG when an x[i] does not fit (not finite,
 * |x| >= 2^62: the device poisons the bin for those). */
int wm_debug_bins_sum(const double *x, size_t n, const unsigned *perm, double *out, long long limbs_out[3]);
/* Developer / tests: the library's own stable radix sort (wm_sort.hpp: what every cloud's Morton order, the NDT voxel
 * order and the voxel filter are built on above 256k points) on HOST arrays: values_out[i] = the input position of the
 * i-th pair in ascending order of the low `bits` bits of the key (equal keys in input order).  key_bytes: 4 or 8. */
int wm_debug_sort_pairs(wm_ctx *ctx, const void *keys, int key_bytes, size_t n, unsigned bits, unsigned *values_out);
/* Tuning knobs by name (tests, benchmarks; the defaults are the product's): "cert_from" (-1: the
 * certificate kernel takes over once an ICP step is small, -2: never, k >= 0: from iteration k of
 * every align), "cert_disp" (that step size, in level-0 grid cells), "cert_pad_mul",
 * "cert_pad_frac", "cert_nb", "gicp_served" (0 / 1 / 2: GICP's objective evaluations launched / served by the
 * resident evaluator / served without the on-chip pair cache), "late" (1: the late ICP iterations -- certificate,
 * searches, sums, solve, stopping rules -- in ONE resident launch, k_nn_cert<.., LATE> + k_late_solver; 0, the
 * default: a launch per iteration, which measures the same or faster), "bins" (0: an ICP iteration's sums as rows
 * of partial sums and a reduction launch; 1, the default: as exact integer limbs in bins), "ndt_vox_split" (NDT model:
 * the points per voxel up to which a lane, not a wave, forms a voxel's sums; -1: the library's choice) and "ndt_keys64"
 * (1: 64-bit voxel sort keys whatever the lattice's size) -- both rebuild the model at the next NDT call.  None of them
 * changes a result.  WM_ERR_ARG for an unknown name. */
int wm_set_option(wm_ctx *ctx, const char *name, double value);
/* developer: out == NULL arms a log of `iterations` launches (0 disarms); otherwise writes, per
 * launch of the certificate kernel since, how many queries it had to search; returns the count */
int wm_debug_cert_log(wm_ctx *ctx, int iterations, unsigned *out, int cap);
/* developer (armed by wm_debug_cert_log with WM_CERT_PROF set): 64 words per launch -- 16 cycle stamps
 * of each of four sampled workgroups (see k_nn_cert) */
int wm_debug_cert_prof(wm_ctx *ctx, unsigned long long *out, int cap);
/* developer: the per-iteration records the solve kernels of the last align published for the host
 * ([k], k >= 1: iteration : 16 | step size bfloat16 : 16 | changed matches : 16 | searched : 16) */
int wm_debug_pub_log(wm_ctx *ctx, unsigned long long *out, int cap);
/* Per-iteration device time (ms) of the correspondence kernel in the last
 * wm_icp_align call that ran with profile >= 1; returns the number written. */
int wm_get_iteration_times(wm_ctx *ctx, float *nn_ms, int cap);

/* Developer aid: shader-clock stamps of the last solve kernel of the last align (start, rows
 * added, statistics expanded, solve + stopping rules done). */
int wm_debug_solve_cycles(wm_ctx *ctx, unsigned long long out[8]);
/* Developer aid: per-query work of the grid search.  out == NULL arms the log for the first
 * `iterations` iterations of the next wm_icp_align on the current clouds (0 disarms and frees it);
 * out != NULL copies the log ([iteration][query in Morton order]: trips of the candidate loop in
 * bits 0-15, row chunks in bits 16-23, scan passes in bits 24-30, bit 31 = handed to the
 * wave-cooperative path) and returns the number of iterations recorded. */
int wm_debug_cost_log(wm_ctx *ctx, int iterations, unsigned *out, size_t cap);
/* with the cost log armed: per iteration, shader-clock cycle sums over the waves of the search kernel's
 * phases -- out[8 k + 0..7] = walk, pooled rounds, walks, prologue, pass loop, cooperative phase + stores,
 * statistics tail, waves.  Returns the number of iterations copied. */
int wm_debug_phase_log(wm_ctx *ctx, unsigned long long *out, int iterations);

/* PCL's icp.correspondences_ after align (read by estimateLUM / estimateCensi,
 * icp_pcl_functions.cpp:191, icp.cpp:213): per source point (caller's order)
 * the matched target index (caller's order; -1 = none) and squared distance. */
int wm_get_correspondences(wm_ctx *ctx, int32_t *match_idx, float *d2, size_t cap);

/* One correspondence pass with a caller-given transform (kernel-level parity
 * and roofline measurement): CorrespondenceEstimation::determineCorrespondences
 * [PCL registration/impl/correspondence_estimation.hpp] as driven by
 * icp.align (icp.cpp:126).  kernel_ms (may be NULL) receives the device time
 * of the level-0 correspondence kernel. */
int wm_nn_search(wm_ctx *ctx, const double T[16], double max_corr, int nn_method,
                 int32_t *match_idx, float *d2, size_t cap, float *kernel_ms);

/* The sufficient statistics one ICP iteration reduces to (what crosses xGMI in
 * the multi-GPU path).  SVD mode: stats[0]=n, [1..3]=sum p, [4..6]=sum q,
 * [7..15]=sum q p^T (row-major), [16]=sum d2.  GN6 mode: [0]=n, [1]=sum d2,
 * [2..22]=upper triangle of J^T J (row-major), [23..28]=J^T r.  Both: [31] = number of
 * source points this context handled (== cloud size unless sharded).  Uses the
 * correspondences of the last wm_nn_search / align iteration. */
#define WM_STATS_LEN 32
int wm_icp_stats_for(wm_ctx *ctx, const double T[16], int mode, double stats[WM_STATS_LEN]);

/* Host-only solvers on those statistics (no GPU touched; used by every rank
 * after the all-reduce).  Tk_out is the incremental transform of one step:
 * TransformationEstimationSVD / pcl::umeyama [PCL transformation_estimation_svd.hpp]. */
int wm_umeyama_from_stats(const double stats[WM_STATS_LEN], double Tk_out[16]);
int wm_gn6_from_stats(const double stats[WM_STATS_LEN], double Tk_out[16]);

/* ------------------------------------------------------------------ GICP */
typedef struct {
    int corr_rand;        /* GICPMatcherParams::corr_rand, gicp.hpp:34 -> gicp.cpp:31 */
    int max_iter;         /* GICPMatcherParams::max_iter,  gicp.hpp:35 -> gicp.cpp:32 */
    double r_eps;         /* GICPMatcherParams::r_eps,     gicp.hpp:36 -> gicp.cpp:33 */
    double t_eps;         /* PCL default 5e-4 (libwave never sets it) */
    double max_corr;      /* PCL-GICP default 5 m (libwave never sets it) */
    double gicp_epsilon;  /* PCL default 1e-3 */
    int max_inner;        /* PCL default 20 BFGS iterations per outer iteration */
    int force_iterations; /* >0: exactly this many outer iterations (bench) */
    int objective;        /* how estimateRigidTransformationBFGS's objective is evaluated: WM_GICP_OBJECTIVE_* below */
    int reserved;
} wm_gicp_params;
/* WM_GICP_OBJECTIVE_PCL_SUMS (0, the default -- what the reference runs, wave_matching/src/gicp.cpp:58): PCL's per-pair
 * float path (OptimizationFunctorWithIndices::fdf summed pair by pair at every trial point), bit for bit the oracle's
 * default mode (oracle/gicp.c, objective mode 0).
 * WM_GICP_OBJECTIVE_STATISTICS (1, an explicit opt-in: NOT PCL's arithmetic): between two correspondence searches the
 * pairs and their Mahalanobis matrices are fixed and the residual is affine in the transform's entries, so the objective PCL sums pair by pair at
 * every trial point of the line search (OptimizationFunctorWithIndices::fdf, ~170 passes over the pairs per
 * registration) is formed ONCE per outer iteration as 74 sufficient statistics around the pairing transform; every
 * evaluation is then scalar work (libwave_amd/csrc/wm_gicp_quad.hpp).  The statistics' base point is PCL's float
 * residual; away from it the per-point float rounding of PCL's transform (a relative ~4e-7 of f) is absent.  PCL's
 * BFGS stops at a gradient tolerance of 1e-2 wherever its line search lands, so registrations of noisy pairs end
 * 1e-5 .. 1e-3 m apart between the two objectives -- the spread PCL's own result has against its summation order --
 * and pairs that register sharply (the reference's test cases) within 1e-4 m / 1e-5 rad; neither is systematically
 * closer to ground truth (tests/test_gicp_quad_gpu.py, tests/test_oracle_cpu.py).  That spread is ABOVE north_star's
 * 1e-4 m on noisy pairs, which is why this form is not the default.  Held bit for bit to the oracle's restatement of the
 * builder's reformulation (oracle/gicp.c, objective mode 1 -- a restatement check, not reference parity), on the
 * one-pair and the batched path.  The numbers equal the oracle's objective modes. */
#define WM_GICP_OBJECTIVE_PCL_SUMS 0
#define WM_GICP_OBJECTIVE_STATISTICS 1

typedef struct {
    int converged, iterations, n_corr, inner_total, evaluations;
    double f_final;
    float fdf_kernel_ms; /* summed device time of the objective/gradient kernel */
    int served_evaluations; /* of `evaluations`: answered by the resident evaluator (no kernel launch each) */
} wm_gicp_stats;

void wm_gicp_default_params(wm_gicp_params *p);
/* pcl::GeneralizedIterativeClosestPoint::align + hasConverged + getFinalTransformation
 * (wave_matching/src/gicp.cpp:58-60) on the clouds given by wm_set_source/wm_set_target.
 * Per-point covariances (computeCovariances, k = corr_rand) are computed on device and
 * cached per cloud. */
int wm_gicp_align(wm_ctx *ctx, const wm_gicp_params *p, double T_out[16], wm_gicp_stats *stats);
/* GICPMatcher::setRef + setTarget + match in one call (gicp.cpp:37-64): both clouds are
 * voxel-filtered on device when res > 0. */
int wm_gicp_match(wm_ctx *ctx, const void *ref, size_t n_ref, const void *target, size_t n_target,
                  size_t stride_bytes, int mem, const wm_gicp_params *p, float res, double T_out[16],
                  wm_gicp_stats *stats);
/* GICPMatcher::setRef / setTarget with res > 0 (wave_matching/src/gicp.cpp:38-45, 48-55): VoxelGrid
 * the cloud on the device and make the FILTERED copy the registration's source / target -- at the
 * time of the call (a snapshot, as the reference's filtered copy is). */
int wm_set_source_filtered(wm_ctx *ctx, const void *pts, size_t n, size_t stride_bytes, int mem, float leaf);
int wm_set_target_filtered(wm_ctx *ctx, const void *pts, size_t n, size_t stride_bytes, int mem, float leaf);
/* GICPMatcher::setRef + setTarget + match (wave_matching/src/gicp.cpp:37-64) for MANY pairs in one launch --
 * what a wave::MultiMatcher<GICPMatcher> has waiting in its queue (multi_matcher.hpp:29-34): one registration per
 * compute unit, the whole of align inside the kernel (grids, covariances, the outer loop with its 1-NN search,
 * Mahalanobis matrices and the BFGS minimisation -- the optimiser's code runs on the device here).
 *   res > 0: every cloud of the batch is voxel-filtered first (one pass of device-wide kernels), as
 *            setRef / setTarget do; res <= 0: the clouds as given, at most WM_GICP_BATCH_MAX_POINTS points each
 *            (beyond: WM_ERR_ARG).  A pair whose FILTERED cloud is larger is registered by wm_gicp_match.
 * Per item k: status[k] as wm_gicp_match would return it (WM_OK / WM_NOT_CONVERGED / WM_TOO_FEW_CORRESPONDENCES;
 * WM_ERR_STATE for an empty cloud), T_out + 16 k written when WM_OK, stats[k] (may be NULL).
 * Same neighbours, covariances, objective terms and (double-double) sums as wm_gicp_align, and the float
 * sinf / cosf / atan2f / asinf PCL's transform goes through are glibc's algorithms restated (wm_bfgs.hpp), so on
 * every pair tested the transform, objective value and iteration / evaluation counts EQUAL the one-pair path's and
 * the oracle's -- which is what tests/test_gicp_batch_gpu.py asserts (np.array_equal).  What is GUARANTEED is
 * 1e-6 m / 1e-6 rad: the double sin / cos of the gradient's rotation part are the device library's, and a
 * last-bit difference there can, after hundreds of evaluations of a registration that does not converge, take a
 * line-search branch the other way.
 * kernel_ms (may be NULL): device time of the launch. */
#define WM_GICP_BATCH_MAX_POINTS 100000
int wm_gicp_batch_match(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride_bytes, int mem,
                        const wm_gicp_params *p, float res, double *T_out, wm_gicp_stats *stats, int *status,
                        float *kernel_ms);
/* OptimizationFunctorWithIndices::fdf once: pairs + Mahalanobis matrices formed with
 * T_pair as one outer iteration does, then f and gradient at x = (t, roll, pitch, yaw). */
int wm_gicp_eval(wm_ctx *ctx, const wm_gicp_params *p, const double T_pair[16], const double x[6],
                 double *f, double g[6], int *n_pairs);
/* computeCovariances of both clouds (9 doubles per point, caller order); either output
 * may be NULL (kernel-level parity). */
int wm_gicp_covariances(wm_ctx *ctx, int k, double eps, double *cov_source, double *cov_target);

/* ------------------------------------------------------------------- NDT */
typedef struct {
    double res;           /* NDTMatcherParams::res,       ndt.hpp:40 -> ndt.cpp:32 setResolution */
    double step_size;     /* NDTMatcherParams::step_size, ndt.hpp:37 -> ndt.cpp:31 setStepSize  */
    double t_eps;         /* NDTMatcherParams::t_eps,     ndt.hpp:39 -> ndt.cpp:30              */
    int max_iter;         /* NDTMatcherParams::max_iter,  ndt.hpp:38 -> ndt.cpp:33              */
    double outlier_ratio; /* PCL default 0.55 (not set by libwave) */
    int skip_line_search; /* 1: PCL 1.8.x behaviour (More-Thuente loop never runs) */
    int pcl_d1_sign;      /* 1: PCL's h_ang_d1[2] = +sy; 0: the true derivative -sy */
    int force_iterations; /* >0: exactly this many Newton iterations (bench) */
} wm_ndt_params;

typedef struct {
    int converged, iterations, n_voxels, evaluations;
    double score;          /* trans_probability_: score / number of source points */
    float deriv_kernel_ms; /* summed device time of the derivative kernel; 0 unless WM_NDT_PROFILE=1 */
    int model_builds;      /* voxel models this context has built so far (an unchanged target keeps its
                              model across align calls, as PCL's setInputTarget does) */
} wm_ndt_stats;

void wm_ndt_default_params(wm_ndt_params *p);
/* pcl::NormalDistributionsTransform::align + hasConverged + getFinalTransformation
 * (wave_matching/src/ndt.cpp:59-61) on the clouds given by wm_set_source /
 * wm_set_target; the voxel statistics of setInputTarget (ndt.cpp:55) are (re)built on
 * device when the target or `res` changed. */
int wm_ndt_align(wm_ctx *ctx, const wm_ndt_params *p, double T_out[16], wm_ndt_stats *stats);
/* pcl::NormalDistributionsTransform::setInputTarget builds the voxel grid when it is called
 * (wave_matching/src/ndt.cpp:55): the model of the current target at resolution `res`, now; later
 * wm_ndt_align calls with the same target and res reuse it. */
int wm_ndt_build_model(wm_ctx *ctx, double res);
/* NDTMatcher::setRef + setTarget + match (wave_matching/src/ndt.cpp:48-65) for MANY pairs in one launch -- what a
 * wave::MultiMatcher<NDTMatcher> has waiting in its queue (multi_matcher.hpp:29-34): one registration per compute unit,
 * the target's voxel model (pcl::VoxelGridCovariance), the Newton steps and the More-Thuente line search all inside the
 * kernel.  Clouds of at most WM_NDT_BATCH_MAX_POINTS points (beyond: WM_ERR_ARG); a pair whose voxel lattice does not
 * fit the kernel's table (262 144 cells) is registered by wm_ndt_align.
 * Per item k: status[k] as wm_ndt_align would return it (WM_OK / WM_NOT_CONVERGED; WM_ERR_STATE for an empty cloud;
 * WM_ERR_ARG for a target spanning more than 2^20 voxels along an axis -- e.g. one far-away garbage return),
 * T_out + 16 k written when WM_OK, stats[k] (may be NULL).  Same voxel membership, radius tests and terms as
 * wm_ndt_align; a voxel's sums are formed in double-double instead of in point order and exp / log / sin / cos are the
 * device library's: results agree with the one-pair path to ~1e-9, not bit for bit. */
#define WM_NDT_BATCH_MAX_POINTS 200000
int wm_ndt_batch_match(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride_bytes, int mem,
                       const wm_ndt_params *p, double *T_out, wm_ndt_stats *stats, int *status, float *kernel_ms);
/* computeDerivatives at pose (tx,ty,tz,rx,ry,rz): score, gradient(6), Hessian(36) */
int wm_ndt_derivatives(wm_ctx *ctx, const wm_ndt_params *p, const double pose[6], double *score,
                       double grad[6], double hess[36], int *n_voxels);

/* ---------------------------------------------- sharded registration (multi-GPU)
 * One registration spread over the GPUs of a node, one process (rank) per GPU:
 * every rank holds the full source cloud and the target points of one x-slab
 * [x_lo - max_corr, x_hi + max_corr] (wm_set_target of that subset), handles the
 * source points whose TRANSFORMED x falls into [x_lo, x_hi) (so each source point
 * is handled by exactly one rank and its true neighbour within max_corr is in that
 * rank's subset), and the WM_STATS_LEN-double statistics block is summed over ranks
 * (RCCL all-reduce over xGMI) once per iteration; every rank then applies the same
 * solve, so the transform never needs broadcasting.
 *   wm_icp_shard_begin        reset the iteration state (like the start of align())
 *   wm_icp_shard_local_stats  enqueue search + reduction; writes this rank's partial
 *                             statistics to stats_dev (WM_STATS_LEN doubles in HBM)
 *   <caller all-reduces stats_dev (sum) on a stream ordered after the ctx stream>
 *   wm_icp_shard_apply        enqueue the solve + stopping rules from stats_dev
 *   wm_icp_shard_poll         sync; report done / transform / statistics
 * All enqueue calls are asynchronous on the context's stream (wm_ctx_set_stream).
 * A rank need not hold the full source cloud: any superset of the points that can fall
 * into its slab will do (e.g. those within a band around the slab).  To keep that exact,
 * pass expect_owned_total = the number of (finite) source points of the whole cloud:
 * stats[WM_STATS_LEN-1] carries the number of points each rank handled, and every
 * iteration in which the all-reduced count differs is tallied in
 * wm_icp_stats.owned_violations -- the caller then redoes the registration with wider
 * bands / full clouds.  0 disables the check. */
int wm_icp_shard_begin(wm_ctx *ctx, const wm_icp_params *p, double x_lo, double x_hi,
                       size_t expect_owned_total);
int wm_icp_shard_local_stats(wm_ctx *ctx, void *stats_dev);
int wm_icp_shard_apply(wm_ctx *ctx, const void *stats_dev);
int wm_icp_shard_poll(wm_ctx *ctx, int *done, double T_out[16], wm_icp_stats *stats);

/* Sharded NDT ("replicate the target, shard ref by index range"; no reference
 * counterpart -- PCL's NDT is single-threaded).  Every rank sets the SAME two clouds and builds
 * the same voxel model (cheap: < 0.5 ms at 2M points); the Morton-ordered source is dealt out
 * to the ranks in chunks of 4096 points, round-robin (every rank works on the whole scene at
 * 1/world density: equal work), rank r evaluates its chunks in every derivative pass, and the
 * sums of that pass -- n = 28 (score, gradient, upper triangle of the Hessian) or n = 7 (score,
 * gradient: the line search's passes) -- are summed over the ranks by `reduce` before the
 * Newton / More-Thuente logic sees them.  `reduce` must leave bit-identical
 * values on every rank (an RCCL / gloo all-reduce does): all ranks then take the same decisions
 * and return the same transform, with no broadcast.  It is called on the thread that called
 * wm_ndt_align, with a host array; return 0 on success (anything else aborts the registration
 * with WM_ERR_STATE).  world = 1 or reduce = NULL restores the single-GPU behaviour. */
typedef int (*wm_allreduce_fn)(double *vals, int n, void *user);
int wm_ndt_set_shard(wm_ctx *ctx, int rank, int world, wm_allreduce_fn reduce, void *user);

/* ------------------------------------------ sharded registration, driven from C (RCCL inside)
 * A wm_comm is one rank's handle on a group of `world` ranks, one GPU each (librccl is linked into this
 * library).  The ICP loop's exchange step -- 34 doubles per rank and iteration -- goes through MAILBOXES where
 * they can be set up: fine-grained device memory of every rank, mapped into its peers (IPC handles all-gathered
 * over the communicator between processes, peer access inside one), into which the solve kernel of every rank
 * stores its block over xGMI and out of which it adds all blocks in rank order (csrc/wm_xchg.hpp: no launch
 * between a rank's sums and its solve).  A probe exchange at creation and an all-reduced verdict decide, alike
 * on every rank, whether they are used; otherwise, and with WM_COMM_P2P=0, the step is an RCCL all-reduce on the
 * calling context's stream.  A peer whose block does not arrive within WM_COMM_P2P_TIMEOUT_MS (default 5000)
 * fails the registration with WM_ERR_RCCL instead of hanging -- and that rank's communicator then goes back to the
 * collective exchange for good (wm_multi_* run the failed registration once more by themselves; bench.py --gpus N
 * rebuilds its communicators without mailboxes if its first registration fails on any rank).
 * wm_icp_stats.exchange_in_kernel says which exchange ran.
 * No reference counterpart: libwave's only parallelism is one matcher per thread
 * (wave_matching/include/wave/matching/multi_matcher.hpp:32).
 *   wm_comm_get_unique_id + wm_comm_init_rank   one rank per process (or thread): rank 0 creates the
 *        128-byte id, the launcher hands it to every rank (bench.py: torch.distributed broadcast;
 *        MPI_Bcast; a file), every rank calls init_rank on its device.   = ncclCommInitRank
 *   wm_comm_init_all     all ranks in one process, `devices[r]` for rank r.  = ncclCommInitAll
 *   wm_comm_init_local   test stand-in: `n` ranks on ONE device whose all-reduce is a host-side sum
 *        in rank order at a barrier (each rank must run on its own thread).  WM_COMM_P2P_LOCAL=1 gives these
 *        ranks mailboxes too (tests of the protocol; on one GPU a rank's first-call allocations wait for the
 *        other rank's polling kernel, so contexts must have registered once before). */
typedef struct wm_comm wm_comm;
#define WM_COMM_ID_BYTES 128
int wm_comm_get_unique_id(void *id_out /* WM_COMM_ID_BYTES */);
int wm_comm_init_rank(wm_comm **out, int device, const void *id, int rank, int world);
int wm_comm_init_all(wm_comm **comms /* [n] */, const int *devices, int n);
int wm_comm_init_local(wm_comm **comms /* [n] */, int n, int device);
void wm_comm_destroy(wm_comm *comm);
/* us per exchange of the loop's block (the mailbox exchange in a kernel of its own, or ncclAllReduce), `reps` back
 * to back on the context's stream (collective) */
int wm_comm_allreduce_probe(wm_ctx *ctx, wm_comm *comm, int reps, double *us_out);
int wm_comm_rank(const wm_comm *comm);
int wm_comm_world(const wm_comm *comm);
/* 1 while the group's sharded ICP loop exchanges through the ranks' mailboxes, 0 once it uses ncclAllReduce (never set
 * up, WM_COMM_P2P=0, or dropped after an exchange that failed -- on EVERY rank alike: the loop's commit round makes the
 * ranks agree).  wm_comm_set_exchange_timeout_ms: how long a solve kernel waits for a peer's block (default 5000 ms;
 * 0 = a block that is not there at once counts as never coming: tests). */
int wm_comm_mailboxes(const wm_comm *comm);
int wm_comm_set_exchange_timeout_ms(wm_comm *comm, int ms);

/* pcl::IterativeClosestPoint::align (wave_matching/src/icp.cpp:126-129) as ONE registration over
 * all ranks of `comm`.  Collective: every rank calls it with the same two (full) clouds and
 * parameters, on its own context (`mem` = where the clouds live for THIS rank).  Inside the call:
 * equal-count x-slabs of the target from a histogram of a fixed sub-sample of 16 384 points (identical on
 * all ranks, no communication), this rank's slab + max_corr halo of the target and its band of the source
 * selected out of the clouds as the caller laid them out (one stable compaction of both), the index over
 * them, then per iteration search + local sums -> exchange of 34 doubles (the ranks' mailboxes inside the
 * solve kernel, or ncclAllReduce: see wm_comm above) -> solve, all enqueued on the context's stream.  A peer
 * whose block does not arrive in time fails this rank's call with WM_ERR_RCCL.  Host clouds: with an RCCL communicator rank 0 uploads, the others receive
 * over xGMI (ncclBroadcast).  Every rank returns the same transform and status; `stats` carries the
 * rank's per-phase budget.  A rank that fails with a HIP / RCCL error aborts the communicator (its
 * peers' pending collectives fail instead of waiting for ever); the communicator is finished then.
 * comm == NULL or a world of 1 is plain wm_set_source + wm_set_target + wm_icp_align. */
int wm_icp_align_sharded(wm_ctx *ctx, wm_comm *comm, const void *ref, size_t n_ref, const void *target,
                         size_t n_target, size_t stride_bytes, int mem, const wm_icp_params *p,
                         double T_out[16], wm_icp_stats *stats);
/* ICPMatcher::match() (wave_matching/src/icp.cpp:75-133) over the ranks of `comm`, voxel-filtered
 * branches included -- the reference's DEFAULT parameters (res 0.1, three coarser scales,
 * icp.hpp:54,59): every rank filters both clouds itself (deterministic: all hold the same filtered
 * clouds) and the align of every scale is sharded.  res <= 0: wm_icp_align_sharded. */
int wm_icp_match_sharded(wm_ctx *ctx, wm_comm *comm, const void *ref, size_t n_ref, const void *target,
                         size_t n_target, size_t stride_bytes, int mem, const wm_icp_params *p, float res,
                         int multiscale_steps, double T_out[16], wm_icp_stats *stats);
/* wm_icp_info after a sharded registration (ICPMatcher::estimateInfo, icp.cpp:135-142).  Collective:
 * every rank adds up the pairs of the queries it owned, the 16 / 1 / 42 sums are exchanged over `comm`,
 * every rank finishes the same 6x6.  (wm_icp_info itself refuses after a sharded align: one rank's
 * pairs alone say nothing.) */
int wm_icp_info_sharded(wm_ctx *ctx, wm_comm *comm, int method, const double T_result[16], double lin_covar,
                        double ang_covar, double max_corr, double info_out[36], int *degenerate);

/* Sharded NDT with the exchange on the device: like wm_ndt_set_shard, but the sums of every
 * derivative pass are all-reduced in HBM over `comm` (RCCL) before the host fetches them -- no
 * callback, no host bounce.  comm == NULL switches sharding off. */
int wm_ndt_set_comm(wm_ctx *ctx, wm_comm *comm);

/* All ranks in ONE process: one context and one worker thread per device, RCCL communicators from
 * ncclCommInitAll (emulate != 0: `n_devices` ranks on devices[0] with the host stand-in exchange).
 * wm_multi_icp_align runs one sharded registration of two HOST clouds (uploaded once, broadcast over
 * xGMI) and returns rank 0's result; wm_multi_icp_match is ICPMatcher::match() with its voxel filter
 * and scales; wm_multi_icp_info the estimators after either. */
typedef struct wm_multi wm_multi;
int wm_multi_create(wm_multi **out, const int *devices, int n_devices, int emulate);
void wm_multi_destroy(wm_multi *m);
int wm_multi_size(const wm_multi *m);
int wm_multi_icp_align(wm_multi *m, const void *ref, size_t n_ref, const void *target, size_t n_target,
                       size_t stride_bytes, const wm_icp_params *p, double T_out[16], wm_icp_stats *stats);
int wm_multi_icp_match(wm_multi *m, const void *ref, size_t n_ref, const void *target, size_t n_target,
                       size_t stride_bytes, const wm_icp_params *p, float res, int multiscale_steps,
                       double T_out[16], wm_icp_stats *stats);
int wm_multi_icp_info(wm_multi *m, int method, const double T_result[16], double lin_covar, double ang_covar,
                      double max_corr, double info_out[36], int *degenerate);

/* Host-only twin of the per-iteration solve + PCL stopping rules (no GPU touched):
 * the very function the device runs after the all-reduce, callable on the CPU so
 * that the sharded control flow can be exercised without a GPU. */
typedef struct wm_host_icp wm_host_icp;
int wm_host_icp_create(wm_host_icp **out, const wm_icp_params *p, size_t expect_owned_total);
void wm_host_icp_destroy(wm_host_icp *h);
int wm_host_icp_apply(wm_host_icp *h, const double stats[WM_STATS_LEN]);
int wm_host_icp_get(const wm_host_icp *h, int *done, double T_out[16], wm_icp_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* WAVEMATCH_H */
