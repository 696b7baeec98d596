/*
 * wm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the registration algorithms that libwave's
 * wave_matching module delegates to PCL 1.8 (ICP / GICP / NDT / VoxelGrid /
 * transformPointCloud) plus libwave's own information-matrix estimators
 * (wave_matching/src/icp.cpp:167-397, wave_matching/src/icp_pcl_functions.cpp:51-289).
 *
 * PARITY STATUS: "parity unpinned" BY THE REFERENCE -- PCL / Eigen / FLANN are not vendored in
 * the reference tree and are not installed in the build image, and the reference ships no golden
 * transforms; its tests only assert |T - T_gt|_F < 0.1 / 0.12
 * (wave_matching/tests/icp_tests.cpp:37,59-61).  What this oracle IS pinned to:
 *   (1) those reference-test assertions on the reference's own fixture
 *       (tests/golden/testscan.pcd, sha256 c22245b9...);
 *   (2) independent numpy / scipy restatements whose outputs are committed under tests/golden/:
 *       icp_golden.json (make_golden.py: ICP + VoxelGrid with cKDTree and numpy SVD; final 4x4s,
 *       iteration counts, stop states) and gicp_ndt_golden.json (make_golden_gicp_ndt.py: NDT voxel
 *       model, score / gradient / Hessian from rotation-matrix derivative products -- not from the
 *       tabulated entries restated here --, the optimum a converged NDT reaches; GICP covariances,
 *       objective, gradient and the registration's fixed point).  tests/test_oracle_cpu.py and
 *       tests/test_golden_gicp_ndt_cpu.py hold the oracle to them; the HIP path is held to the
 *       same files (tests/test_golden_gicp_ndt_gpu.py, tests/test_match_gpu.py);
 *   (3) optionally, a system PCL: oracle/pcl_ref/ builds a PCL-backed third opinion when PCL >= 1.8
 *       is installed (it is not in this image; tests/test_pcl_third_opinion.py skips without it).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product path (libwave_amd/csrc) never does.
 *
 * All clouds are packed float32 XYZ arrays (n x 3, row-major).
 * All 4x4 / 6x6 matrices are row-major doubles.
 */
#ifndef WM_ORACLE_H
#define WM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------- kd-tree */
typedef struct wmo_kdtree wmo_kdtree;
wmo_kdtree *wmo_kdtree_build(const float *xyz, int n);
void wmo_kdtree_free(wmo_kdtree *t);
/* exact 1-NN, float L2^2 computed as (dx*dx + dy*dy) + dz*dz (no FMA),
 * ties broken towards the lowest point index. returns index, writes d2 */
int wmo_kdtree_nn(const wmo_kdtree *t, const float q[3], float *d2);
/* k-NN, ascending (d2, index). returns number found (min(k, n)) */
int wmo_kdtree_knn(const wmo_kdtree *t, const float q[3], int k, int *idx, float *d2);
/* all n queries; idx/d2 arrays of length nq */
void wmo_nn_batch(const wmo_kdtree *t, const float *q, int nq, int *idx, float *d2);
/* O(n*m) brute force with identical arithmetic & tie-break (validates the tree) */
void wmo_nn_brute(const float *tgt, int m, const float *q, int nq, int *idx, float *d2);

/* ------------------------------------------------------------- transforms */
/* pcl::transformPointCloud(cloud, cloud, Eigen::Affine3d) -- computed in
 * double, stored float (reference call: wave_matching/src/icp.cpp:84-86) */
void wmo_transform_cloud_d(const float *in, int n, const double T[16], float *out);
/* PCL ICP's internal float transform: ((m00*x + m01*y) + m02*z) + m03 */
void wmo_transform_cloud_f(const float *in, int n, const float T[16], float *out);

/* -------------------------------------------------------------- VoxelGrid */
/* pcl::VoxelGrid<PointXYZ>::filter (reference calls: icp.cpp:81-90,106-113,
 * gicp.cpp:39-40,49-50).  out must hold n points. returns number of output
 * points (== n and a verbatim copy when the index space overflows int). */
int wmo_voxel_grid(const float *in, int n, float leaf, float *out);

/* -------------------------------------------------------------------- ICP */
enum { WMO_ICP_SVD = 0, WMO_ICP_GN6 = 1 };
enum {
    WMO_CONV_NOT_CONVERGED = 0,
    WMO_CONV_ITERATIONS = 1,
    WMO_CONV_TRANSFORM = 2,
    WMO_CONV_ABS_MSE = 3,
    WMO_CONV_REL_MSE = 4,
    WMO_CONV_NO_CORRESPONDENCES = 5,
    WMO_CONV_FORCED = 6
};

typedef struct {
    double max_corr;       /* ICPMatcherParams::max_corr  (icp.hpp:35)  */
    int max_iter;          /* ICPMatcherParams::max_iter  (icp.hpp:37)  */
    double t_eps;          /* ICPMatcherParams::t_eps     (icp.hpp:41)  */
    double fit_eps;        /* ICPMatcherParams::fit_eps   (icp.hpp:43)  */
    int force_iterations;  /* >0: run exactly this many iterations, no stop tests */
    int mode;              /* WMO_ICP_SVD (PCL) or WMO_ICP_GN6 */
    int float_sums;        /* 1: accumulate Umeyama sums in float (Scalar=float) */
    int incremental_float; /* 1: PCL-literal in-place float re-transform + float
                              4x4 compounding; 0: apply cumulative double T to the
                              original cloud each iteration (what the GPU does) */
    double prev_mse_in;    /* carried DefaultConvergenceCriteria state; <0 = DBL_MAX */
} wmo_icp_params;

typedef struct {
    int converged;     /* pcl hasConverged() */
    int iterations;    /* nr_iterations_ */
    int state;         /* WMO_CONV_* */
    int n_corr;        /* correspondences in the last iteration */
    double mse;        /* mean d2 of the last iteration's correspondences */
    double prev_mse_out;
} wmo_icp_result;

void wmo_icp_default_params(wmo_icp_params *p);

/* src = wave `ref` (PCL source, moved), tgt = wave `target` (indexed).
 * T_out maps src -> tgt.  Optional outputs (may be NULL):
 *   corr_idx[n]  match index per source point of the LAST iteration (-1 = none)
 *   corr_d2[n]   its squared distance
 *   final_xyz[n*3] the aligned source cloud (PCL `output`)
 *   trace[2*max_iter]  per-iteration (n_corr, mse)
 * returns 0 on success (converged), 1 when not converged. */
int wmo_icp_align(const float *src, int n, const float *tgt, int m,
                  const wmo_icp_params *p, double T_out[16], wmo_icp_result *res,
                  int *corr_idx, float *corr_d2, float *final_xyz, double *trace);

/* ICPMatcher::match() incl. voxel / multiscale branches (icp.cpp:75-133).
 * res<=0 full resolution; res>0 && steps==0 single filtered; else multiscale.
 * Also leaves the down-sampled clouds / final cloud / correspondences needed
 * by the information estimators in *state (opaque, free with wmo_match_free). */
typedef struct wmo_match_state wmo_match_state;
int wmo_icp_match(const float *ref, int n, const float *target, int m,
                  const wmo_icp_params *p, float res, int multiscale_steps,
                  double T_out[16], wmo_icp_result *res_out, wmo_match_state **state);
void wmo_match_free(wmo_match_state *s);
int wmo_match_counts(const wmo_match_state *s, int *n_ref, int *n_target, int *n_corr);

/* ----------------------------------------------- information estimators */
/* estimateLUM (icp_pcl_functions.cpp:182-289): returns 0 ok, 1 = identity fallback */
int wmo_info_lum(const wmo_match_state *s, double info[36]);
/* estimateLUMold (icp_pcl_functions.cpp:51-179) incl. the fall-through quirk */
int wmo_info_lumold(const wmo_match_state *s, double max_corr, double info[36]);
/* estimateCensi (icp.cpp:167-397) */
int wmo_info_censi(const wmo_match_state *s, const double T[16], double lin_covar,
                   double ang_covar, double info[36]);

/* raw-array forms used for kernel-level parity tests.  p = aligned source
 * points (`final`), q = matched target points, both n_corr x 3 */
int wmo_lum_from_pairs(const float *p, const float *q, int n_corr, double info[36],
                       double mm_out[36], double mz_out[6], float *ss_out);
int wmo_censi_from_pairs(const float *ref_pts, const float *tgt_pts, int n_corr,
                         const double T[16], double lin_covar, double ang_covar,
                         double info[36], double d2j_dx2[36], double middle[36]);

/* ------------------------------------------------------------------- GICP */
typedef struct {
    int corr_rand;         /* k (gicp.hpp:34) */
    int max_iter;          /* outer iterations */
    double r_eps;          /* rotation_epsilon_ */
    double t_eps;          /* transformation_epsilon_ (PCL GICP default 5e-4) */
    double max_corr;       /* corr_dist_threshold_ (PCL GICP default 5.0) */
    double gicp_epsilon;   /* 1e-3 */
    int max_inner;         /* 20 */
    int force_iterations;
} wmo_gicp_params;
typedef struct {
    int converged, iterations, n_corr, inner_total;
    double f_final;
    int evaluations;       /* objective evaluations of all minimisations */
    int reserved;
} wmo_gicp_result;
void wmo_gicp_default_params(wmo_gicp_params *p);
/* summation of the objective's sums: 0 double-double (default), 1 PCL-literal plain doubles in index order */
void wmo_gicp_set_summation(int mode);
int wmo_gicp_get_summation(void);
/* the objective the minimisations evaluate: 0 (default) PCL's per-pair sums through the float transform
 * (OptimizationFunctorWithIndices::fdf); 1 the same objective as 74 sufficient statistics formed once per outer
 * iteration around the pairing transform -- the HIP path's default (libwave_amd/csrc/wm_gicp_quad.hpp) */
void wmo_gicp_set_objective(int mode);
int wmo_gicp_get_objective(void);
/* the statistics objective once: pairs found under the float transform T0 (row-major 4x4), evaluated at state x
 * on `base`; Q_out (may be NULL) receives the 74 sums */
double wmo_gicp_fdf_statistics(const float *src, const float *tgt, const int *src_idx, const int *tgt_idx,
                               const double *mahal, int n_pairs, const double base[16], const float T0[16],
                               const double x[6], double g[6], double Q_out[74]);
/* per-point covariances (n x 9 doubles, row-major 3x3) */
int wmo_gicp_covariances(const float *xyz, int n, int k, double eps, double *cov);
int wmo_gicp_align(const float *src, int n, const float *tgt, int m,
                   const wmo_gicp_params *p, double T_out[16], wmo_gicp_result *res);
/* f and gradient of the GICP objective for given pairs (kernel-level parity):
 * x = (tx,ty,tz,roll,pitch,yaw); base = 4x4 applied first.  mahal = n_src x 9, indexed by
 * source index like PCL's mahalanobis_ vector */
double wmo_gicp_fdf(const float *src, const float *tgt, const int *src_idx,
                    const int *tgt_idx, const double *mahal, int n_pairs,
                    const double base[16], const double x[6], double g[6]);

/* -------------------------------------------------------------------- NDT */
typedef struct {
    double res;            /* NDTMatcherParams::res */
    double step_size;      /* NDTMatcherParams::step_size (int in the reference) */
    double t_eps;
    int max_iter;
    double outlier_ratio;  /* 0.55 */
    int skip_line_search;  /* PCL-1.8 "interval_converged initialised true" variant */
    int pcl_d1_sign;       /* 1 = PCL's h_ang d1[2] = +sy (thesis typo); 0 = true -sy */
    int force_iterations;
} wmo_ndt_params;
typedef struct {
    int converged, iterations, n_voxels;
    double score;          /* trans_probability-like: score / n */
} wmo_ndt_result;
void wmo_ndt_default_params(wmo_ndt_params *p);
typedef struct wmo_ndt_grid wmo_ndt_grid;
wmo_ndt_grid *wmo_ndt_grid_build(const float *tgt, int m, double res);
void wmo_ndt_grid_free(wmo_ndt_grid *g);
int wmo_ndt_grid_size(const wmo_ndt_grid *g);
/* export voxels sorted by (k,j,i): ijk[3*V], mean[3*V], icov[9*V], count[V] */
void wmo_ndt_grid_export(const wmo_ndt_grid *g, int *ijk, double *mean, double *icov,
                         int *count);
/* score, gradient(6), Hessian(36) at pose p=(tx,ty,tz,rx,ry,rz) */
double wmo_ndt_derivatives(const wmo_ndt_grid *g, const float *src, int n,
                           const wmo_ndt_params *prm, const double p[6], double grad[6],
                           double hess[36]);
int wmo_ndt_align(const float *src, int n, const float *tgt, int m,
                  const wmo_ndt_params *p, double T_out[16], wmo_ndt_result *res);

/* ----------------------------------------------------------- small linalg */
/* exposed for unit tests */
void wmo_svd(int n, const double *A, double *U, double *S, double *V);
/* one-sided Jacobi SVD of a 3x3, IEEE operations in a fixed order (GICP covariances) */
void wmo_svd3_jacobi(const double *A, double *U, double *S, double *V);
void wmo_sym_eig(int n, const double *A, double *evals, double *evecs);
int wmo_inverse(int n, const double *A, double *Ainv);
void wmo_umeyama(const float *src, const float *dst, int n, int float_sums, double T[16]);
void wmo_euler_angles_012(const double R[9], double e[3]);

#ifdef __cplusplus
}
#endif
#endif
