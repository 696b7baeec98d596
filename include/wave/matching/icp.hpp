// wave::ICPMatcher -- drop-in for the reference's
// wave_matching/include/wave/matching/icp.hpp:30-125.  Same parameter struct (field
// names, defaults, YAML constructor), same class surface; the PCL members
// (pcl::IterativeClosestPoint, pcl::VoxelGrid; icp.hpp:100-102) are replaced by a
// wm_ctx handle into libwavematch_hip.so (include/wavematch.h).
#ifndef WAVE_MATCHING_ICP_HPP
#define WAVE_MATCHING_ICP_HPP

#include <string>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"

struct wm_ctx;

namespace wave {

struct ICPMatcherParams {
    ICPMatcherParams(const std::string &config_path);
    ICPMatcherParams() {}

    /// Maximum distance to correspond points for icp
    double max_corr = 3;
    /// Maximum iterations of ICP
    int max_iter = 100;
    /// Transformation epsilon. Stopping criteria.
    double t_eps = 1e-8;
    /// Stopping criteria, if cost function decreases by less than this, stop
    double fit_eps = 1e-2;
    /// Angular variance for lidar sensor model (Censi covariance estimation)
    double lidar_ang_covar = 7.78e-9;
    /// Linear variance for lidar sensor model (Censi covariance estimation)
    double lidar_lin_covar = 2.5e-4;
    /// >0: each match is performed from a coarse to fine scale; each step doubles the resolution
    int multiscale_steps = 3;
    /// Voxel side length for downsampling (<= 0: none)
    float res = 0.1;
    enum covar_method : int { LUM, CENSI, LUMold } covar_estimator = covar_method::LUM;
};

class ICPMatcher : public Matcher<PCLPointCloudPtr> {
 public:
    explicit ICPMatcher(ICPMatcherParams params1);
    ICPMatcher(const ICPMatcher &other);  // MultiMatcher stores matchers by value
    ICPMatcher &operator=(const ICPMatcher &other);
    ~ICPMatcher();

    /** sets the reference pointcloud (aliased, read at match() time; icp.cpp:67-69) */
    void setRef(const PCLPointCloudPtr &ref);
    /** sets the target pointcloud (icp.cpp:71-73) */
    void setTarget(const PCLPointCloudPtr &target);
    /** runs ICP matcher. Blocks until finished. true if successful (icp.cpp:75-133) */
    bool match();
    /** runs the covariance estimators (icp.cpp:135-142) */
    void estimateInfo();

    ICPMatcherParams params;

    /** extension: HIP device ordinal new matchers bind to (default: env
     * WAVE_MATCHING_DEVICE, else 0) */
    static void setDefaultDevice(int device);

 private:
    wm_ctx *ctx;  // created lazily in the thread that first matches
    int device;
    bool converged;
    PCLPointCloudPtr ref, target;

    bool ensureContext();
    void estimateLUM();
    void estimateLUMold();
    void estimateCensi();
};

}  // namespace wave

#endif  // WAVE_MATCHING_ICP_HPP
