// wave::GICPMatcher -- drop-in for the reference's
// wave_matching/include/wave/matching/gicp.hpp:30-70.  Same parameter struct and class
// surface; the pcl::GeneralizedIterativeClosestPoint / pcl::VoxelGrid members
// (gicp.hpp:61-62) are replaced by a wm_ctx.
#ifndef WAVE_MATCHING_GICP_HPP
#define WAVE_MATCHING_GICP_HPP

#include <string>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"

struct wm_ctx;

namespace wave {

struct GICPMatcherParams {
    GICPMatcherParams(const std::string &config_path);
    GICPMatcherParams() {}

    int corr_rand = 10;
    int max_iter = 100;
    double r_eps = 1e-8;
    double fit_eps = 1e-2;
    float res = 0.1;
};

class GICPMatcher : public Matcher<PCLPointCloudPtr> {
 public:
    explicit GICPMatcher(GICPMatcherParams params1);
    GICPMatcher(const GICPMatcher &other);
    ~GICPMatcher();

    /** sets the reference pointcloud (voxel-filtered at match() when res > 0; gicp.cpp:37-45) */
    void setRef(const PCLPointCloudPtr &ref);
    /** sets the target pointcloud (gicp.cpp:47-55) */
    void setTarget(const PCLPointCloudPtr &target);
    /** runs GICP matcher. Blocks until finished. true if successful (gicp.cpp:57-64) */
    bool match();

 private:
    wm_ctx *ctx;
    int device;
    PCLPointCloudPtr ref, target;
    GICPMatcherParams params;
    bool ensureContext();
};

}  // namespace wave

#endif  // WAVE_MATCHING_GICP_HPP
