// wave::NDTMatcher -- drop-in for the reference's
// wave_matching/include/wave/matching/ndt.hpp:33-85.  Same parameter struct (incl. the
// const min_res member and the int step_size), same class surface; the
// pcl::NormalDistributionsTransform member (ndt.hpp:72) is replaced by a wm_ctx.
#ifndef WAVE_MATCHING_NDT_HPP
#define WAVE_MATCHING_NDT_HPP

#include <string>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"

struct wm_ctx;

namespace wave {

struct NDTMatcherParams {
    NDTMatcherParams(){};
    NDTMatcherParams(const std::string &config_path);

    int step_size = 3;
    int max_iter = 100;
    double t_eps = 1e-8;
    float res = 5;
    const float min_res = 0.05f;
};

class NDTMatcher : public Matcher<PCLPointCloudPtr> {
 public:
    explicit NDTMatcher(NDTMatcherParams params1);
    NDTMatcher(const NDTMatcher &other);
    ~NDTMatcher();

    /** sets the reference pointcloud (ndt.cpp:48-51) */
    void setRef(const PCLPointCloudPtr &ref);
    /** sets the target pointcloud; builds the voxel model (ndt.cpp:53-56) */
    void setTarget(const PCLPointCloudPtr &target);
    /** runs NDT matcher, blocks until finished.  Note this version of ndt is SLOW on the
     * reference's CPU path (ndt.hpp:65); here the derivative passes run on device. */
    bool match();

 private:
    wm_ctx *ctx;
    int device;
    PCLPointCloudPtr ref, target;
    NDTMatcherParams params;
    bool ref_dirty, target_dirty;
    bool ensureContext();
};

}  // namespace wave

#endif  // WAVE_MATCHING_NDT_HPP
