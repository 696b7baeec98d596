// wave::GICPMatcher on the MI355X back end.
//
// match() is wm_gicp_match (include/wavematch.h): voxel filter of both clouds, k-nearest-
// neighbour covariances with the (1, 1, epsilon) spectrum, then PCL-GICP's outer loop --
// correspondences, per-pair Mahalanobis matrices, a BFGS minimisation whose objective and
// gradient are device reductions -- until the rotation / translation deltas fall under the
// thresholds.  PCL-GICP defaults libwave never overrides stay as they are (max correspondence
// distance 5, transformation epsilon 5e-4, gicp_epsilon 1e-3, 20 inner iterations).
//
// Kept from the reference (wave_matching/include/wave/matching/gicp.hpp:30-70, src/gicp.cpp):
// the parameter struct, the class surface, and the YAML constructor's quirk of parsing the file
// but keeping the defaults (gicp.cpp:8-13).
#ifndef WAVE_MATCHING_GICP_HPP
#define WAVE_MATCHING_GICP_HPP

#include <string>
#include <utility>
#include <vector>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"

struct wm_ctx;

namespace wave {

struct GICPMatcherParams {
    GICPMatcherParams() {}
    GICPMatcherParams(const std::string &config_path);

    int corr_rand = 10;     // neighbours per covariance estimate
    int max_iter = 100;     // outer iterations
    double r_eps = 1e-8;    // rotation epsilon
    double fit_eps = 1e-2;  // euclidean fitness epsilon
    float res = 0.1;        // voxel edge (<= 0: none)
};

class GICPMatcher : public Matcher<PCLPointCloudPtr> {
 public:
    explicit GICPMatcher(GICPMatcherParams params1);
    // HIP device of the matchers the calling thread constructs from now on (see ICPMatcher::setThreadDevice)
    static void setThreadDevice(int device);
    GICPMatcher(const GICPMatcher &other);  // for MultiMatcher; the copy gets its own context
    GICPMatcher &operator=(const GICPMatcher &other);
    ~GICPMatcher();

    // As in the reference (gicp.cpp:37-55): with res > 0 the cloud is voxel-filtered HERE and the
    // filtered copy is what match() registers (later changes of the caller's cloud are not seen);
    // with res <= 0 the handle is kept and the cloud is read by match().
    void setRef(const PCLPointCloudPtr &ref);
    void setTarget(const PCLPointCloudPtr &target);
    bool match();  // blocks until the registration is done; true when it converged

    // NOT in the reference: how the BFGS minimisations' objective is evaluated.  PclSums (the default) is the
    // reference's algorithm -- PCL's OptimizationFunctorWithIndices summed pair by pair through the float transform at
    // every trial point.  Statistics forms the same objective once per outer iteration as 74 sufficient statistics
    // (2.4x faster at 500k points); its result is NOT PCL's bit for bit: 0 .. 3e-5 m apart on pairs that register
    // sharply, up to 1e-3 m on noisy pairs (the spread PCL's own result has against its summation order).  A matcher
    // that was never told takes env WAVE_GICP_OBJECTIVE=statistics, else PclSums.
    enum class Objective { PclSums, Statistics };
    void setObjective(Objective o);
    Objective getObjective() const;

    // Many pairs in ONE device launch (wm_gicp_batch_match: one registration per compute unit, the whole of
    // align inside the kernel) -- what wave::MultiMatcher<GICPMatcher> hands its workers when several pairs are
    // queued.  out[k] = {match() result, getResult(), getInfo()} as the worker loop would have read them after
    // pair k: setup(ref, target) + match() + estimateInfo(); a failed match leaves the transform of the pair
    // before it, and GICPMatcher has no information estimate (getInfo() stays what it was).
    struct BatchOutcome {
        EIGEN_MAKE_ALIGNED_OPERATOR_NEW
        bool matched;
        Eigen::Affine3d transform;
        Mat6 info;
    };
    typedef std::vector<BatchOutcome, Eigen::aligned_allocator<BatchOutcome>> BatchOutcomes;
    // does the pair go down the batched path?  (clouds of at most 100 000 points as they are, 400 000 when they
    // are voxel-filtered first; bigger ones are registered one at a time, on the whole device)
    bool batchable(const PCLPointCloudPtr &ref, const PCLPointCloudPtr &target) const;
    bool matchBatch(const std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> &pairs, BatchOutcomes &out);

 private:
    wm_ctx *ctx;
    int device;
    PCLPointCloudPtr ref, target;
    GICPMatcherParams params;
    bool ref_on_device, target_on_device;  // res > 0: the filtered snapshot already sits in the context
    int objective;                         // wm_gicp_params::objective, or -1: not told (environment, else PCL's)
    bool ensureContext();
};

}  // namespace wave

#endif  // WAVE_MATCHING_GICP_HPP
