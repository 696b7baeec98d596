// wave::NDTMatcher on the MI355X back end.
//
// setTarget() uploads the cloud and builds the voxel model at once (as ndt.cpp:53-56 ->
// setInputTarget does: per-voxel means and conditioned inverse covariances in a device table); the
// model is kept across match() calls until the next setTarget().  match() uploads the source and
// runs wm_ndt_align: Newton iterations whose score / gradient / Hessian passes are device
// reductions over (point, neighbouring voxel) pairs, with the More-Thuente step search PCL's code
// contains (setPcl18StepRule(true): the undamped rule PCL 1.8 actually executes).
//
// Kept from the reference (wave_matching/include/wave/matching/ndt.hpp:33-85, src/ndt.cpp): the
// parameter struct including the const `min_res` floor and the integer `step_size`, the clamp +
// LOG_ERROR when res < min_res (ndt.cpp:23-26), the class surface.
#ifndef WAVE_MATCHING_NDT_HPP
#define WAVE_MATCHING_NDT_HPP

#include <string>

#include <utility>
#include <vector>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"

struct wm_ctx;

namespace wave {

struct NDTMatcherParams {
    NDTMatcherParams(const std::string &config_path);
    NDTMatcherParams(){};

    int step_size = 3;            // longest Newton step, cloud units
    int max_iter = 100;
    double t_eps = 1e-8;          // transformation epsilon
    float res = 5;                // voxel edge of the target model
    const float min_res = 0.05f;  // smaller edges are refused
};

class NDTMatcher : public Matcher<PCLPointCloudPtr> {
 public:
    explicit NDTMatcher(NDTMatcherParams params1);
    // HIP device of the matchers the calling thread constructs from now on (see ICPMatcher::setThreadDevice)
    static void setThreadDevice(int device);
    NDTMatcher(const NDTMatcher &other);
    NDTMatcher &operator=(const NDTMatcher &other);
    ~NDTMatcher();

    // As in the reference (ndt.cpp:48-56): setTarget builds the voxel model of the target at once (PCL's
    // setInputTarget does), and later match() calls on the same target reuse it; the ref handle is
    // kept and the cloud is read by match().
    void setRef(const PCLPointCloudPtr &ref);
    void setTarget(const PCLPointCloudPtr &target);
    bool match();  // blocks; true when the Newton iteration converged

    // Many pairs in ONE device launch (wm_ndt_batch_match: one registration per compute unit, the target's voxel
    // model and the whole of align inside the kernel) -- what wave::MultiMatcher<NDTMatcher> hands its workers when
    // several pairs are queued.  out[k] = {match() result, getResult(), getInfo()} as the worker loop would have read
    // them after pair k; a failed match leaves the transform of the pair before it.
    struct BatchOutcome {
        EIGEN_MAKE_ALIGNED_OPERATOR_NEW
        bool matched;
        Eigen::Affine3d transform;
        Mat6 info;
    };
    typedef std::vector<BatchOutcome, Eigen::aligned_allocator<BatchOutcome>> BatchOutcomes;
    bool batchable(const PCLPointCloudPtr &ref, const PCLPointCloudPtr &target) const;  // clouds of at most 200 000 points
    bool matchBatch(const std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> &pairs, BatchOutcomes &out);

    // PCL 1.8's NDT initialises its More-Thuente loop flag so that the loop body never runs (the step
    // is Newton's, its norm clamped to step_size); later PCL releases run the search.  The reference
    // asks for "PCL 1.8" without pinning a patch level, and on its own smallDisplacement test the
    // clamped-only variant wanders off, so the search runs by default here.  true (or env
    // WAVE_NDT_PCL18_STEP_RULE=1) selects the literal 1.8 behaviour for matchers constructed afterwards.
    static void setPcl18StepRule(bool on);

 private:
    wm_ctx *ctx;
    int device;
    PCLPointCloudPtr ref, target;
    NDTMatcherParams params;
    bool pcl18_step_rule;
    bool target_on_device;  // the target and its voxel model are in the context (setTarget put them there)
    bool ensureContext();
};

}  // namespace wave

#endif  // WAVE_MATCHING_NDT_HPP
