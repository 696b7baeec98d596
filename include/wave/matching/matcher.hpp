// Abstract registration interface of the MI355X back end.
//
// A libwave program sees `wave::Matcher<CloudT>` exactly as before: hand over two clouds, call
// match(), read a rigid transform and (optionally) a 6x6 information matrix.  What changes is
// below this interface -- the concrete matchers own a wm_ctx (include/wavematch.h) instead of
// PCL objects.  Member names and call semantics follow the reference so that existing callers
// and MultiMatcher compile unchanged (wave_matching/include/wave/matching/matcher.hpp:23-99).
//
// Conventions worth knowing (none of them invented here):
//   * no initial guess: matchers start from identity; pre-transform `target` if you have one;
//   * getResult() maps `ref` onto `target` (the reference's tests are the authority:
//     tests/icp_tests.cpp:31-32,59);
//   * information is ordered (x, y, z, then the three Euler angles) in the frame of `ref`.
#ifndef WAVE_MATCHING_MATCHER_HPP
#define WAVE_MATCHING_MATCHER_HPP

#include "wave/utils/utils.hpp"

namespace wave {

template <typename T>
class Matcher {
 public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    Matcher() : resolution(-1.0f) {}               // full resolution, no voxel filter
    Matcher(float res) : resolution(res) {}        // voxel edge used before matching
    virtual ~Matcher() = default;

    // -- inputs (cloud handles are kept, not copied)
    virtual void setRef(const T &ref) = 0;
    virtual void setTarget(const T &target) = 0;
    void setup(const T &ref, const T &target) { setRef(ref), setTarget(target); }

    // -- work: the base class has nothing to register, hence "no success"
    virtual bool match() { return false; }
    virtual void estimateInfo() { information = Mat6::Identity(); }

    // -- outputs of the last successful match()
    const Eigen::Affine3d getResult() { return result; }
    const Mat6 &getInfo() { return information; }
    float getRes() { return resolution; }

 protected:
    float resolution;   // voxel edge in cloud units; -1 = none
    Affine3 result;     // ref -> target
    Mat6 information;
};

}  // namespace wave

#endif  // WAVE_MATCHING_MATCHER_HPP
