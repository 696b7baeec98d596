// wave::Matcher<T> -- same public surface as the reference's
// wave_matching/include/wave/matching/matcher.hpp:23-99 (getResult by value,
// getInfo by const reference, getRes, setRef/setTarget pure virtual, setup,
// match() defaulting to false, estimateInfo() defaulting to identity).
#ifndef WAVE_MATCHING_MATCHER_HPP
#define WAVE_MATCHING_MATCHER_HPP

#include "wave/utils/utils.hpp"

namespace wave {

template <typename T>
class Matcher {
 public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    /** @param res resolution (voxel leaf) used when down-sampling; matcher.hpp:32 */
    Matcher(float res) : resolution(res) {}
    /** No down-sampling (resolution = -1); matcher.hpp:36-38 */
    Matcher() {
        resolution = -1;
    }
    virtual ~Matcher() {}

    const Eigen::Affine3d getResult() {
        return this->result;
    };
    const Mat6 &getInfo() {
        return this->information;
    };
    float getRes() {
        return this->resolution;
    };

    virtual void setRef(const T &ref) = 0;
    virtual void setTarget(const T &target) = 0;
    void setup(const T &ref, const T &target) {
        this->setRef(ref);
        this->setTarget(target);
    };

    /** Actually performs the match. Any heavy processing is done here.
     * @returns true if match was successful, false otherwise */
    virtual bool match() {
        return 0;
    }

    virtual void estimateInfo() {
        this->information = Mat6::Identity(6, 6);
    }

 protected:
    float resolution;
    /** transformation calculated by the scan registration algorithm: maps ref -> target
     * (what the reference's tests assert, tests/icp_tests.cpp:31-32,59) */
    Affine3 result;
    Mat6 information;
};

}  // namespace wave

#endif  // WAVE_MATCHING_MATCHER_HPP
