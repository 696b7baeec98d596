// The reference's wave_matching/tests/ndt_tests.cpp re-expressed against the drop-in API.
#include "wave/matching/ndt.hpp"
#include "wave_test.hpp"

namespace wave {

static const std::string TEST_SCAN = wave_test_path("tests/golden/testscan.pcd");
static const std::string TEST_CONFIG = wave_test_path("tests/golden/config/ndt.yaml");

class NDTTest : public testing::Test {
 protected:
    NDTTest() : matcher(nullptr) {}
    virtual ~NDTTest() {
        if (this->matcher) delete this->matcher;
    }
    virtual void SetUp() {
        this->ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        this->target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        pcl::io::loadPCDFile(TEST_SCAN, *(this->ref));
    }
    void initMatcher(const NDTMatcherParams params, const Affine3 perturb) {
        this->matcher = new NDTMatcher(params);
        pcl::transformPointCloud(*(this->ref), *(this->target), perturb);
        this->matcher->setup(this->ref, this->target);
    }
    pcl::PointCloud<pcl::PointXYZ>::Ptr ref, target;
    NDTMatcher *matcher;
    const float threshold = 0.12;
};

TEST(NDTTests, initialization) {
    NDTMatcher matcher{NDTMatcherParams()};
    NDTMatcherParams tiny;
    tiny.res = 0.001f;  // clamped to min_res with a LOG_ERROR (ndt.cpp:23-26)
    NDTMatcher clamped(tiny);
    EXPECT_TRUE(clamped.getRes() >= 0.05f);
}

// Zero displacement without downsampling
TEST_F(NDTTest, fullResNullMatch) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0, 0, 0;
    this->initMatcher(NDTMatcherParams(TEST_CONFIG), perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Zero displacement using resolution set by constructor
TEST_F(NDTTest, nullDisplacement) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0, 0, 0;
    NDTMatcherParams params(TEST_CONFIG);
    params.res = 0.1f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Small displacement using resolution set by constructor
TEST_F(NDTTest, smallDisplacement) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;
    NDTMatcherParams params(TEST_CONFIG);
    params.res = 0.3f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

}  // namespace wave
