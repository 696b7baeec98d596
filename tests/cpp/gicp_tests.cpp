// The reference's wave_matching/tests/gicp_tests.cpp re-expressed against the drop-in API.
#include "wave/matching/gicp.hpp"
#include "wave_test.hpp"

namespace wave {

static const std::string TEST_SCAN = wave_test_path("tests/golden/testscan.pcd");
static const std::string TEST_CONFIG = wave_test_path("tests/golden/config/gicp.yaml");

class GICPTest : public testing::Test {
 protected:
    GICPTest() : matcher(nullptr) {}
    virtual ~GICPTest() {
        if (this->matcher) delete this->matcher;
    }
    virtual void SetUp() {
        this->ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        this->target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        pcl::io::loadPCDFile(TEST_SCAN, *(this->ref));
    }
    void initMatcher(const GICPMatcherParams params, const Affine3 perturb) {
        this->matcher = new GICPMatcher(params);
        pcl::transformPointCloud(*(this->ref), *(this->target), perturb);
        this->matcher->setup(this->ref, this->target);
    }
    pcl::PointCloud<pcl::PointXYZ>::Ptr ref, target;
    GICPMatcher *matcher;
    const float threshold = 0.1;
};

TEST(GICPTests, initialization) {
    GICPMatcher matcher{GICPMatcherParams()};
    GICPMatcherParams from_file(TEST_CONFIG);
    EXPECT_EQ(from_file.corr_rand, 10);  // file values are discarded (gicp.cpp:8-13)
}

// Zero displacement without downsampling
TEST_F(GICPTest, fullResNullMatch) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0, 0, 0;
    GICPMatcherParams params(TEST_CONFIG);
    params.res = -1;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Zero displacement using voxel downsampling
TEST_F(GICPTest, nullDisplacement) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0, 0, 0;
    GICPMatcherParams params(TEST_CONFIG);
    params.res = 0.05f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Small displacement using voxel downsampling
TEST_F(GICPTest, smallDisplacement) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;
    GICPMatcherParams params(TEST_CONFIG);
    params.res = 0.05f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

}  // namespace wave
