// The reference's wave_matching/tests/icp_tests.cpp re-expressed against the drop-in
// API: same fixture (testscan.pcd), same YAML config, same perturbations, same
// assertions (|T - T_gt|_F < 0.1, info(0,0) > 0, |LUMold - LUM| < 0.01).
#include <cstdlib>
#include <random>

#include "wave/matching/icp.hpp"
#include "wave_test.hpp"

namespace wave {

static const std::string TEST_SCAN = wave_test_path("tests/golden/testscan.pcd");
static const std::string TEST_CONFIG = wave_test_path("tests/golden/config/icp.yaml");

class ICPTest : public testing::Test {
 protected:
    ICPTest() : matcher(nullptr) {}
    virtual ~ICPTest() {
        if (this->matcher) delete this->matcher;
    }
    virtual void SetUp() {
        this->ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        this->target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        pcl::io::loadPCDFile(TEST_SCAN, *(this->ref));
    }
    void initMatcher(const ICPMatcherParams params, const Affine3 perturb) {
        this->matcher = new ICPMatcher(params);
        pcl::transformPointCloud(*(this->ref), *(this->target), perturb);
        this->matcher->setup(this->ref, this->target);
    }
    pcl::PointCloud<pcl::PointXYZ>::Ptr ref, target;
    ICPMatcher *matcher;
    const float threshold = 0.1;
};

TEST(ICPTests, initialization) {
    ICPMatcher matcher{ICPMatcherParams()};
    EXPECT_TRUE(matcher.getRes() > 0);  // default res = 0.1 (icp.hpp:59)
}

TEST(ICPTests, paramsFromYaml) {
    ICPMatcherParams params(TEST_CONFIG);
    EXPECT_EQ(params.max_corr, 3.0);
    EXPECT_EQ(params.max_iter, 100);
    EXPECT_EQ(params.multiscale_steps, 0);
    EXPECT_EQ(params.fit_eps, 1e-2);  // not read from the file (icp.cpp:9-16)
    bool threw = false;
    try {
        ICPMatcherParams bad(wave_test_path("tests/golden/config/does_not_exist.yaml"));
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT_TRUE(threw);  // icp.cpp:18-20
    threw = false;
    try {  // ndt.yaml lacks ICP's keys -> KeyError -> throw
        ICPMatcherParams bad(wave_test_path("tests/golden/config/ndt.yaml"));
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT_TRUE(threw);
}

// Zero displacement without downsampling
TEST_F(ICPTest, fullResNullMatch) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0, 0, 0;
    ICPMatcherParams params(TEST_CONFIG);
    params.res = -1;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Zero displacement using voxel downsampling
TEST_F(ICPTest, nullDisplacement) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0, 0, 0;
    ICPMatcherParams params(TEST_CONFIG);
    params.res = 0.05f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Small displacement using voxel downsampling
TEST_F(ICPTest, smallDisplacement) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;
    ICPMatcherParams params(TEST_CONFIG);
    params.res = 0.05f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

// Small information using voxel downsampling
TEST_F(ICPTest, smallinfo) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;
    ICPMatcherParams params(TEST_CONFIG);
    params.res = 0.05f;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    matcher->estimateInfo();
    auto info = matcher->getInfo();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_GT(info(0, 0), 0);
    EXPECT_LT(diff, this->threshold);
}

// Small displacement using voxel downsampling and multiscale matching
TEST_F(ICPTest, multiscale) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;
    ICPMatcherParams params(TEST_CONFIG);
    params.res = 0.1f;
    params.multiscale_steps = 3;
    this->initMatcher(params, perturb);
    bool match_success = matcher->match();
    matcher->estimateInfo();
    double diff = (matcher->getResult().matrix() - perturb.matrix()).norm();
    EXPECT_TRUE(match_success);
    EXPECT_LT(diff, this->threshold);
}

TEST(ICPTests, lumvslum) {
    pcl::PointCloud<pcl::PointXYZ>::Ptr ref, target;
    ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    pcl::io::loadPCDFile(TEST_SCAN, *(ref));
    Affine3 perturb;
    double lower_bound = -0.3;
    double upper_bound = 0.3;
    std::uniform_real_distribution<double> unif(lower_bound, upper_bound);
    std::default_random_engine re;
    perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;

    pcl::transformPointCloud(*(ref), *(target), perturb);
    // need to distort one scan or there will be infinite information
    for (size_t i = 0; i < target->size(); i++) {
        target->at(i).x += unif(re);
        target->at(i).y += unif(re);
        target->at(i).z += unif(re);
    }

    ICPMatcherParams params(TEST_CONFIG);
    params.res = 0.05f;
    params.covar_estimator = ICPMatcherParams::covar_method::LUMold;
    ICPMatcher matcher1(params);
    matcher1.setup(ref, target);
    matcher1.match();
    matcher1.estimateInfo();
    auto info1 = matcher1.getInfo();

    params.covar_estimator = ICPMatcherParams::covar_method::LUM;
    ICPMatcher matcher2(params);
    matcher2.setup(ref, target);
    matcher2.match();
    matcher2.estimateInfo();
    auto info2 = matcher2.getInfo();

    double diff = (info1 - info2).norm();
    EXPECT_GT(info1(0, 0), 0);
    EXPECT_LT(diff, 0.01);
}

// not in the reference: match() == false leaves the previous result untouched (icp.cpp:132)
TEST_F(ICPTest, failedMatchKeepsResult) {
    Affine3 perturb = Affine3::Identity();
    perturb.translation() << 0.2, 0, 0;
    ICPMatcherParams params(TEST_CONFIG);
    params.res = -1;
    this->initMatcher(params, perturb);
    EXPECT_TRUE(matcher->match());
    const auto before = matcher->getResult().matrix();
    Affine3 far = Affine3::Identity();
    far.translation() << 500, 0, 0;
    pcl::transformPointCloud(*(this->ref), *(this->target), far);
    EXPECT_FALSE(matcher->match());
    EXPECT_LT((matcher->getResult().matrix() - before).norm(), 1e-15);
}

}  // namespace wave
