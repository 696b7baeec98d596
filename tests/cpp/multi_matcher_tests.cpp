// The reference's wave_matching/tests/multi_matcher_tests.cpp re-expressed, plus a check
// of getResult() (declared but never defined in the reference, multi_matcher.hpp:73).
#include <chrono>
#include <set>
#include <thread>

#include "wave/matching/icp.hpp"
#include "wave/matching/multi_matcher.hpp"
#include "wave_test.hpp"

namespace wave {

static const std::string TEST_SCAN = wave_test_path("tests/golden/testscan.pcd");

class MultiTest : public testing::Test {
 protected:
    MultiTest() : matcher(4, 10, fullRes()) {}
    static ICPMatcherParams fullRes() {
        ICPMatcherParams p;
        p.res = 0.1f;
        p.multiscale_steps = 0;
        return p;
    }
    virtual void SetUp() {
        this->cld = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        pcl::io::loadPCDFile(TEST_SCAN, *(this->cld));
    }
    pcl::PointCloud<pcl::PointXYZ>::Ptr cld;
    MultiMatcher<ICPMatcher, ICPMatcherParams> matcher;
};

// Tests that threads are created and destroyed properly
TEST(MultiTests, initialization) {
    MultiMatcher<ICPMatcher, ICPMatcherParams> matcher(3);
}

TEST_F(MultiTest, simultaneousmatching) {
    pcl::PointCloud<pcl::PointXYZ>::Ptr dupes[9];
    for (int i = 0; i < 9; i++) {
        dupes[i] = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
        *(dupes[i]) = *(this->cld);
    }
    for (int i = 0; i < 8; i++) {
        this->matcher.insert(i, dupes[i], dupes[i + 1]);
    }
    while (!this->matcher.done()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    std::set<int> ids;
    int id;
    Eigen::Affine3d T;
    Mat6 info;
    while (this->matcher.getResult(&id, &T, &info)) {
        ids.insert(id);
        EXPECT_LT((T.matrix() - Eigen::Affine3d::Identity().matrix()).norm(), 1e-6);
    }
    EXPECT_EQ(ids.size(), 8u);
    EXPECT_FALSE(this->matcher.getResult(&id, &T, &info));
}

}  // namespace wave
