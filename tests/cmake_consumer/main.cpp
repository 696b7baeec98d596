// The reference's usage, unchanged: construct, setup, match, getResult.
#include <cstdio>

#include <wave/matching/icp.hpp>

int main() {
    wave::ICPMatcherParams params;
    params.res = -1;
    wave::ICPMatcher matcher(params);
    auto ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    auto target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    for (int i = 0; i < 2000; ++i) {
        pcl::PointXYZ p;
        p.x = (float) (i % 50) * 0.1f;
        p.y = (float) (i / 50) * 0.1f;
        p.z = 0.01f * (float) ((i * 7) % 13);
        ref->points.push_back(p);
        p.x += 0.05f;
        target->points.push_back(p);
    }
    matcher.setup(ref, target);
    const bool ok = matcher.match();  // false without a GPU: the library has no CPU fallback
    std::printf("match=%d tx=%.4f\n", ok ? 1 : 0, matcher.getResult().translation()(0));
    return 0;
}
