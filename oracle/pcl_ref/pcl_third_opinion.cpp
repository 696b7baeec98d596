// ORACLE / TEST INFRASTRUCTURE -- optional third opinion.
//
// Builds ONLY where a system PCL >= 1.8 is installed (oracle/pcl_ref/Makefile probes pkg-config;
// this image has none, and absence is not an error): the three PCL registration classes configured
// exactly as libwave's wrappers configure them, run on two binary float32 XYZ clouds, final 4x4
// printed as 16 numbers.  tests/test_pcl_third_opinion.py feeds it the reference fixture and
// compares the HIP path and the C oracle with what REAL PCL returns.
//
//   pcl_third_opinion icp|gicp|ndt ref.f32 target.f32 [res]
//
// Configuration sources: wave_matching/src/icp.cpp:47-50 (max_corr 3, max_iter 100, t_eps 1e-8,
// fit_eps 1e-2), gicp.cpp:31-34 (k 10, max_iter 100, r_eps 1e-8, fit_eps 1e-2), ndt.cpp:30-33
// (t_eps 1e-8, step 3, res, max_iter 100) -- the defaults of the reference's parameter structs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/gicp.h>
#include <pcl/registration/icp.h>
#include <pcl/registration/ndt.h>

typedef pcl::PointCloud<pcl::PointXYZ> Cloud;

static Cloud::Ptr load(const char *path) {
    Cloud::Ptr c(new Cloud);
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        std::perror(path);
        std::exit(2);
    }
    float p[3];
    while (std::fread(p, sizeof(float), 3, f) == 3) c->push_back(pcl::PointXYZ(p[0], p[1], p[2]));
    std::fclose(f);
    return c;
}

template <class Reg>
static int run(Reg &reg, const Cloud::Ptr &ref, const Cloud::Ptr &target) {
    Cloud out;
    reg.setInputSource(ref);
    reg.setInputTarget(target);
    reg.align(out);
    const Eigen::Matrix4f T = reg.getFinalTransformation();
    std::printf("%d", reg.hasConverged() ? 1 : 0);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) std::printf(" %.9g", (double) T(r, c));
    std::printf("\n");
    return reg.hasConverged() ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s icp|gicp|ndt ref.f32 target.f32 [res]\n", argv[0]);
        return 2;
    }
    const Cloud::Ptr ref = load(argv[2]), target = load(argv[3]);
    const double res = argc > 4 ? std::atof(argv[4]) : 0.3;
    if (!std::strcmp(argv[1], "icp")) {
        pcl::IterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> icp;
        icp.setMaxCorrespondenceDistance(3);
        icp.setMaximumIterations(100);
        icp.setTransformationEpsilon(1e-8);
        icp.setEuclideanFitnessEpsilon(1e-2);
        return run(icp, ref, target);
    }
    if (!std::strcmp(argv[1], "gicp")) {
        pcl::GeneralizedIterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> gicp;
        gicp.setCorrespondenceRandomness(10);
        gicp.setMaximumIterations(100);
        gicp.setRotationEpsilon(1e-8);
        gicp.setEuclideanFitnessEpsilon(1e-2);
        return run(gicp, ref, target);
    }
    pcl::NormalDistributionsTransform<pcl::PointXYZ, pcl::PointXYZ> ndt;
    ndt.setTransformationEpsilon(1e-8);
    ndt.setStepSize(3);
    ndt.setResolution((float) res);
    ndt.setMaximumIterations(100);
    return run(ndt, ref, target);
}
