// C++ acceptance program for the drop-in wave:: matchers (libwave_matching.so over the C ABI).
//
// WHAT is checked is the reference's own test strategy (SURVEY.md section 4:
// wave_matching/tests/{icp,gicp,ndt,multi_matcher}_tests.cpp): load the fixture scan, make the
// target a rigidly shifted copy, register, and require |T - T_true|_F below 0.1 (0.12 for NDT);
// information(0,0) > 0; LUM and LUMold within 0.01 of each other on a jittered copy.  HOW it is
// written is this repository's: the registration cases are rows of a table run by one generic
// routine, and the checker is forty lines instead of googletest (an empty submodule in the
// reference tree, not installed here).
//
//   usage: wave_matching_tests [substring-filter]      (env WAVE_TEST_ROOT = repository root)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "wave/matching/gicp.hpp"
#include "wave/matching/icp.hpp"
#include "wave/matching/multi_matcher.hpp"
#include "wave/matching/ndt.hpp"

namespace {

// ------------------------------------------------------------------ checker
int g_checks_failed = 0;

void expect(bool ok, const char *what, const char *file, int line) {
    if (ok) return;
    ++g_checks_failed;
    std::printf("%s:%d: Failure: %s\n", file, line, what);
}
#define EXPECT(cond) expect((cond), #cond, __FILE__, __LINE__)

struct NamedCase {
    std::string name;
    std::function<void()> body;
};
std::vector<NamedCase> &cases() {
    static std::vector<NamedCase> all;
    return all;
}
struct AddCase {
    AddCase(const std::string &name, std::function<void()> body) { cases().push_back({name, std::move(body)}); }
};

std::string repoPath(const std::string &rel) {
    const char *root = std::getenv("WAVE_TEST_ROOT");
    return std::string(root ? root : ".") + "/" + rel;
}
const std::string kScan = "tests/golden/testscan.pcd";
std::string configOf(const char *which) { return repoPath(std::string("tests/golden/config/") + which + ".yaml"); }

wave::PCLPointCloudPtr loadScan() {
    auto cloud = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    pcl::io::loadPCDFile(repoPath(kScan), *cloud);
    return cloud;
}
wave::PCLPointCloudPtr shifted(const wave::PCLPointCloudPtr &in, const wave::Affine3 &by) {
    auto out = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    pcl::transformPointCloud(*in, *out, by);
    return out;
}
wave::Affine3 translationX(double dx) {
    wave::Affine3 t = wave::Affine3::Identity();
    t.translation() << dx, 0, 0;
    return t;
}
double distanceTo(const wave::Affine3 &truth, const Eigen::Affine3d &got) { return (got.matrix() - truth.matrix()).norm(); }

// ------------------------------------------------------- registration table
// One row = "register the scan against itself shifted by dx along x with this voxel edge".
enum class Kind { ICP, GICP, NDT };
struct Row {
    const char *name;
    Kind kind;
    float res;          // voxel edge / NDT cell; 0 = leave what the config file says
    int multiscale;     // ICP only; -1 = leave
    double dx;
    double tolerance;   // on |T - T_true|_F
    bool want_info;     // ICP only: also require information(0,0) > 0
};
const Row kRows[] = {
    // icp_tests.cpp:49-149
    {"ICPTest.fullResNullMatch", Kind::ICP, -1.f, -1, 0.0, 0.1, false},
    {"ICPTest.nullDisplacement", Kind::ICP, 0.05f, -1, 0.0, 0.1, false},
    {"ICPTest.smallDisplacement", Kind::ICP, 0.05f, -1, 0.2, 0.1, false},
    {"ICPTest.smallinfo", Kind::ICP, 0.05f, -1, 0.2, 0.1, true},
    {"ICPTest.multiscale", Kind::ICP, 0.1f, 3, 0.2, 0.1, false},
    // gicp_tests.cpp:47-101
    {"GICPTest.fullResNullMatch", Kind::GICP, -1.f, -1, 0.0, 0.1, false},
    {"GICPTest.nullDisplacement", Kind::GICP, 0.05f, -1, 0.0, 0.1, false},
    {"GICPTest.smallDisplacement", Kind::GICP, 0.05f, -1, 0.2, 0.1, false},
    // ndt_tests.cpp:48-103 (threshold 0.12 there)
    {"NDTTest.fullResNullMatch", Kind::NDT, 0.f, -1, 0.0, 0.12, false},
    {"NDTTest.nullDisplacement", Kind::NDT, 0.1f, -1, 0.0, 0.12, false},
    {"NDTTest.smallDisplacement", Kind::NDT, 0.3f, -1, 0.2, 0.12, false},
};

template <class MatcherT>
void registerAndCheck(MatcherT &m, const Row &row) {
    const auto ref = loadScan();
    const wave::Affine3 truth = translationX(row.dx);
    m.setup(ref, shifted(ref, truth));
    const bool ok = m.match();
    EXPECT(ok);
    const double err = distanceTo(truth, m.getResult());
    if (!(err < row.tolerance)) std::printf("  |T - T_true| = %g, allowed %g\n", err, row.tolerance);
    EXPECT(err < row.tolerance);
}

void runRow(const Row &row) {
    switch (row.kind) {
        case Kind::ICP: {
            wave::ICPMatcherParams p(configOf("icp"));
            p.res = row.res;
            if (row.multiscale >= 0) p.multiscale_steps = row.multiscale;
            wave::ICPMatcher m(p);
            registerAndCheck(m, row);
            if (row.want_info) {
                m.estimateInfo();
                EXPECT(m.getInfo()(0, 0) > 0);
            }
            break;
        }
        case Kind::GICP: {
            wave::GICPMatcherParams p(configOf("gicp"));
            p.res = row.res;
            wave::GICPMatcher m(p);
            registerAndCheck(m, row);
            break;
        }
        case Kind::NDT: {
            wave::NDTMatcherParams p(configOf("ndt"));
            if (row.res > 0) p.res = row.res;
            wave::NDTMatcher m(p);
            registerAndCheck(m, row);
            break;
        }
    }
}

struct RegisterRows {
    RegisterRows() {
        for (const Row &row : kRows) cases().push_back({row.name, [&row] { runRow(row); }});
    }
} g_register_rows;

// --------------------------------------------------- construction / parameters (CPU only)
bool throwsOnLoad(const std::string &path) {
    try {
        wave::ICPMatcherParams p(path);
        (void) p;
    } catch (const std::runtime_error &) {
        return true;
    }
    return false;
}

AddCase c_icp_init("ICPTests.initialization", [] {
    wave::ICPMatcher m{wave::ICPMatcherParams()};
    EXPECT(m.getRes() > 0);  // default voxel edge 0.1 (icp.hpp:59)
});

AddCase c_icp_yaml("ICPTests.paramsFromYaml", [] {
    wave::ICPMatcherParams p(configOf("icp"));
    EXPECT(p.max_corr == 3.0);
    EXPECT(p.max_iter == 100);
    EXPECT(p.multiscale_steps == 0);
    EXPECT(p.fit_eps == 1e-2);                     // the loader never reads it (icp.cpp:9-16)
    EXPECT(throwsOnLoad(repoPath("tests/golden/config/no_such_file.yaml")));  // icp.cpp:18-20
    EXPECT(throwsOnLoad(configOf("ndt")));         // a file without ICP's keys
});

AddCase c_gicp_init("GICPTests.initialization", [] {
    wave::GICPMatcher m{wave::GICPMatcherParams()};
    wave::GICPMatcherParams from_file(configOf("gicp"));
    EXPECT(from_file.corr_rand == 10);  // parsed, then discarded (gicp.cpp:8-13)
});

AddCase c_ndt_init("NDTTests.initialization", [] {
    wave::NDTMatcher m{wave::NDTMatcherParams()};
    wave::NDTMatcherParams too_fine;
    too_fine.res = 0.001f;  // below min_res: clamped, with a LOG_ERROR (ndt.cpp:23-26)
    wave::NDTMatcher clamped(too_fine);
    EXPECT(clamped.getRes() >= 0.05f);
});

// ------------------------------------------------------------ information estimators
// icp_tests.cpp:151-196: LUM against LUMold on a target jittered by +-0.3 per coordinate
// (an exact copy would make the information infinite)
AddCase c_lum("ICPTests.lumvslum", [] {
    const auto ref = loadScan();
    auto target = shifted(ref, translationX(0.2));
    std::default_random_engine engine;
    std::uniform_real_distribution<double> jitter(-0.3, 0.3);
    for (auto &pt : target->points) {
        pt.x += jitter(engine);
        pt.y += jitter(engine);
        pt.z += jitter(engine);
    }
    wave::ICPMatcherParams p(configOf("icp"));
    p.res = 0.05f;
    wave::Mat6 info[2];
    const wave::ICPMatcherParams::covar_method which[2] = {wave::ICPMatcherParams::LUMold,
                                                           wave::ICPMatcherParams::LUM};
    for (int k = 0; k < 2; ++k) {
        p.covar_estimator = which[k];
        wave::ICPMatcher m(p);
        m.setup(ref, target);
        m.match();
        m.estimateInfo();
        info[k] = m.getInfo();
    }
    EXPECT(info[0](0, 0) > 0);
    EXPECT((info[0] - info[1]).norm() < 0.01);
});

// not in the reference: a failed match() must leave the previous result alone (icp.cpp:132)
AddCase c_keep("ICPTest.failedMatchKeepsResult", [] {
    const auto ref = loadScan();
    wave::ICPMatcherParams p(configOf("icp"));
    p.res = -1;
    wave::ICPMatcher m(p);
    auto target = shifted(ref, translationX(0.2));
    m.setup(ref, target);
    EXPECT(m.match());
    const auto before = m.getResult().matrix();
    *target = *shifted(ref, translationX(500.0));  // nothing within max_corr any more
    EXPECT(!m.match());
    EXPECT((m.getResult().matrix() - before).norm() < 1e-15);
});

// ---- call-order semantics of the reference wrappers
// gicp.cpp:37-55: with res > 0, setRef / setTarget filter the cloud WHEN CALLED and register the copy
AddCase c_gicp_snapshot("GICPTest.setRefSnapshotsTheFilteredCloud", [] {
    const auto scan = loadScan();
    wave::GICPMatcherParams p;  // res = 0.1
    auto ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>(*scan);
    auto target = shifted(scan, translationX(0.2));
    wave::GICPMatcher m(p);
    m.setup(ref, target);
    *ref = *shifted(scan, translationX(400.0));     // the matcher holds its own filtered copies
    *target = *shifted(scan, translationX(-400.0));
    EXPECT(m.match());
    EXPECT(distanceTo(translationX(0.2), m.getResult()) < 0.1);
    // assignment: the assigned-to matcher releases its context and rebuilds from the handles
    wave::GICPMatcher other(p);
    other = m;
    *ref = *scan;
    *target = *shifted(scan, translationX(0.2));
    other.setup(ref, target);
    EXPECT(other.match());
    EXPECT(distanceTo(translationX(0.2), other.getResult()) < 0.1);
});

// The objective of GICP's minimisations (not in the reference): the default is the reference's algorithm -- PCL's per-pair
// objective --, the statistics form is an explicit opt-in per matcher, copies carry the choice, and both register the
// reference's test pair; the opt-in's result lies within the documented 1e-4 m of the default's on this sharp pair.
AddCase c_gicp_objective("GICPTest.objectiveIsPclsUnlessAskedOtherwise", [] {
    const auto scan = loadScan();
    wave::GICPMatcherParams p;
    p.res = 0.05f;
    auto target = shifted(scan, translationX(0.2));
    wave::GICPMatcher a(p), b(p);
    EXPECT(a.getObjective() == wave::GICPMatcher::Objective::PclSums);   // (unless WAVE_GICP_OBJECTIVE=statistics is in the environment)
    b.setObjective(wave::GICPMatcher::Objective::Statistics);
    EXPECT(b.getObjective() == wave::GICPMatcher::Objective::Statistics);
    wave::GICPMatcher c(b);                                              // a copy (MultiMatcher makes them) keeps the choice
    EXPECT(c.getObjective() == wave::GICPMatcher::Objective::Statistics);
    a.setup(scan, target);
    b.setup(scan, target);
    EXPECT(a.match());
    EXPECT(b.match());
    EXPECT(distanceTo(translationX(0.2), a.getResult()) < 0.1);
    EXPECT(distanceTo(translationX(0.2), b.getResult()) < 0.1);
    EXPECT((a.getResult().translation() - b.getResult().translation()).norm() < 1e-4);
    b.setObjective(wave::GICPMatcher::Objective::PclSums);               // ... and back: the same matrix as `a`, bit for bit
    b.setup(scan, target);
    EXPECT(b.match());
    EXPECT((a.getResult().matrix() - b.getResult().matrix()).norm() == 0.0);
});

// ndt.cpp:53-56: setTarget builds the voxel model at once; match() reads the ref (aliased) each time
AddCase c_ndt_eager("NDTTest.setTargetBuildsTheModelAtOnce", [] {
    const auto scan = loadScan();
    wave::NDTMatcherParams p(configOf("ndt"));
    p.res = 0.3f;
    auto ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>(*scan);
    auto target = shifted(scan, translationX(0.2));
    wave::NDTMatcher m(p);
    m.setup(ref, target);
    *target = *shifted(scan, translationX(300.0));  // too late: the model was built from the first content
    EXPECT(m.match());
    EXPECT(distanceTo(translationX(0.2), m.getResult()) < 0.12);
    const auto first = m.getResult().matrix();
    EXPECT(m.match());                               // same target, same model, same answer
    EXPECT((m.getResult().matrix() - first).norm() < 1e-12);
    wave::NDTMatcher copy(p);
    copy = m;                                        // handles only: reads the (now moved) target itself
    *target = *shifted(scan, translationX(0.2));
    copy.setup(ref, target);
    EXPECT(copy.match());
    EXPECT(distanceTo(translationX(0.2), copy.getResult()) < 0.12);
});

// One registration spread over several ranks (ICPMatcher::setDevices; naming device 0 twice or
// four times runs the ranks on it with the library's host-side exchange): the single-GPU answer
AddCase c_icp_devices("ICPTest.setDevicesShardsOneRegistration", [] {
    const auto ref = loadScan();
    auto target = shifted(ref, translationX(0.2));
    wave::ICPMatcherParams p(configOf("icp"));
    p.res = -1;
    wave::ICPMatcher one(p), many(p);
    one.setup(ref, target);
    EXPECT(one.match());
    for (int ranks : {2, 4}) {
        many.setDevices(std::vector<int>((size_t) ranks, 0));
        many.setup(ref, target);
        EXPECT(many.match());
        EXPECT((many.getResult().matrix() - one.getResult().matrix()).norm() < 1e-6);
    }
    many.setDevices({0});
    many.setup(ref, target);
    EXPECT(many.match());
    EXPECT((many.getResult().matrix() - one.getResult().matrix()).norm() < 1e-12);
});

// ... with the reference's DEFAULT parameters (voxel filter + three coarser scales, icp.hpp:54,59) and
// estimateInfo() after it, as MultiMatcher's worker loop calls it (impl/multi_matcher_impl.hpp:48)
AddCase c_icp_devices_default("ICPTest.setDevicesDefaultParametersAndInfo", [] {
    const auto ref = loadScan();
    auto target = shifted(ref, translationX(0.2));
    wave::ICPMatcherParams p;  // defaults: res 0.1, multiscale_steps 3, LUM
    wave::ICPMatcher one(p), many(p);
    one.setup(ref, target);
    EXPECT(one.match());
    one.estimateInfo();
    many.setDevices({0, 0, 0});
    many.setup(ref, target);
    EXPECT(many.match());
    EXPECT((many.getResult().matrix() - one.getResult().matrix()).norm() < 1e-6);
    many.estimateInfo();
    EXPECT(many.getInfo()(0, 0) > 0);
    EXPECT((many.getInfo() - one.getInfo()).norm() < 1e-4 * one.getInfo().norm());
    // a matchBatch() leaves nothing behind for the estimators: information keeps its value
    many.setDevices({});
    wave::ICPMatcher::BatchOutcomes out;
    std::vector<std::pair<wave::PCLPointCloudPtr, wave::PCLPointCloudPtr>> pairs(2, std::make_pair(ref, target));
    EXPECT(many.matchBatch(pairs, out));
    many.estimateInfo();
    EXPECT((many.getInfo() - out.back().info).norm() == 0);
});

// ---------------------------------------------------------------------- MultiMatcher
wave::ICPMatcherParams singleScale() {
    wave::ICPMatcherParams p;
    p.res = 0.1f;
    p.multiscale_steps = 0;
    return p;
}

AddCase c_multi_init("MultiTests.initialization", [] {
    wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams> pool(3);  // starts and joins cleanly
});

// multi_matcher_tests.cpp:31-45: eight identical pairs through four workers; plus the
// getResult() the reference declares without defining
AddCase c_multi_run("MultiTest.simultaneousmatching", [] {
    wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams> pool(4, 10, singleScale());
    const auto scan = loadScan();
    std::vector<wave::PCLPointCloudPtr> copies;
    for (int k = 0; k < 9; ++k) copies.push_back(boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>(*scan));
    for (int k = 0; k < 8; ++k) pool.insert(k, copies[k], copies[k + 1]);
    while (!pool.done()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    std::set<int> seen;
    int id = -1;
    Eigen::Affine3d T;
    wave::Mat6 info;
    while (pool.getResult(&id, &T, &info)) {
        seen.insert(id);
        EXPECT(distanceTo(wave::Affine3::Identity(), T) < 1e-6);
    }
    EXPECT(seen.size() == 8u);
    EXPECT(!pool.getResult(&id, &T, &info));
});

// The batched path under the pool (ICPMatcher::matchBatch, wm_icp_batch_match): full-resolution
// matchers on small clouds take everything that is queued into one launch.  Every pair must come out
// as a matcher used on that pair alone gives it: match() + estimateInfo() of the worker loop
// (impl/multi_matcher_impl.hpp:45-53).
wave::PCLPointCloudPtr subsample(const wave::PCLPointCloudPtr &c, size_t every, float dx) {
    auto out = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    for (size_t i = 0; i < c->points.size(); i += every) {
        pcl::PointXYZ p = c->points[i];
        p.x += dx;
        out->points.push_back(p);
    }
    out->width = (uint32_t) out->points.size();
    out->height = 1;
    return out;
}

// getResult() as the reference DOCUMENTS it (multi_matcher.hpp:64-77; it declares the method and never defines it):
// "Will block until a result is ready if the output buffer is empty but there are matches pending ... false if the
// output queue is empty and there are no matches pending".  Insert N pairs, then `while (getResult(...))` collects
// exactly N -- without polling done(), and with pairs of both kinds (batched launches and one-by-one registrations).
AddCase c_multi_blocking("MultiTest.getResultBlocksWhilePairsArePending", [] {
    wave::ICPMatcherParams p;
    p.res = -1;
    p.multiscale_steps = 0;
    wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams> pool(3, 10, p);
    const auto scan = loadScan();
    std::vector<wave::PCLPointCloudPtr> small;  // (these batch: full resolution, a few thousand points)
    for (int k = 0; k < 4; ++k) small.push_back(subsample(scan, 9 + (size_t) k, 0.f));
    const int kPairs = 24;
    for (int k = 0; k < kPairs; ++k) {
        if (k % 6 == 5) pool.insert(k, scan, scan);  // (55k points: beyond the batched path, one by one)
        else pool.insert(k, small[(size_t) k % 4], small[(size_t) k % 4]);
    }
    std::set<int> seen;
    int id = -1;
    Eigen::Affine3d T;
    wave::Mat6 info;
    while (pool.getResult(&id, &T, &info)) {
        seen.insert(id);
        EXPECT(distanceTo(wave::Affine3::Identity(), T) < 1e-6);
    }
    EXPECT(seen.size() == (size_t) kPairs);
    EXPECT(pool.done());
    EXPECT(!pool.getResult(&id, &T, &info));  // nothing pending, nothing waiting: false at once
});

AddCase c_batch_direct("ICPTest.matchBatchEqualsOneByOne", [] {
    const auto scan = loadScan();
    wave::ICPMatcherParams p;
    p.res = -1;
    std::vector<std::pair<wave::PCLPointCloudPtr, wave::PCLPointCloudPtr>> pairs;
    for (int k = 0; k < 5; ++k) pairs.emplace_back(subsample(scan, 8 + k, 0.f), subsample(scan, 7 + k, 0.05f * (float) k));
    pairs.emplace_back(subsample(scan, 9, 0.f), subsample(scan, 9, 400.f));  // nothing within max_corr: match() fails
    pairs.emplace_back(subsample(scan, 11, 0.f), subsample(scan, 10, 0.1f));
    wave::ICPMatcher batch(p);
    for (const auto &pr : pairs) EXPECT(batch.batchable(pr.first, pr.second));
    EXPECT(!batch.batchable(scan, scan));  // 55k points: the pool registers those one by one (faster there)
    pairs.emplace_back(subsample(scan, 2, 0.f), subsample(scan, 2, 0.2f));  // 27.5k points: the target stays in HBM
    EXPECT(batch.batchable(pairs.back().first, pairs.back().second));
    wave::ICPMatcher::BatchOutcomes got;
    EXPECT(batch.matchBatch(pairs, got));
    EXPECT(got.size() == pairs.size());
    wave::Affine3 last = wave::Affine3::Identity();
    for (size_t k = 0; k < pairs.size() && k < got.size(); ++k) {
        wave::ICPMatcher one(p);  // a fresh matcher: fresh stopping criteria, as every batch item has
        one.setup(pairs[k].first, pairs[k].second);
        const bool ok = one.match();
        one.estimateInfo();
        EXPECT(ok == got[k].matched);
        EXPECT(ok == (k != 5));  // (pair 5: nothing within max_corr)
        if (ok) {
            EXPECT(distanceTo(one.getResult(), got[k].transform) < 1e-9);
            EXPECT((one.getInfo() - got[k].info).norm() <= 1e-5 * one.getInfo().norm());
            last = got[k].transform;
        } else {
            EXPECT(distanceTo(last, got[k].transform) == 0.0);  // `result` is left alone (icp.cpp:132)
        }
    }
});

// ... and with the reference's DEFAULT parameters (voxel filter 0.1 m, three coarser scales) and its
// test configuration (single scale): the whole batch goes through pcl::VoxelGrid at once, scale by scale
AddCase c_batch_filtered("ICPTest.matchBatchWithVoxelFilterEqualsOneByOne", [] {
    const auto scan = loadScan();
    std::vector<std::pair<wave::PCLPointCloudPtr, wave::PCLPointCloudPtr>> pairs;
    pairs.emplace_back(scan, subsample(scan, 1, 0.2f));
    pairs.emplace_back(subsample(scan, 3, 0.f), subsample(scan, 2, 0.1f));
    pairs.emplace_back(scan, subsample(scan, 1, 600.f));  // fails at the coarsest scale
    pairs.emplace_back(subsample(scan, 2, 0.f), scan);
    for (int variant = 0; variant < 2; ++variant) {
        wave::ICPMatcherParams p;  // defaults: res 0.1, multiscale_steps 3
        if (variant == 1) p = singleScale();
        wave::ICPMatcher batch(p);
        for (const auto &pr : pairs) EXPECT(batch.batchable(pr.first, pr.second));
        wave::ICPMatcher::BatchOutcomes got;
        EXPECT(batch.matchBatch(pairs, got));
        EXPECT(got.size() == pairs.size());
        wave::Affine3 last = wave::Affine3::Identity();
        for (size_t k = 0; k < pairs.size() && k < got.size(); ++k) {
            wave::ICPMatcher one(p);
            one.setup(pairs[k].first, pairs[k].second);
            const bool ok = one.match();
            one.estimateInfo();
            EXPECT(ok == got[k].matched);
            EXPECT(ok == (k != 2));
            if (ok) {
                EXPECT(distanceTo(one.getResult(), got[k].transform) < 1e-7);
                EXPECT((one.getInfo() - got[k].info).norm() <= 1e-4 * one.getInfo().norm());
                last = got[k].transform;
            } else {
                EXPECT(distanceTo(last, got[k].transform) == 0.0);
            }
        }
    }
});

AddCase c_multi_batch("MultiTest.queuedPairsShareOneLaunch", [] {
    const auto scan = loadScan();
    wave::ICPMatcherParams p;
    p.res = -1;
    std::vector<wave::PCLPointCloudPtr> refs, targets;
    for (int k = 0; k < 6; ++k) {
        refs.push_back(subsample(scan, 8 + (size_t) k, 0.f));
        targets.push_back(subsample(scan, 7 + (size_t) k, 0.04f * (float) k));
    }
    std::vector<wave::Affine3, Eigen::aligned_allocator<wave::Affine3>> want;
    std::vector<wave::Mat6, Eigen::aligned_allocator<wave::Mat6>> want_info;
    for (int k = 0; k < 6; ++k) {
        wave::ICPMatcher one(p);
        one.setup(refs[(size_t) k], targets[(size_t) k]);
        EXPECT(one.match());
        one.estimateInfo();
        want.push_back(one.getResult());
        want_info.push_back(one.getInfo());
    }
    wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams> pool(2, 64, p);
    const int jobs = 150;
    for (int j = 0; j < jobs; ++j) pool.insert(j, refs[(size_t) j % 6], targets[(size_t) j % 6]);
    while (!pool.done()) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    std::set<int> seen;
    int id = -1;
    Eigen::Affine3d T;
    wave::Mat6 info;
    while (pool.getResult(&id, &T, &info)) {
        seen.insert(id);
        EXPECT(id >= 0 && id < jobs);
        EXPECT(distanceTo(want[(size_t) id % 6], T) < 1e-9);
        EXPECT((want_info[(size_t) id % 6] - info).norm() <= 1e-5 * info.norm());
    }
    EXPECT(seen.size() == (size_t) jobs);
});

// One pool over the GPUs of a node: the workers are dealt onto the listed devices (here the one
// device this box has, named twice -- the binding goes through the same per-thread default).
AddCase c_multi_devices("MultiTest.workersDealtOverDevices", [] {
    typedef wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams> Pool;
    Pool::setDevices({0, 0});
    EXPECT(Pool::devices().size() == 2u);
    {
        Pool pool(4, 16, singleScale());
        const auto scan = loadScan();
        for (int k = 0; k < 6; ++k) pool.insert(k, scan, scan);
        while (!pool.done()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        int id = -1, seen = 0;
        Eigen::Affine3d T;
        wave::Mat6 info;
        while (pool.getResult(&id, &T, &info)) {
            ++seen;
            EXPECT(distanceTo(wave::Affine3::Identity(), T) < 1e-6);
        }
        EXPECT(seen == 6);
    }
    Pool::setDevices({});
    wave::ICPMatcher::setThreadDevice(-1);
});

// The batched path of GICPMatcher (matchBatch, wm_gicp_batch_match: one registration per compute unit, the whole of
// align in the kernel).  Every pair must come out as a matcher used on that pair alone gives it.
AddCase c_gicp_batch("GICPTest.matchBatchEqualsOneByOne", [] {
    const auto scan = loadScan();
    for (int variant = 0; variant < 2; ++variant) {
        wave::GICPMatcherParams p;  // defaults: res 0.1
        if (variant == 1) p.res = -1;
        std::vector<std::pair<wave::PCLPointCloudPtr, wave::PCLPointCloudPtr>> pairs;
        for (int k = 0; k < 4; ++k) pairs.emplace_back(subsample(scan, 6 + k, 0.f), subsample(scan, 5 + k, 0.05f * (float) k));
        pairs.emplace_back(subsample(scan, 9, 0.f), subsample(scan, 9, 400.f));  // nothing within max_corr: match() fails
        pairs.emplace_back(subsample(scan, 7, 0.f), subsample(scan, 6, 0.1f));
        wave::GICPMatcher batch(p);
        for (const auto &pr : pairs) EXPECT(batch.batchable(pr.first, pr.second));
        wave::GICPMatcher::BatchOutcomes got;
        EXPECT(batch.matchBatch(pairs, got));
        EXPECT(got.size() == pairs.size());
        wave::Affine3 last = wave::Affine3::Identity();
        for (size_t k = 0; k < pairs.size() && k < got.size(); ++k) {
            wave::GICPMatcher one(p);
            one.setup(pairs[k].first, pairs[k].second);
            const bool ok = one.match();
            EXPECT(ok == got[k].matched);
            EXPECT(ok == (k != 4));
            if (ok) {
                EXPECT(distanceTo(one.getResult(), got[k].transform) < 1e-6);
                last = got[k].transform;
            } else {
                EXPECT(distanceTo(last, got[k].transform) == 0.0);  // `result` is left alone (gicp.cpp:59-63)
            }
        }
    }
});

// ... under the pool, at the reference's default queue of ten (multi_matcher.hpp:32-34)
AddCase c_gicp_pool("MultiTest.gicpPairsGoThroughTheBatchedPath", [] {
    wave::GICPMatcherParams p;
    p.res = -1;
    wave::MultiMatcher<wave::GICPMatcher, wave::GICPMatcherParams> pool(4, 10, p);
    const auto scan = loadScan();
    std::vector<wave::PCLPointCloudPtr> refs, targets;
    for (int k = 0; k < 24; ++k) {
        refs.push_back(subsample(scan, 10 + (size_t) (k % 3), 0.f));
        targets.push_back(subsample(scan, 9 + (size_t) (k % 3), 0.02f * (float) (k % 5)));
    }
    for (int k = 0; k < 24; ++k) pool.insert(k, refs[(size_t) k], targets[(size_t) k]);
    while (!pool.done()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    std::set<int> seen;
    int id = -1;
    Eigen::Affine3d T;
    wave::Mat6 info;
    while (pool.getResult(&id, &T, &info)) {
        seen.insert(id);
        wave::GICPMatcher one(p);
        one.setup(refs[(size_t) id], targets[(size_t) id]);
        EXPECT(one.match());
        EXPECT(distanceTo(one.getResult(), T) < 1e-6);
    }
    EXPECT(seen.size() == 24u);
});

// The batched path of NDTMatcher (matchBatch, wm_ndt_batch_match: one registration per compute unit, voxel model and
// align in the kernel) -- directly and under the pool at the reference's default queue of ten.
AddCase c_ndt_batch("NDTTest.matchBatchEqualsOneByOne", [] {
    const auto scan = loadScan();
    wave::NDTMatcherParams p;
    p.res = 2.0f;
    std::vector<std::pair<wave::PCLPointCloudPtr, wave::PCLPointCloudPtr>> pairs;
    for (int k = 0; k < 4; ++k) pairs.emplace_back(subsample(scan, 3 + k, 0.f), subsample(scan, 2 + k, 0.05f * (float) k));
    pairs.emplace_back(subsample(scan, 5, 0.f), subsample(scan, 4, 0.1f));
    wave::NDTMatcher batch(p);
    for (const auto &pr : pairs) EXPECT(batch.batchable(pr.first, pr.second));
    wave::NDTMatcher::BatchOutcomes got;
    EXPECT(batch.matchBatch(pairs, got));
    EXPECT(got.size() == pairs.size());
    for (size_t k = 0; k < pairs.size() && k < got.size(); ++k) {
        wave::NDTMatcher one(p);
        one.setup(pairs[k].first, pairs[k].second);
        const bool ok = one.match();
        EXPECT(ok == got[k].matched);
        if (ok) EXPECT(distanceTo(one.getResult(), got[k].transform) < 1e-6);
    }
});

AddCase c_ndt_pool("MultiTest.ndtPairsGoThroughTheBatchedPath", [] {
    wave::NDTMatcherParams p;
    p.res = 2.0f;
    wave::MultiMatcher<wave::NDTMatcher, wave::NDTMatcherParams> pool(4, 10, p);
    const auto scan = loadScan();
    std::vector<wave::PCLPointCloudPtr> refs, targets;
    for (int k = 0; k < 24; ++k) {
        refs.push_back(subsample(scan, 4 + (size_t) (k % 3), 0.f));
        targets.push_back(subsample(scan, 3 + (size_t) (k % 3), 0.02f * (float) (k % 5)));
    }
    for (int k = 0; k < 24; ++k) pool.insert(k, refs[(size_t) k], targets[(size_t) k]);
    while (!pool.done()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    std::set<int> seen;
    int id = -1;
    Eigen::Affine3d T;
    wave::Mat6 info;
    while (pool.getResult(&id, &T, &info)) {
        seen.insert(id);
        wave::NDTMatcher one(p);
        one.setup(refs[(size_t) id], targets[(size_t) id]);
        EXPECT(one.match());
        EXPECT(distanceTo(one.getResult(), T) < 1e-6);
    }
    EXPECT(seen.size() == 24u);
});

}  // namespace

int main(int argc, char **argv) {
    const std::string filter = argc > 1 ? argv[1] : "";
    int ran = 0, failed = 0;
    for (const auto &c : cases()) {
        if (!filter.empty() && c.name.find(filter) == std::string::npos) continue;
        std::printf("[ RUN      ] %s\n", c.name.c_str());
        std::fflush(stdout);
        const int before = g_checks_failed;
        c.body();
        ++ran;
        const bool ok = g_checks_failed == before;
        failed += !ok;
        std::printf("[%s] %s\n", ok ? "       OK " : "  FAILED  ", c.name.c_str());
    }
    std::printf("[==========] %d tests ran, %d failed\n", ran, failed);
    return failed ? 1 : 0;
}
