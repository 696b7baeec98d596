// wave::ICPMatcher: parameters in, one wm_icp_match out, wm_icp_info for the information
// matrix.  No registration arithmetic happens on the host.  Reference behaviour followed:
// wave_matching/src/icp.cpp (parameters :6-51, handles :67-73, match :75-133, estimator
// dispatch :135-142) and src/icp_pcl_functions.cpp (LUM / LUMold).
#include "wave/matching/icp.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "shim.hpp"

namespace wave {

void ICPMatcher::setDefaultDevice(int device) { shim::setDefaultDevice(device); }
void ICPMatcher::setThreadDevice(int device) { shim::setThreadDevice(device); }

namespace {
std::atomic<int> &benchForceSetting() {
    static std::atomic<int> n{-1};  // -1: not set (the environment decides)
    return n;
}
int benchForceIterations() {
    const int n = benchForceSetting().load();
    if (n >= 0) return n;
    static const int from_env = [] {
        const char *e = std::getenv("WAVE_ICP_BENCH_FORCE_ITERATIONS");
        const int v = e ? std::max(0, std::atoi(e)) : 0;
        // (a bench-only knob with no counterpart in the reference: say so ONCE when it is active -- a stray variable in
        // a production environment switches PCL's stopping rules off for every ICPMatcher of the process)
        if (v > 0)
            std::fprintf(stderr, "[INFO] WAVE_ICP_BENCH_FORCE_ITERATIONS=%d: every ICPMatcher::match() of this process runs exactly "
                                 "that many iterations (PCL's stopping rules are OFF) -- a benchmark setting\n", v);
        return v;
    }();
    return from_env;
}
}  // namespace
void ICPMatcher::setBenchForceIterations(int n) { benchForceSetting().store(n > 0 ? n : 0); }

ICPMatcherParams::ICPMatcherParams(const std::string &config_path) {
    int estimator = 0;
    // `fit_eps` is deliberately absent: the reference's loader never reads it (icp.cpp:9-16)
    shim::loadYaml(config_path, {{"max_corr", &max_corr},
                                 {"max_iter", &max_iter},
                                 {"t_eps", &t_eps},
                                 {"lidar_ang_covar", &lidar_ang_covar},
                                 {"lidar_lin_covar", &lidar_lin_covar},
                                 {"covar_estimator", &estimator},
                                 {"res", &res},
                                 {"multiscale_steps", &multiscale_steps}});
    if (estimator < covar_method::LUM || estimator > covar_method::LUMold) {
        LOG_ERROR("Invalid covariance estimate method, using LUM");
        estimator = covar_method::LUM;
    }
    covar_estimator = static_cast<covar_method>(estimator);
}

ICPMatcher::ICPMatcher(ICPMatcherParams params1)
    : params(params1), ctx(nullptr), multi(nullptr), device(shim::defaultDevice()), converged(false),
      lastMatch(kNone), ref(shim::emptyCloud()), target(shim::emptyCloud()) {
    resolution = params.res;
}

// Copies share parameters and cloud handles, never device state: like the PCL members of the
// reference class, a context belongs to one object (and is created by the thread that uses it).
ICPMatcher::ICPMatcher(const ICPMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), params(o.params), ctx(nullptr), multi(nullptr), devices(o.devices),
      device(o.device), converged(false), lastMatch(kNone), ref(o.ref), target(o.target) {}

ICPMatcher &ICPMatcher::operator=(const ICPMatcher &o) {
    if (this == &o) return *this;
    shim::release(ctx);
    if (multi) wm_multi_destroy(multi);
    multi = nullptr;
    Matcher<PCLPointCloudPtr>::operator=(o);
    params = o.params;
    devices = o.devices;
    device = o.device;
    converged = false;
    lastMatch = kNone;
    ref = o.ref;
    target = o.target;
    return *this;
}

ICPMatcher::~ICPMatcher() {
    shim::release(ctx);
    if (multi) wm_multi_destroy(multi);
}

void ICPMatcher::setDevices(const std::vector<int> &list) {
    if (multi) wm_multi_destroy(multi);
    multi = nullptr;
    lastMatch = kNone;
    devices = list;
    if (devices.size() == 1) {  // a plain single-device matcher on that device
        if (devices[0] != device) shim::release(ctx);
        device = devices[0];
        devices.clear();
    }
}

bool ICPMatcher::ensureContext() { return shim::acquire(ctx, device); }

void ICPMatcher::setRef(const PCLPointCloudPtr &cloud) { ref = cloud; }

void ICPMatcher::setTarget(const PCLPointCloudPtr &cloud) { target = cloud; }

bool ICPMatcher::match() {
    converged = false;
    lastMatch = kNone;
    wm_icp_params p;
    wm_icp_default_params(&p);
    p.max_corr = params.max_corr;  // setMaxCorrespondenceDistance,  icp.cpp:47
    p.max_iter = params.max_iter;  // setMaximumIterations,          icp.cpp:48
    p.t_eps = params.t_eps;        // setTransformationEpsilon,      icp.cpp:49
    p.fit_eps = params.fit_eps;    // setEuclideanFitnessEpsilon,    icp.cpp:50
    p.carry_state = 1;             // one PCL object per matcher: its criteria remember the last MSE
    if (const int forced = benchForceIterations()) {  // (bench only: see setBenchForceIterations)
        p.force_iterations = forced;
        p.max_iter = std::max(p.max_iter, forced);
    }
    double T[16];
    wm_icp_stats stats;
    if (devices.size() > 1) {  // one registration over several GPUs (setDevices)
        if (!multi) {
            bool repeated = false;
            for (size_t a = 0; a < devices.size(); ++a)
                for (size_t b = a + 1; b < devices.size(); ++b) repeated = repeated || devices[a] == devices[b];
            const int rc0 = wm_multi_create(&multi, devices.data(), (int) devices.size(), repeated ? 1 : 0);
            if (rc0 != WM_OK) {
                LOG_ERROR("ICPMatcher: cannot open the device group (%s)", wm_strerror(rc0));
                multi = nullptr;
                return false;
            }
        }
        const int rcm = wm_multi_icp_match(multi, cloudData(ref), cloudSize(ref), cloudData(target), cloudSize(target),
                                           kCloudStride, &p, params.res, params.multiscale_steps, T, &stats);
        if (!shim::succeeded(rcm, "wm_multi_icp_match", nullptr)) return false;
        shim::toAffine(T, result);
        converged = true;
        lastMatch = kOnMulti;
        return true;
    }
    if (!ensureContext()) return false;
    const int rc = wm_icp_match(ctx, cloudData(ref), cloudSize(ref), cloudData(target), cloudSize(target),
                                kCloudStride, WM_MEM_HOST, &p, params.res, params.multiscale_steps, T,
                                &stats);
    // anything but WM_OK leaves `result` as it was (icp.cpp:132)
    if (!shim::succeeded(rc, "wm_icp_match", ctx)) return false;
    shim::toAffine(T, result);
    converged = true;
    lastMatch = kOnCtx;
    return true;
}

namespace {
void storeInfo(const double in[36], Mat6 &out) {
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) out(r, c) = in[6 * r + c];
}
void warnIfDegenerate(int degenerate) {
    if (degenerate) LOG_ERROR("Covariance matrix calculation was unsuccessful");
}
}  // namespace

// ---- many pairs, one launch
// wm_icp_batch_match takes targets of up to WM_BATCH_MAX_TARGET_POINTS = 65 535 points, but near that size
// the one-workgroup registration (target in HBM scratch: 256 pairs x 60 000 points no longer fit the
// Infinity Cache) is no faster than a registration of its own on the whole device.  Measured, 256 pairs per
// launch against the stream-per-worker pool: 20 000 points 43 000 pairs/s against 7 700, 30 000 points
// 24 500 against 5 200, 45 000 points 7 800 against 3 800, 60 000 points ~3 000 against ~3 500.
size_t ICPMatcher::maxBatchTargetPoints() { return 50000; }

bool ICPMatcher::batchable(const PCLPointCloudPtr &r, const PCLPointCloudPtr &t) const {
    if (devices.size() > 1 || !r || !t) return false;
    // voxel-filtered matchers: what counts is the size AFTER the filter, which the batch finds out on the
    // device (pairs that stay too large are registered one by one inside wm_icp_batch_match); raw scans
    // of up to 200 000 points are taken
    if (params.res > 0) return cloudSize(r) <= 200000 && cloudSize(t) <= 200000;
    // (the source of a resident registration may be any size in principle, but a launch stages the sources
    // of up to 256 pairs at once: a queue of dense scans against sparse key frames would cost gigabytes
    // of staging per worker and run on one compute unit each -- those go one by one, on the whole device)
    return cloudSize(t) <= maxBatchTargetPoints() && cloudSize(r) <= 200000;
}

bool ICPMatcher::matchBatch(const std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> &pairs,
                            BatchOutcomes &out) {
    out.clear();
    if (pairs.empty()) return true;
    for (const auto &pr : pairs)
        if (!batchable(pr.first, pr.second)) return false;
    if (!ensureContext()) return false;
    wm_icp_params p;
    wm_icp_default_params(&p);
    p.max_corr = params.max_corr;  // as match() sets them (icp.cpp:47-50)
    p.max_iter = params.max_iter;
    p.t_eps = params.t_eps;
    p.fit_eps = params.fit_eps;
    const size_t n = pairs.size();
    std::vector<wm_batch_item> items(n);
    for (size_t k = 0; k < n; ++k) {
        items[k].src = cloudData(pairs[k].first);
        items[k].n_src = cloudSize(pairs[k].first);
        items[k].target = cloudData(pairs[k].second);
        items[k].n_target = cloudSize(pairs[k].second);
    }
    std::vector<double> T(16 * n), info(36 * n);
    std::vector<int> status(n, WM_ERR_STATE);
    // estimateInfo(): whatever estimator is configured, the switch without `break` (icp.cpp:136-141)
    // ends in estimateLUMold, whose result is the one that stays in `information`
    const int rc = wm_icp_batch_match(ctx, items.data(), (int) n, kCloudStride, WM_MEM_HOST, &p, params.res,
                                      params.multiscale_steps, 1, T.data(), info.data(), nullptr, status.data());
    if (!shim::succeeded(rc, "wm_icp_batch_match", ctx)) return false;
    out.resize(n);
    for (size_t k = 0; k < n; ++k) {
        ref = pairs[k].first;
        target = pairs[k].second;
        converged = status[k] == WM_OK;
        if (converged) shim::toAffine(&T[16 * k], result);  // anything else leaves `result` as it was (icp.cpp:132)
        if (status[k] >= 0 && items[k].n_src > 0 && items[k].n_target > 0) storeInfo(&info[36 * k], information);
        out[k].matched = converged;
        out[k].transform = result;
        out[k].info = information;
    }
    converged = false;  // (no correspondences are kept on the device for a later estimateLUM / estimateCensi)
    lastMatch = kNone;  // ... nor for estimateLUMold: the context may still hold an older match()'s
    return true;
}

// ---- information matrix
// The reference's dispatch (icp.cpp:136-141) is a switch without `break`: LUM runs all three
// estimators, CENSI the last two, and whatever was asked for, LUMold writes `information` last.
// Kept, because results depend on it.
void ICPMatcher::estimateInfo() {
    const int first = params.covar_estimator;
    if (first <= ICPMatcherParams::LUM) estimateLUM();
    if (first <= ICPMatcherParams::CENSI) estimateCensi();
    if (first <= ICPMatcherParams::LUMold) estimateLUMold();
}


void ICPMatcher::estimateLUM() {
    if (lastMatch == kNone || !converged) return;  // hasConverged() guard of icp_pcl_functions.cpp:190
    double info[36];
    int degenerate = 0;
    const int rc = lastMatch == kOnMulti ? wm_multi_icp_info(multi, WM_INFO_LUM, nullptr, 0, 0, 0, info, &degenerate)
                                         : wm_icp_info(ctx, WM_INFO_LUM, nullptr, 0, 0, 0, info, &degenerate);
    if (rc != WM_OK) return;
    warnIfDegenerate(degenerate);
    storeInfo(info, information);
}

void ICPMatcher::estimateLUMold() {
    // (the reference's estimateLUMold has no hasConverged() guard, but it works on the clouds of the last
    // align of this matcher: after a failed match, a matchBatch() or before any match there are none here)
    if (lastMatch == kNone) return;
    double info[36];
    int degenerate = 0;
    const int rc = lastMatch == kOnMulti
                       ? wm_multi_icp_info(multi, WM_INFO_LUMOLD, nullptr, 0, 0, params.max_corr, info, &degenerate)
                       : wm_icp_info(ctx, WM_INFO_LUMOLD, nullptr, 0, 0, params.max_corr, info, &degenerate);
    if (rc != WM_OK) return;
    warnIfDegenerate(degenerate);
    storeInfo(info, information);
}

void ICPMatcher::estimateCensi() {
    if (lastMatch == kNone || !converged) return;  // icp.cpp:174
    double info[36], T[16];
    shim::fromAffine(result, T);
    const int rc = lastMatch == kOnMulti
                       ? wm_multi_icp_info(multi, WM_INFO_CENSI, T, params.lidar_lin_covar, params.lidar_ang_covar, 0,
                                           info, nullptr)
                       : wm_icp_info(ctx, WM_INFO_CENSI, T, params.lidar_lin_covar, params.lidar_ang_covar, 0, info,
                                     nullptr);
    if (rc != WM_OK) return;
    storeInfo(info, information);
}

}  // namespace wave
