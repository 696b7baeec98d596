// wave::GICPMatcher: parameters in, one wm_gicp_match out (reference behaviour:
// wave_matching/src/gicp.cpp:6-64).
#include "wave/matching/gicp.hpp"

#include <cstdlib>
#include <string>

#include "shim.hpp"

namespace wave {

GICPMatcherParams::GICPMatcherParams(const std::string &config_path) {
    // Faithful to a reference quirk (gicp.cpp:8-13): the loader there binds the YAML keys to
    // local variables that shadow the members, so a valid file is REQUIRED but its values never
    // reach the struct -- the defaults stay.
    int corr_rand_unused = 0, max_iter_unused = 0;
    double r_eps_unused = 0, fit_eps_unused = 0;
    shim::loadYaml(config_path, {{"corr_rand", &corr_rand_unused},
                                 {"max_iter", &max_iter_unused},
                                 {"r_eps", &r_eps_unused},
                                 {"fit_eps", &fit_eps_unused}});
}

void GICPMatcher::setThreadDevice(int device) { shim::setThreadDevice(device); }

GICPMatcher::GICPMatcher(GICPMatcherParams params1)
    : ctx(nullptr), device(shim::defaultDevice()), ref(shim::emptyCloud()), target(shim::emptyCloud()),
      params(params1), ref_on_device(false), target_on_device(false), objective(-1) {
    resolution = params.res > 0 ? params.res : -1.0f;
}

// A copy shares parameters and cloud handles, never device state (a filtered snapshot held by the
// original's context is re-made from the handle the next time the copy matches).
GICPMatcher::GICPMatcher(const GICPMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), ctx(nullptr), device(o.device), ref(o.ref), target(o.target),
      params(o.params), ref_on_device(false), target_on_device(false), objective(o.objective) {}

GICPMatcher &GICPMatcher::operator=(const GICPMatcher &o) {
    if (this == &o) return *this;
    shim::release(ctx);
    Matcher<PCLPointCloudPtr>::operator=(o);
    device = o.device;
    ref = o.ref;
    target = o.target;
    params = o.params;
    objective = o.objective;
    ref_on_device = target_on_device = false;
    return *this;
}

GICPMatcher::~GICPMatcher() { shim::release(ctx); }

bool GICPMatcher::ensureContext() { return shim::acquire(ctx, device); }

void GICPMatcher::setRef(const PCLPointCloudPtr &cloud) {
    ref = cloud;
    ref_on_device = false;
    if (resolution > 0 && ensureContext())  // gicp.cpp:38-42: filter now, register the filtered copy
        ref_on_device = shim::succeeded(wm_set_source_filtered(ctx, cloudData(ref), cloudSize(ref), kCloudStride,
                                                               WM_MEM_HOST, resolution),
                                        "wm_set_source_filtered", ctx);
}

void GICPMatcher::setTarget(const PCLPointCloudPtr &cloud) {
    target = cloud;
    target_on_device = false;
    if (resolution > 0 && ensureContext())  // gicp.cpp:48-52
        target_on_device = shim::succeeded(wm_set_target_filtered(ctx, cloudData(target), cloudSize(target),
                                                                  kCloudStride, WM_MEM_HOST, resolution),
                                           "wm_set_target_filtered", ctx);
}

namespace {
// how the minimisations' objective is evaluated (include/wavematch.h: wm_gicp_params::objective).  The default is the
// REFERENCE's: PCL's per-pair sums through the float transform (gicp.cpp:58 -> pcl::GeneralizedIterativeClosestPoint).
// The 74-sufficient-statistics form (2.4x faster at 500k points, not PCL's arithmetic: its registrations of noisy
// pairs end up to 1e-3 m from PCL's) is an explicit opt-in: GICPMatcher::setObjective(GICPMatcher::Objective::
// Statistics) on the matcher, or env WAVE_GICP_OBJECTIVE=statistics for matchers that were not told.
int envObjective() {
    static const int v = [] {
        const char *e = std::getenv("WAVE_GICP_OBJECTIVE");
        return e && std::string(e) == "statistics" ? WM_GICP_OBJECTIVE_STATISTICS : WM_GICP_OBJECTIVE_PCL_SUMS;
    }();
    return v;
}
}  // namespace

void GICPMatcher::setObjective(Objective o) {
    objective = o == Objective::Statistics ? WM_GICP_OBJECTIVE_STATISTICS : WM_GICP_OBJECTIVE_PCL_SUMS;
}

GICPMatcher::Objective GICPMatcher::getObjective() const {
    const int o = objective >= 0 ? objective : envObjective();
    return o == WM_GICP_OBJECTIVE_STATISTICS ? Objective::Statistics : Objective::PclSums;
}

bool GICPMatcher::match() {
    if (!ensureContext()) return false;
    wm_gicp_params p;
    wm_gicp_default_params(&p);
    p.objective = objective >= 0 ? objective : envObjective();
    p.corr_rand = params.corr_rand;  // setCorrespondenceRandomness, gicp.cpp:31
    p.max_iter = params.max_iter;    // setMaximumIterations,        gicp.cpp:32
    p.r_eps = params.r_eps;          // setRotationEpsilon,          gicp.cpp:33
    // fit_eps goes to setEuclideanFitnessEpsilon (gicp.cpp:34), which PCL-GICP's loop never reads
    double T[16];
    wm_gicp_stats stats;
    // clouds that are not in the context yet: res <= 0 (handles are read now, as PCL reads its aliased
    // inputs in align), or a copy of a matcher / a context that could not be opened at set time
    if (!ref_on_device) {
        const int rs = resolution > 0 ? wm_set_source_filtered(ctx, cloudData(ref), cloudSize(ref), kCloudStride,
                                                               WM_MEM_HOST, resolution)
                                      : wm_set_source(ctx, cloudData(ref), cloudSize(ref), kCloudStride, WM_MEM_HOST);
        if (!shim::succeeded(rs, "wm_set_source", ctx)) return false;
        ref_on_device = resolution > 0;
    }
    if (!target_on_device) {
        const int rt = resolution > 0 ? wm_set_target_filtered(ctx, cloudData(target), cloudSize(target), kCloudStride,
                                                               WM_MEM_HOST, resolution)
                                      : wm_set_target(ctx, cloudData(target), cloudSize(target), kCloudStride,
                                                      WM_MEM_HOST);
        if (!shim::succeeded(rt, "wm_set_target", ctx)) return false;
        target_on_device = resolution > 0;
    }
    const int rc = wm_gicp_align(ctx, &p, T, &stats);
    if (!shim::succeeded(rc, "wm_gicp_align", ctx)) return false;
    shim::toAffine(T, result);
    return true;
}

bool GICPMatcher::batchable(const PCLPointCloudPtr &r, const PCLPointCloudPtr &t) const {
    if (!r || !t) return false;
    const size_t cap = resolution > 0 ? (size_t) 4 * WM_GICP_BATCH_MAX_POINTS : (size_t) WM_GICP_BATCH_MAX_POINTS;
    return cloudSize(r) <= cap && cloudSize(t) <= cap;
}

bool GICPMatcher::matchBatch(const std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> &pairs, BatchOutcomes &out) {
    out.clear();
    if (pairs.empty()) return true;
    for (const auto &pr : pairs)
        if (!batchable(pr.first, pr.second)) return false;
    if (!ensureContext()) return false;
    wm_gicp_params p;
    wm_gicp_default_params(&p);
    p.objective = objective >= 0 ? objective : envObjective();
    p.corr_rand = params.corr_rand;  // as match() sets them (gicp.cpp:31-33)
    p.max_iter = params.max_iter;
    p.r_eps = params.r_eps;
    const size_t n = pairs.size();
    std::vector<wm_batch_item> items(n);
    for (size_t k = 0; k < n; ++k) {
        items[k].src = cloudData(pairs[k].first);
        items[k].n_src = cloudSize(pairs[k].first);
        items[k].target = cloudData(pairs[k].second);
        items[k].n_target = cloudSize(pairs[k].second);
    }
    std::vector<double> T(16 * n);
    std::vector<int> status(n, WM_ERR_STATE);
    const int rc = wm_gicp_batch_match(ctx, items.data(), (int) n, kCloudStride, WM_MEM_HOST, &p, resolution, T.data(), nullptr,
                                       status.data(), nullptr);
    if (!shim::succeeded(rc, "wm_gicp_batch_match", ctx)) return false;
    out.resize(n);
    for (size_t k = 0; k < n; ++k) {
        ref = pairs[k].first;
        target = pairs[k].second;
        const bool ok = status[k] == WM_OK;
        if (ok) shim::toAffine(&T[16 * k], result);  // anything else leaves `result` as it was (gicp.cpp:59-63)
        out[k].matched = ok;
        out[k].transform = result;
        out[k].info = information;
    }
    ref_on_device = target_on_device = false;  // (the context holds no snapshot of the last pair's clouds)
    return true;
}

}  // namespace wave
