// wave::NDTMatcher: parameters in, one wm_ndt_align out (reference behaviour:
// wave_matching/src/ndt.cpp:6-65).
#include "wave/matching/ndt.hpp"

#include <cstdlib>

#include "shim.hpp"

namespace wave {

NDTMatcherParams::NDTMatcherParams(const std::string &config_path) {
    shim::loadYaml(config_path, {{"step_size", &step_size}, {"max_iter", &max_iter}, {"t_eps", &t_eps}, {"res", &res}});
}

namespace {
int &pcl18Setting() {
    static int setting = -1;  // -1: follow the environment
    return setting;
}
bool pcl18StepRule() {
    if (pcl18Setting() >= 0) return pcl18Setting() != 0;
    const char *e = std::getenv("WAVE_NDT_PCL18_STEP_RULE");
    return e && std::atoi(e) != 0;
}
}  // namespace

void NDTMatcher::setPcl18StepRule(bool on) { pcl18Setting() = on ? 1 : 0; }

void NDTMatcher::setThreadDevice(int device) { shim::setThreadDevice(device); }

NDTMatcher::NDTMatcher(NDTMatcherParams params1)
    : ctx(nullptr), device(shim::defaultDevice()), ref(shim::emptyCloud()), target(shim::emptyCloud()),
      params(params1), pcl18_step_rule(pcl18StepRule()), target_on_device(false) {
    if (params.res < params.min_res) {  // ndt.cpp:23-26: refuse, say so, carry on with the floor
        LOG_ERROR("Invalid resolution given, using minimum");
        params.res = params.min_res;
    }
    resolution = params.res;
}

NDTMatcher::NDTMatcher(const NDTMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), ctx(nullptr), device(o.device), ref(o.ref), target(o.target),
      params(o.params), pcl18_step_rule(o.pcl18_step_rule), target_on_device(false) {}

NDTMatcher &NDTMatcher::operator=(const NDTMatcher &o) {
    if (this == &o) return *this;
    shim::release(ctx);
    Matcher<PCLPointCloudPtr>::operator=(o);
    device = o.device;
    ref = o.ref;
    target = o.target;
    params.step_size = o.params.step_size;  // (member-wise: min_res is const)
    params.max_iter = o.params.max_iter;
    params.t_eps = o.params.t_eps;
    params.res = o.params.res;
    pcl18_step_rule = o.pcl18_step_rule;
    target_on_device = false;
    return *this;
}

NDTMatcher::~NDTMatcher() { shim::release(ctx); }

bool NDTMatcher::ensureContext() { return shim::acquire(ctx, device); }

void NDTMatcher::setRef(const PCLPointCloudPtr &cloud) { ref = cloud; }  // read by match() (PCL aliases it)

void NDTMatcher::setTarget(const PCLPointCloudPtr &cloud) {
    target = cloud;
    target_on_device = false;
    if (!ensureContext()) return;
    // ndt.cpp:55: setInputTarget builds the voxel grid now
    target_on_device =
        shim::succeeded(wm_set_target(ctx, cloudData(target), cloudSize(target), kCloudStride, WM_MEM_HOST),
                        "wm_set_target", ctx) &&
        (cloudSize(target) == 0 || shim::succeeded(wm_ndt_build_model(ctx, params.res), "wm_ndt_build_model", ctx));
}

bool NDTMatcher::match() {
    if (!ensureContext()) return false;
    if (!shim::succeeded(wm_set_source(ctx, cloudData(ref), cloudSize(ref), kCloudStride, WM_MEM_HOST),
                         "wm_set_source", ctx))
        return false;
    if (!target_on_device) {  // a copy of a matcher, or a context that could not be opened at set time
        if (!shim::succeeded(wm_set_target(ctx, cloudData(target), cloudSize(target), kCloudStride, WM_MEM_HOST),
                             "wm_set_target", ctx))
            return false;
        target_on_device = true;
    }

    wm_ndt_params p;
    wm_ndt_default_params(&p);
    p.res = params.res;              // setResolution,            ndt.cpp:32
    p.step_size = params.step_size;  // setStepSize,              ndt.cpp:31
    p.t_eps = params.t_eps;          // setTransformationEpsilon, ndt.cpp:30
    p.max_iter = params.max_iter;    // setMaximumIterations,     ndt.cpp:33
    p.skip_line_search = pcl18_step_rule ? 1 : 0;
    double T[16];
    wm_ndt_stats stats;
    if (!shim::succeeded(wm_ndt_align(ctx, &p, T, &stats), "wm_ndt_align", ctx)) return false;
    shim::toAffine(T, result);
    return true;
}

bool NDTMatcher::batchable(const PCLPointCloudPtr &r, const PCLPointCloudPtr &t) const {
    return r && t && cloudSize(r) <= (size_t) WM_NDT_BATCH_MAX_POINTS && cloudSize(t) <= (size_t) WM_NDT_BATCH_MAX_POINTS;
}

bool NDTMatcher::matchBatch(const std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> &pairs, BatchOutcomes &out) {
    out.clear();
    if (pairs.empty()) return true;
    for (const auto &pr : pairs)
        if (!batchable(pr.first, pr.second)) return false;
    if (!ensureContext()) return false;
    wm_ndt_params p;
    wm_ndt_default_params(&p);
    p.res = params.res;  // as match() sets them (ndt.cpp:30-33)
    p.step_size = params.step_size;
    p.t_eps = params.t_eps;
    p.max_iter = params.max_iter;
    p.skip_line_search = pcl18_step_rule ? 1 : 0;
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
    const int rc = wm_ndt_batch_match(ctx, items.data(), (int) n, kCloudStride, WM_MEM_HOST, &p, T.data(), nullptr, status.data(), nullptr);
    if (!shim::succeeded(rc, "wm_ndt_batch_match", ctx)) return false;
    out.resize(n);
    for (size_t k = 0; k < n; ++k) {
        ref = pairs[k].first;
        target = pairs[k].second;
        const bool ok = status[k] == WM_OK;
        if (ok) shim::toAffine(&T[16 * k], result);  // anything else leaves `result` as it was (ndt.cpp:59-63)
        out[k].matched = ok;
        out[k].transform = result;
        out[k].info = information;
    }
    target_on_device = false;  // (the context holds no model of the last pair's target)
    return true;
}

}  // namespace wave
