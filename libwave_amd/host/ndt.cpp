// wave::NDTMatcher: parameters in, one wm_ndt_align out (reference behaviour:
// wave_matching/src/ndt.cpp:6-65).
#include "wave/matching/ndt.hpp"

#include "shim.hpp"

namespace wave {

NDTMatcherParams::NDTMatcherParams(const std::string &config_path) {
    shim::loadYaml(config_path, {{"step_size", &step_size}, {"max_iter", &max_iter}, {"t_eps", &t_eps}, {"res", &res}});
}

NDTMatcher::NDTMatcher(NDTMatcherParams params1)
    : ctx(nullptr), device(shim::defaultDevice()), ref(shim::emptyCloud()), target(shim::emptyCloud()),
      params(params1), ref_dirty(true), target_dirty(true) {
    if (params.res < params.min_res) {  // ndt.cpp:23-26: refuse, say so, carry on with the floor
        LOG_ERROR("Invalid resolution given, using minimum");
        params.res = params.min_res;
    }
    resolution = params.res;
}

NDTMatcher::NDTMatcher(const NDTMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), ctx(nullptr), device(o.device), ref(o.ref), target(o.target),
      params(o.params), ref_dirty(true), target_dirty(true) {}

NDTMatcher::~NDTMatcher() { shim::release(ctx); }

bool NDTMatcher::ensureContext() { return shim::acquire(ctx, device); }

// The handles are only remembered here; the clouds cross to the device inside match(), so a
// caller may still fill them after setRef / setTarget (as with the reference's aliasing).
void NDTMatcher::setRef(const PCLPointCloudPtr &cloud) {
    ref = cloud;
    ref_dirty = true;
}

void NDTMatcher::setTarget(const PCLPointCloudPtr &cloud) {
    target = cloud;
    target_dirty = true;
}

bool NDTMatcher::match() {
    if (!ensureContext()) return false;
    if (!shim::succeeded(wm_set_source(ctx, cloudData(ref), cloudSize(ref), kCloudStride, WM_MEM_HOST),
                         "wm_set_source", ctx) ||
        !shim::succeeded(wm_set_target(ctx, cloudData(target), cloudSize(target), kCloudStride, WM_MEM_HOST),
                         "wm_set_target", ctx))
        return false;
    ref_dirty = target_dirty = false;

    wm_ndt_params p;
    wm_ndt_default_params(&p);
    p.res = params.res;              // setResolution,            ndt.cpp:32
    p.step_size = params.step_size;  // setStepSize,              ndt.cpp:31
    p.t_eps = params.t_eps;          // setTransformationEpsilon, ndt.cpp:30
    p.max_iter = params.max_iter;    // setMaximumIterations,     ndt.cpp:33
    double T[16];
    wm_ndt_stats stats;
    if (!shim::succeeded(wm_ndt_align(ctx, &p, T, &stats), "wm_ndt_align", ctx)) return false;
    shim::toAffine(T, result);
    return true;
}

}  // namespace wave
