// wave::ICPMatcher over the C ABI of libwavematch_hip.so.  Function-by-function mirror
// of the reference's wave_matching/src/icp.cpp (constructor / YAML params :6-51,
// setRef/setTarget :67-73, match :75-133, estimateInfo :135-142) with the PCL calls
// replaced by wm_* calls; no registration arithmetic happens on the host.
#include "wave/matching/icp.hpp"

#include <cstdlib>
#include <stdexcept>

#include "wavematch.h"

namespace wave {

namespace {
int g_default_device = -1;
int default_device() {
    if (g_default_device >= 0) return g_default_device;
    const char *e = std::getenv("WAVE_MATCHING_DEVICE");
    return e ? std::atoi(e) : 0;
}
void to_affine(const double T[16], Affine3 *out) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out->matrix()(i, j) = T[i * 4 + j];
}
void from_affine(const Affine3 &a, double T[16]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) T[i * 4 + j] = a.matrix()(i, j);
}
}  // namespace

void ICPMatcher::setDefaultDevice(int device) {
    g_default_device = device;
}

ICPMatcherParams::ICPMatcherParams(const std::string &config_path) {
    ConfigParser parser;
    int covar_est_temp;
    parser.addParam("max_corr", &(this->max_corr));
    parser.addParam("max_iter", &(this->max_iter));
    parser.addParam("t_eps", &(this->t_eps));
    parser.addParam("lidar_ang_covar", &(this->lidar_ang_covar));
    parser.addParam("lidar_lin_covar", &(this->lidar_lin_covar));
    parser.addParam("covar_estimator", &covar_est_temp);
    parser.addParam("res", &(this->res));
    parser.addParam("multiscale_steps", &(this->multiscale_steps));
    // NB: like the reference (icp.cpp:9-16) `fit_eps` is NOT read from the file.

    if (parser.load(config_path) != ConfigStatus::OK) {
        throw std::runtime_error{"Failed to Load Matcher Config"};
    }

    if ((covar_est_temp >= ICPMatcherParams::covar_method::LUM) &&
        (covar_est_temp <= ICPMatcherParams::covar_method::LUMold)) {
        this->covar_estimator = static_cast<ICPMatcherParams::covar_method>(covar_est_temp);
    } else {
        LOG_ERROR("Invalid covariance estimate method, using LUM");
        this->covar_estimator = ICPMatcherParams::covar_method::LUM;
    }
}

ICPMatcher::ICPMatcher(ICPMatcherParams params1)
    : params(params1), ctx(nullptr), device(default_device()), converged(false) {
    this->ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    this->target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    this->resolution = this->params.res;
}

// A copy is a fresh matcher with the same parameters and cloud handles: device
// state (like the PCL members of the reference) is per object.
ICPMatcher::ICPMatcher(const ICPMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), params(o.params), ctx(nullptr), device(o.device),
      converged(false), ref(o.ref), target(o.target) {}

ICPMatcher &ICPMatcher::operator=(const ICPMatcher &o) {
    if (this != &o) {
        if (this->ctx) wm_ctx_destroy(this->ctx);
        this->ctx = nullptr;
        Matcher<PCLPointCloudPtr>::operator=(o);
        this->params = o.params;
        this->device = o.device;
        this->converged = false;
        this->ref = o.ref;
        this->target = o.target;
    }
    return *this;
}

ICPMatcher::~ICPMatcher() {
    if (this->ctx) wm_ctx_destroy(this->ctx);
}

bool ICPMatcher::ensureContext() {
    if (this->ctx) return true;
    int rc = wm_ctx_create(&this->ctx, this->device);
    if (rc != WM_OK) {
        LOG_ERROR("wm_ctx_create(device %d) failed: %s", this->device, wm_strerror(rc));
        this->ctx = nullptr;
        return false;
    }
    return true;
}

void ICPMatcher::setRef(const PCLPointCloudPtr &ref) {
    this->ref = ref;
}

void ICPMatcher::setTarget(const PCLPointCloudPtr &target) {
    this->target = target;
}

bool ICPMatcher::match() {
    this->converged = false;
    if (!this->ensureContext()) return false;
    wm_icp_params p;
    wm_icp_default_params(&p);
    p.max_corr = this->params.max_corr;   // icp.cpp:47
    p.max_iter = this->params.max_iter;   // icp.cpp:48
    p.t_eps = this->params.t_eps;         // icp.cpp:49
    p.fit_eps = this->params.fit_eps;     // icp.cpp:50
    p.carry_state = 1;                    // one PCL object per matcher: criteria state persists
    double T[16];
    wm_icp_stats st;
    static_assert(sizeof(pcl::PointXYZ) == 16, "PointXYZ stride");
    const int rc = wm_icp_match(this->ctx, this->ref->points.data(), this->ref->points.size(),
                                this->target->points.data(), this->target->points.size(),
                                sizeof(pcl::PointXYZ), WM_MEM_HOST, &p, this->params.res,
                                this->params.multiscale_steps, T, &st);
    if (rc < 0) {
        LOG_ERROR("wm_icp_match failed: %s [%s]", wm_strerror(rc), wm_last_error(this->ctx));
        return false;
    }
    if (rc != WM_OK) return false;  // not converged: `result` left untouched (icp.cpp:132)
    to_affine(T, &this->result);
    this->converged = true;
    return true;
}

void ICPMatcher::estimateInfo() {
    // The reference's switch has no `break`s (icp.cpp:136-141): LUM runs LUM, Censi and
    // LUMold; CENSI runs Censi and LUMold; the final `information` is always LUMold's.
    switch (this->params.covar_estimator) {
        case ICPMatcherParams::covar_method::LUM: this->estimateLUM();       // fall through
        case ICPMatcherParams::covar_method::CENSI: this->estimateCensi();   // fall through
        case ICPMatcherParams::covar_method::LUMold: this->estimateLUMold();
        default: return;
    }
}

static void store_info(const double info[36], Mat6 *out) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) (*out)(i, j) = info[i * 6 + j];
}

void ICPMatcher::estimateLUM() {
    if (!this->ctx || !this->converged) return;  // hasConverged() guard, icp_pcl_functions.cpp:190
    double info[36];
    int degenerate = 0;
    if (wm_icp_info(this->ctx, WM_INFO_LUM, nullptr, 0, 0, 0, info, &degenerate) != WM_OK) return;
    if (degenerate) LOG_ERROR("Covariance matrix calculation was unsuccessful");
    store_info(info, &this->information);
}

void ICPMatcher::estimateLUMold() {
    if (!this->ctx) return;
    double info[36];
    int degenerate = 0;
    if (wm_icp_info(this->ctx, WM_INFO_LUMOLD, nullptr, 0, 0, this->params.max_corr, info,
                    &degenerate) != WM_OK)
        return;
    if (degenerate) LOG_ERROR("Covariance matrix calculation was unsuccessful");
    store_info(info, &this->information);
}

void ICPMatcher::estimateCensi() {
    if (!this->ctx || !this->converged) return;  // icp.cpp:174
    double info[36], T[16];
    from_affine(this->result, T);
    if (wm_icp_info(this->ctx, WM_INFO_CENSI, T, this->params.lidar_lin_covar,
                    this->params.lidar_ang_covar, 0, info, nullptr) != WM_OK)
        return;
    store_info(info, &this->information);
}

}  // namespace wave
