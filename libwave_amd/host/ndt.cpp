// wave::NDTMatcher over the C ABI (mirror of wave_matching/src/ndt.cpp:6-65).
#include "wave/matching/ndt.hpp"

#include <cstdlib>
#include <stdexcept>

#include "wavematch.h"

namespace wave {

NDTMatcherParams::NDTMatcherParams(const std::string &config_path) {
    ConfigParser parser;
    parser.addParam("step_size", &this->step_size);
    parser.addParam("max_iter", &this->max_iter);
    parser.addParam("t_eps", &this->t_eps);
    parser.addParam("res", &this->res);

    if (parser.load(config_path) != ConfigStatus::OK) {
        throw std::runtime_error{"Failed to Load Matcher Config"};
    }
}

NDTMatcher::NDTMatcher(NDTMatcherParams params1)
    : ctx(nullptr), device(0), params(params1), ref_dirty(true), target_dirty(true) {
    const char *e = std::getenv("WAVE_MATCHING_DEVICE");
    this->device = e ? std::atoi(e) : 0;
    this->ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    this->target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();

    if (this->params.res < this->params.min_res) {
        LOG_ERROR("Invalid resolution given, using minimum");
        this->params.res = this->params.min_res;
    }
    this->resolution = this->params.res;
}

NDTMatcher::NDTMatcher(const NDTMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), ctx(nullptr), device(o.device), ref(o.ref), target(o.target),
      params(o.params), ref_dirty(true), target_dirty(true) {}

NDTMatcher::~NDTMatcher() {
    if (this->ctx) wm_ctx_destroy(this->ctx);
}

bool NDTMatcher::ensureContext() {
    if (this->ctx) return true;
    int rc = wm_ctx_create(&this->ctx, this->device);
    if (rc != WM_OK) {
        LOG_ERROR("wm_ctx_create(device %d) failed: %s", this->device, wm_strerror(rc));
        this->ctx = nullptr;
        return false;
    }
    return true;
}

void NDTMatcher::setRef(const PCLPointCloudPtr &ref) {
    this->ref = ref;
    this->ref_dirty = true;  // uploaded at match(): the cloud is aliased, like ndt.cpp:49-50
}

void NDTMatcher::setTarget(const PCLPointCloudPtr &target) {
    this->target = target;
    this->target_dirty = true;
}

bool NDTMatcher::match() {
    if (!this->ensureContext()) return false;
    int rc;
    rc = wm_set_source(this->ctx, this->ref->points.data(), this->ref->points.size(),
                       sizeof(pcl::PointXYZ), WM_MEM_HOST);
    if (rc == WM_OK)
        rc = wm_set_target(this->ctx, this->target->points.data(), this->target->points.size(),
                           sizeof(pcl::PointXYZ), WM_MEM_HOST);
    if (rc != WM_OK) {
        LOG_ERROR("NDT cloud upload failed: %s [%s]", wm_strerror(rc), wm_last_error(this->ctx));
        return false;
    }
    wm_ndt_params p;
    wm_ndt_default_params(&p);
    p.t_eps = this->params.t_eps;          // ndt.cpp:30
    p.step_size = this->params.step_size;  // ndt.cpp:31
    p.res = this->params.res;              // ndt.cpp:32
    p.max_iter = this->params.max_iter;    // ndt.cpp:33
    double T[16];
    wm_ndt_stats st;
    rc = wm_ndt_align(this->ctx, &p, T, &st);
    if (rc < 0) {
        LOG_ERROR("wm_ndt_align failed: %s [%s]", wm_strerror(rc), wm_last_error(this->ctx));
        return false;
    }
    if (rc != WM_OK) return false;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) this->result.matrix()(i, j) = T[i * 4 + j];
    return true;
}

}  // namespace wave
