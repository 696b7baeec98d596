// wave::GICPMatcher over the C ABI (mirror of wave_matching/src/gicp.cpp:6-64).
#include "wave/matching/gicp.hpp"

#include <cstdlib>
#include <stdexcept>

#include "wavematch.h"

namespace wave {

GICPMatcherParams::GICPMatcherParams(const std::string &config_path) {
    // The reference parses into shadowing locals (gicp.cpp:8-13): the file must exist and
    // hold the keys, but its values are discarded and the struct keeps its defaults.
    ConfigParser parser;
    double r_eps = 1e-8, fit_eps = 1e-2;
    int corr_rand = 10, max_iter = 100;
    parser.addParam("corr_rand", &corr_rand);
    parser.addParam("max_iter", &max_iter);
    parser.addParam("r_eps", &r_eps);
    parser.addParam("fit_eps", &fit_eps);

    if (parser.load(config_path) != ConfigStatus::OK) {
        throw std::runtime_error{"Failed to Load Matcher Config"};
    }
}

GICPMatcher::GICPMatcher(GICPMatcherParams params1) : ctx(nullptr), device(0), params(params1) {
    const char *e = std::getenv("WAVE_MATCHING_DEVICE");
    this->device = e ? std::atoi(e) : 0;
    this->ref = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    this->target = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    if (params.res > 0) {
        this->resolution = params.res;
    } else {
        this->resolution = -1;
    }
}

GICPMatcher::GICPMatcher(const GICPMatcher &o)
    : Matcher<PCLPointCloudPtr>(o), ctx(nullptr), device(o.device), ref(o.ref), target(o.target),
      params(o.params) {}

GICPMatcher::~GICPMatcher() {
    if (this->ctx) wm_ctx_destroy(this->ctx);
}

bool GICPMatcher::ensureContext() {
    if (this->ctx) return true;
    int rc = wm_ctx_create(&this->ctx, this->device);
    if (rc != WM_OK) {
        LOG_ERROR("wm_ctx_create(device %d) failed: %s", this->device, wm_strerror(rc));
        this->ctx = nullptr;
        return false;
    }
    return true;
}

void GICPMatcher::setRef(const PCLPointCloudPtr &ref) {
    this->ref = ref;
}

void GICPMatcher::setTarget(const PCLPointCloudPtr &target) {
    this->target = target;
}

bool GICPMatcher::match() {
    if (!this->ensureContext()) return false;
    wm_gicp_params p;
    wm_gicp_default_params(&p);
    p.corr_rand = this->params.corr_rand;  // gicp.cpp:31
    p.max_iter = this->params.max_iter;    // gicp.cpp:32
    p.r_eps = this->params.r_eps;          // gicp.cpp:33
    // fit_eps -> setEuclideanFitnessEpsilon (gicp.cpp:34): unused by PCL-GICP's loop
    double T[16];
    wm_gicp_stats st;
    const int rc = wm_gicp_match(this->ctx, this->ref->points.data(), this->ref->points.size(),
                                 this->target->points.data(), this->target->points.size(),
                                 sizeof(pcl::PointXYZ), WM_MEM_HOST, &p, this->resolution, T, &st);
    if (rc < 0) {
        LOG_ERROR("wm_gicp_match failed: %s [%s]", wm_strerror(rc), wm_last_error(this->ctx));
        return false;
    }
    if (rc != WM_OK) return false;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) this->result.matrix()(i, j) = T[i * 4 + j];
    return true;
}

}  // namespace wave
