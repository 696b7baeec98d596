#include "shim.hpp"

#include <cstdlib>

namespace wave {
namespace shim {

namespace {
int g_device_override = -1;
thread_local int t_device_override = -1;
}

int defaultDevice() {
    if (t_device_override >= 0) return t_device_override;
    if (g_device_override >= 0) return g_device_override;
    const char *env = std::getenv("WAVE_MATCHING_DEVICE");
    return env ? std::atoi(env) : 0;
}

void setDefaultDevice(int device) { g_device_override = device; }
void setThreadDevice(int device) { t_device_override = device; }

bool acquire(wm_ctx *&ctx, int device) {
    if (ctx) return true;
    const int rc = wm_ctx_create(&ctx, device);
    if (rc == WM_OK) return true;
    LOG_ERROR("wm_ctx_create(device %d) failed: %s", device, wm_strerror(rc));
    ctx = nullptr;
    return false;
}

void loadYaml(const std::string &path, std::initializer_list<YamlField> fields) {
    ConfigParser parser;
    for (const YamlField &f : fields) {
        switch (f.type) {
            case YamlField::INT: parser.addParam(f.key, static_cast<int *>(f.dst)); break;
            case YamlField::FLOAT: parser.addParam(f.key, static_cast<float *>(f.dst)); break;
            case YamlField::DOUBLE: parser.addParam(f.key, static_cast<double *>(f.dst)); break;
        }
    }
    if (parser.load(path) != ConfigStatus::OK) throw std::runtime_error{"Failed to Load Matcher Config"};
}

bool succeeded(int rc, const char *call, wm_ctx *ctx) {
    if (rc < 0) LOG_ERROR("%s failed: %s [%s]", call, wm_strerror(rc), ctx ? wm_last_error(ctx) : "");
    return rc == WM_OK;
}

}  // namespace shim
}  // namespace wave
