// Plumbing shared by the three matcher shims (icp.cpp, gicp.cpp, ndt.cpp): device choice,
// lazy wm_ctx creation, YAML parameter files, row-major 4x4 <-> Affine3.
#ifndef WAVE_MATCHING_HOST_SHIM_HPP
#define WAVE_MATCHING_HOST_SHIM_HPP

#include <initializer_list>
#include <stdexcept>
#include <string>

#include "wave/matching/pcl_common.hpp"
#include "wave/utils/utils.hpp"
#include "wavematch.h"

namespace wave {
namespace shim {

// HIP device new matchers bind to: ICPMatcher::setDefaultDevice, else env
// WAVE_MATCHING_DEVICE, else 0
int defaultDevice();
void setDefaultDevice(int device);
void setThreadDevice(int device);  // matchers constructed by the calling thread from now on (< 0: back to the default)

// Creates the context on first use, in the calling thread.  false (and a LOG_ERROR) if the
// device cannot be opened; the matcher then reports "no match".
bool acquire(wm_ctx *&ctx, int device);
inline void release(wm_ctx *&ctx) {
    if (ctx) wm_ctx_destroy(ctx);
    ctx = nullptr;
}

// One "key -> destination" entry of a matcher's YAML file.
struct YamlField {
    const char *key;
    enum { INT, FLOAT, DOUBLE } type;
    void *dst;
    YamlField(const char *k, int *p) : key(k), type(INT), dst(p) {}
    YamlField(const char *k, float *p) : key(k), type(FLOAT), dst(p) {}
    YamlField(const char *k, double *p) : key(k), type(DOUBLE), dst(p) {}
};
// Reads every listed key from the flat YAML file at `path`; a missing file or key throws
// std::runtime_error("Failed to Load Matcher Config"), as all three reference loaders do
// (icp.cpp:18-20, gicp.cpp:15-17, ndt.cpp:13-15).
void loadYaml(const std::string &path, std::initializer_list<YamlField> fields);

inline void toAffine(const double T[16], Affine3 &out) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out.matrix()(r, c) = T[4 * r + c];
}
inline void fromAffine(const Affine3 &in, double T[16]) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) T[4 * r + c] = in.matrix()(r, c);
}

// rc of a wm_* call -> the bool a matcher returns; negative codes are real errors and logged
bool succeeded(int rc, const char *call, wm_ctx *ctx);

inline PCLPointCloudPtr emptyCloud() { return boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>(); }

}  // namespace shim
}  // namespace wave

#endif  // WAVE_MATCHING_HOST_SHIM_HPP
