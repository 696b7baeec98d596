// Umbrella for the slice of wave_utils the matching module touches: the flat-YAML
// ConfigParser, the logging macros and the Eigen typedefs (Affine3, Mat6, ...).
#ifndef WAVE_UTILS_UTILS_HPP
#define WAVE_UTILS_UTILS_HPP

#include "wave/utils/math.hpp"    // Affine3, Mat6, Vec3 ... (Eigen, or wave/compat/eigen_min.hpp)
#include "wave/utils/log.hpp"     // LOG_ERROR, LOG_INFO
#include "wave/utils/config.hpp"  // ConfigParser, ConfigStatus

namespace wave {
// Which implementation of the matching module a translation unit was compiled against.
constexpr const char *kMatchingBackend = "wavematch-hip (gfx950)";
}  // namespace wave

#endif  // WAVE_UTILS_UTILS_HPP
