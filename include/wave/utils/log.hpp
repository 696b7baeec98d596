// The two logging macros the matching module uses.  Same names and output shape as libwave's
// (wave_utils/include/wave/utils/log.hpp:21-28) so that log scrapers keep working:
//   [ERROR] [file.cpp:123] message        on stderr
//   [INFO] message                        on stdout
#ifndef WAVE_UTILS_LOG_HPP
#define WAVE_UTILS_LOG_HPP

#include <cstdio>

namespace wave {
namespace detail {
// basename of a path literal, evaluated where the macro expands
inline const char *logBasename(const char *path) {
    const char *base = path;
    for (const char *p = path; *p; ++p)
        if (*p == '/') base = p + 1;
    return base;
}
}  // namespace detail
}  // namespace wave

#define FILENAME (::wave::detail::logBasename(__FILE__))
#define LOG_ERROR(M, ...) \
    std::fprintf(stderr, "[ERROR] [%s:%d] " M "\n", FILENAME, __LINE__, ##__VA_ARGS__)
#define LOG_INFO(M, ...) std::fprintf(stdout, "[INFO] " M "\n", ##__VA_ARGS__)

#endif  // WAVE_UTILS_LOG_HPP
