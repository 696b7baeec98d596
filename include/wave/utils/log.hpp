// LOG_ERROR / LOG_INFO exactly as libwave defines them
// (wave_utils/include/wave/utils/log.hpp:21-28).
#ifndef WAVE_UTILS_LOG_HPP
#define WAVE_UTILS_LOG_HPP
#include <cstdio>
#include <cstring>

#define FILENAME (strrchr(__FILE__, '/') ? strrchr(__FILE__, '/') + 1 : __FILE__)
#define LOG_ERROR(M, ...) \
    fprintf(stderr, "[ERROR] [%s:%d] " M "\n", FILENAME, __LINE__, ##__VA_ARGS__)
#define LOG_INFO(M, ...) fprintf(stdout, "[INFO] " M "\n", ##__VA_ARGS__)
#endif
