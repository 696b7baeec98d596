#ifndef WAVE_UTILS_UTILS_HPP
#define WAVE_UTILS_UTILS_HPP
#include "wave/utils/config.hpp"
#include "wave/utils/log.hpp"
#include "wave/utils/math.hpp"
#endif
