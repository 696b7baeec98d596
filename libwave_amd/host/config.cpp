// ConfigParser over flat `key: value  # comment` YAML (see include/wave/utils/config.hpp).
// Mirrors the behaviour of wave_utils/src/config.cpp:80-99: missing file -> FileError,
// every non-optional key must be present -> KeyError, bad conversion -> ConversionError.
#include "wave/utils/config.hpp"

#include <cstdlib>
#include <fstream>
#include <sstream>

namespace wave {

static std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

template <>
bool ConfigParam<double>::assign(const std::string &t) const {
    char *end = nullptr;
    double v = std::strtod(t.c_str(), &end);
    if (end == t.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}
template <>
bool ConfigParam<float>::assign(const std::string &t) const {
    char *end = nullptr;
    double v = std::strtod(t.c_str(), &end);
    if (end == t.c_str() || *end != '\0') return false;
    *out = static_cast<float>(v);
    return true;
}
template <>
bool ConfigParam<int>::assign(const std::string &t) const {
    char *end = nullptr;
    long v = std::strtol(t.c_str(), &end, 10);
    if (end == t.c_str() || *end != '\0') return false;
    *out = static_cast<int>(v);
    return true;
}
template <>
bool ConfigParam<bool>::assign(const std::string &t) const {
    if (t == "true" || t == "True" || t == "1") {
        *out = true;
        return true;
    }
    if (t == "false" || t == "False" || t == "0") {
        *out = false;
        return true;
    }
    return false;
}
template <>
bool ConfigParam<std::string>::assign(const std::string &t) const {
    *out = t;
    return true;
}

ConfigStatus ConfigParser::checkKey(const std::string &key, bool optional) {
    if (values_.count(key)) return ConfigStatus::OK;
    return optional ? ConfigStatus::MissingOptionalKey : ConfigStatus::KeyError;
}

ConfigStatus ConfigParser::load(const std::string &config_file) {
    std::ifstream f(config_file);
    if (!f) return ConfigStatus::FileError;
    values_.clear();
    std::string line;
    while (std::getline(f, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
        if (!key.empty()) values_[key] = val;
    }
    config_loaded = true;
    for (const auto &p : params) {
        ConfigStatus st = checkKey(p->key, p->optional);
        if (st == ConfigStatus::MissingOptionalKey) continue;
        if (st != ConfigStatus::OK) return st;
        if (!p->assign(values_[p->key])) return ConfigStatus::ConversionError;
    }
    return ConfigStatus::OK;
}

}  // namespace wave
