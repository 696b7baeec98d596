// ConfigParser: the subset of wave_utils' YAML parameter loader that the matcher
// parameter structs use (wave_utils/include/wave/utils/config.hpp:27-149,
// wave_utils/src/config.cpp:80-99): scalar parameters addressed by key, every key
// required unless `optional`, load() returns a ConfigStatus.  yaml-cpp is not
// available here, so the files are read with a flat `key: value  # comment` parser,
// which is exactly the shape of wave_matching/{config,tests/config}/*.yaml.
#ifndef WAVE_UTILS_CONFIG_HPP
#define WAVE_UTILS_CONFIG_HPP
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace wave {

enum class ConfigStatus {
    OK = 0,
    MissingOptionalKey = 1,
    FileError = -1,
    KeyError = -2,
    ConversionError = -3
};

struct ConfigParamBase {
    ConfigParamBase(std::string key_, bool optional_) : key(std::move(key_)), optional(optional_) {}
    virtual ~ConfigParamBase() = default;
    virtual bool assign(const std::string &text) const = 0;
    std::string key;
    bool optional;
};

template <typename T>
struct ConfigParam : ConfigParamBase {
    ConfigParam(std::string key_, T *out_, bool optional_)
        : ConfigParamBase(std::move(key_), optional_), out(out_) {}
    bool assign(const std::string &text) const override;
    T *out;
};

class ConfigParser {
 public:
    bool config_loaded = false;
    std::vector<std::shared_ptr<ConfigParamBase>> params;

    template <typename T>
    void addParam(std::string key, T *out, bool optional = false) {
        params.push_back(std::make_shared<ConfigParam<T>>(std::move(key), out, optional));
    }
    ConfigStatus checkKey(const std::string &key, bool optional);
    ConfigStatus load(const std::string &config_file);

 private:
    std::map<std::string, std::string> values_;
};

}  // namespace wave
#endif
