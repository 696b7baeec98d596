// pcl_min.hpp -- the PCL / Boost types that appear in libwave's matcher API
// (`boost::shared_ptr<pcl::PointCloud<pcl::PointXYZ>>`, wave_matching/include/wave/
// matching/pcl_common.hpp:22), for builds where PCL / Boost are not installed.  If
// <pcl/point_cloud.h> is available the real headers are used and this file defines
// nothing.  Layout-compatible with PCL: sizeof(pcl::PointXYZ) == 16, points stored
// contiguously in `points`.
#pragma once

#if defined(WAVE_MATCHING_USE_SYSTEM_PCL) || __has_include(<pcl/point_cloud.h>)
#include <pcl/common/transforms.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#else

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "wave/compat/eigen_min.hpp"

namespace boost {
template <class T>
using shared_ptr = std::shared_ptr<T>;
template <class T, class... A>
std::shared_ptr<T> make_shared(A &&... a) {
    return std::make_shared<T>(std::forward<A>(a)...);
}
}  // namespace boost

namespace pcl {

struct alignas(16) PointXYZ {
    float x, y, z, pad;
    PointXYZ() : x(0), y(0), z(0), pad(1.0f) {}
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), pad(1.0f) {}
};
static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ must be 16 bytes");

template <typename PointT>
class PointCloud {
 public:
    typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
    typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;

    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void resize(size_t n) {
        points.resize(n);
        width = static_cast<uint32_t>(n);
        height = 1;
    }
    void clear() {
        points.clear();
        width = height = 0;
    }
    void push_back(const PointT &p) {
        points.push_back(p);
        width = static_cast<uint32_t>(points.size());
        height = 1;
    }
    PointT &at(size_t i) { return points.at(i); }
    const PointT &at(size_t i) const { return points.at(i); }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }
    typename std::vector<PointT>::iterator begin() { return points.begin(); }
    typename std::vector<PointT>::iterator end() { return points.end(); }
};

// pcl::transformPointCloud(in, out, Eigen::Affine3d): double arithmetic, float store
// (host version for test fixtures; the matchers use the device kernel)
inline void transformPointCloud(const PointCloud<PointXYZ> &in, PointCloud<PointXYZ> &out,
                                const Eigen::Affine3d &T) {
    const Eigen::Matrix4d &m = T.matrix();
    if (&in != &out) {
        out.points.resize(in.points.size());
        out.width = in.width;
        out.height = in.height;
        out.is_dense = in.is_dense;
    }
    for (size_t i = 0; i < in.points.size(); ++i) {
        const double x = in.points[i].x, y = in.points[i].y, z = in.points[i].z;
        PointXYZ p;
        p.x = static_cast<float>(m(0, 0) * x + m(0, 1) * y + m(0, 2) * z + m(0, 3));
        p.y = static_cast<float>(m(1, 0) * x + m(1, 1) * y + m(1, 2) * z + m(1, 3));
        p.z = static_cast<float>(m(2, 0) * x + m(2, 1) * y + m(2, 2) * z + m(2, 3));
        out.points[i] = p;
    }
}

namespace io {
// Minimal PCD v0.7 reader (ascii / binary, float x y z fields anywhere in the record):
// enough for the reference fixture wave_matching/tests/data/testscan.pcd.
// returns 0 on success, -1 on error (as pcl::io::loadPCDFile).
inline int loadPCDFile(const std::string &path, PointCloud<PointXYZ> &cloud) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    std::vector<std::string> fields;
    std::vector<int> sizes, counts;
    std::vector<char> types;
    size_t npts = 0, w = 0, h = 1;
    std::string mode, line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "FIELDS") {
            std::string s;
            while (ss >> s) fields.push_back(s);
        } else if (key == "SIZE") {
            int v;
            while (ss >> v) sizes.push_back(v);
        } else if (key == "TYPE") {
            char c;
            while (ss >> c) types.push_back(c);
        } else if (key == "COUNT") {
            int v;
            while (ss >> v) counts.push_back(v);
        } else if (key == "WIDTH") {
            ss >> w;
        } else if (key == "HEIGHT") {
            ss >> h;
        } else if (key == "POINTS") {
            ss >> npts;
        } else if (key == "DATA") {
            ss >> mode;
            break;
        }
    }
    if (npts == 0) npts = w * h;
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (fields.size() != sizes.size() || fields.size() != counts.size()) return -1;
    size_t off[3] = {0, 0, 0}, stride = 0, col[3] = {0, 0, 0}, ncol = 0;
    bool have[3] = {false, false, false};
    for (size_t i = 0; i < fields.size(); ++i) {
        for (int a = 0; a < 3; ++a)
            if (fields[i] == std::string(1, "xyz"[a])) {
                if (sizes[i] != 4 || (i < types.size() && types[i] != 'F')) return -1;
                off[a] = stride;
                col[a] = ncol;
                have[a] = true;
            }
        stride += static_cast<size_t>(sizes[i]) * counts[i];
        ncol += counts[i];
    }
    if (!(have[0] && have[1] && have[2])) return -1;
    cloud.points.resize(npts);
    cloud.width = static_cast<uint32_t>(npts);
    cloud.height = 1;
    cloud.is_dense = true;
    if (mode == "binary") {
        std::vector<char> rec(stride * npts);
        f.read(rec.data(), static_cast<std::streamsize>(rec.size()));
        if (static_cast<size_t>(f.gcount()) != rec.size()) return -1;
        for (size_t i = 0; i < npts; ++i) {
            float v[3];
            for (int a = 0; a < 3; ++a) std::memcpy(&v[a], &rec[i * stride + off[a]], 4);
            cloud.points[i] = PointXYZ(v[0], v[1], v[2]);
        }
    } else if (mode == "ascii") {
        for (size_t i = 0; i < npts; ++i) {
            if (!std::getline(f, line)) return -1;
            std::istringstream ss(line);
            std::vector<double> vals;
            double v;
            while (ss >> v) vals.push_back(v);
            if (vals.size() < ncol) return -1;
            cloud.points[i] = PointXYZ(static_cast<float>(vals[col[0]]), static_cast<float>(vals[col[1]]),
                                       static_cast<float>(vals[col[2]]));
        }
    } else {
        return -1;
    }
    for (const auto &p : cloud.points)
        if (!(p.x == p.x && p.y == p.y && p.z == p.z)) cloud.is_dense = false;
    return 0;
}
}  // namespace io
}  // namespace pcl
#endif
