// The cloud handle type all matchers exchange, and where its definition comes from.
//
// With PCL installed, <pcl/point_cloud.h> supplies pcl::PointCloud / pcl::PointXYZ and the
// boost::shared_ptr-based ::Ptr; without it (this repository's own builds) the layout-compatible
// stand-ins of wave/compat/pcl_min.hpp do: 16-byte PointXYZ, std::vector storage, loadPCDFile for
// the binary v0.7 fixture, transformPointCloud with PCL's double-precision arithmetic.  Either
// way a matcher reads `cloud->points.data()` with a 16-byte stride and hands that to the C ABI.
#ifndef WAVE_PCL_COMMON_HPP
#define WAVE_PCL_COMMON_HPP

#include "wave/compat/pcl_min.hpp"

namespace wave {

typedef pcl::PointCloud<pcl::PointXYZ>::Ptr PCLPointCloudPtr;  // reference: pcl_common.hpp:22

// stride / size helpers for the C-ABI calls (wm_set_source & co. take pointer + count + stride)
inline const void *cloudData(const PCLPointCloudPtr &c) { return c && !c->points.empty() ? c->points.data() : nullptr; }
inline size_t cloudSize(const PCLPointCloudPtr &c) { return c ? c->points.size() : 0; }
constexpr size_t kCloudStride = sizeof(pcl::PointXYZ);
static_assert(sizeof(pcl::PointXYZ) == 16, "pcl::PointXYZ is x, y, z + padding in 16 bytes");

}  // namespace wave

#endif  // WAVE_PCL_COMMON_HPP
