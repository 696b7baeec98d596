// Mirrors wave_matching/include/wave/matching/pcl_common.hpp:10-22.
#ifndef WAVE_PCL_COMMON_HPP
#define WAVE_PCL_COMMON_HPP

#include "wave/compat/pcl_min.hpp"

namespace wave {

/** Shorthand for the pointcloud object type used by the scan matching
 * implementations (reference: pcl_common.hpp:22). */
typedef pcl::PointCloud<pcl::PointXYZ>::Ptr PCLPointCloudPtr;

}  // namespace wave

#endif  // WAVE_PCL_COMMON_HPP
