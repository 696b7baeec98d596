// The Eigen typedefs wave_matching uses (wave_utils/include/wave/utils/math.hpp:27-44).
#ifndef WAVE_UTILS_MATH_HPP
#define WAVE_UTILS_MATH_HPP
#include "wave/compat/eigen_min.hpp"

namespace wave {
#ifndef EIGEN_TYPEDEF
#define EIGEN_TYPEDEF
typedef Eigen::Vector2d Vec2;
typedef Eigen::Vector3d Vec3;
typedef Eigen::Vector4d Vec4;
typedef Eigen::Matrix<double, 6, 1> Vec6;
typedef Eigen::Matrix2d Mat2;
typedef Eigen::Matrix3d Mat3;
typedef Eigen::Matrix4d Mat4;
typedef Eigen::Matrix<double, 6, 6> Mat6;
typedef Eigen::Affine3d Affine3;
#endif
}  // namespace wave
#endif
