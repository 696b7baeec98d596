// eigen_min.hpp -- the handful of Eigen types libwave's matcher API exposes
// (Eigen::Affine3d, Eigen::Matrix<double,6,6>, ...), for builds where Eigen is not
// installed.  If <Eigen/Geometry> is available it is used instead and this file
// defines nothing.  Only what wave::Matcher's public surface and the reference's
// tests touch is provided: construction, Identity()/Zero(), operator(), comma
// initialisation, + - *, .norm(), .matrix(), .translation(), .rotation().
#pragma once

#if defined(WAVE_MATCHING_USE_SYSTEM_EIGEN) || __has_include(<Eigen/Geometry>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/StdVector>
#else

#include <cmath>
#include <cstddef>
#include <memory>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

template <class M>
class CommaInit {
 public:
    CommaInit(M *m, double first) : m_(m), k_(0) { put(first); }
    CommaInit &operator,(double v) {
        put(v);
        return *this;
    }

 private:
    void put(double v) {
        if (k_ < M::Rows * M::Cols) {
            m_->coeffRef(k_ / M::Cols, k_ % M::Cols) = v;
            ++k_;
        }
    }
    M *m_;
    int k_;
};

template <typename Scalar, int R, int C>
class Matrix {
 public:
    static constexpr int Rows = R, Cols = C;
    Matrix() {
        for (int i = 0; i < R * C; ++i) d_[i] = Scalar(0);
    }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int, int) { return Matrix(); }
    static Matrix Identity() {
        Matrix m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = Scalar(1);
        return m;
    }
    static Matrix Identity(int, int) { return Identity(); }
    Scalar &operator()(int i, int j) { return d_[i * C + j]; }
    const Scalar &operator()(int i, int j) const { return d_[i * C + j]; }
    Scalar &operator()(int i) { return d_[i]; }
    const Scalar &operator()(int i) const { return d_[i]; }
    Scalar &operator[](int i) { return d_[i]; }
    const Scalar &operator[](int i) const { return d_[i]; }
    Scalar &coeffRef(int i, int j) { return d_[i * C + j]; }
    Scalar &x() { return d_[0]; }
    Scalar &y() { return d_[1]; }
    Scalar &z() { return d_[2]; }
    const Scalar &x() const { return d_[0]; }
    const Scalar &y() const { return d_[1]; }
    const Scalar &z() const { return d_[2]; }
    int rows() const { return R; }
    int cols() const { return C; }
    Scalar *data() { return d_; }             // NOTE: row-major (Eigen's default is column-major)
    const Scalar *data() const { return d_; }
    CommaInit<Matrix> operator<<(Scalar v) { return CommaInit<Matrix>(this, v); }
    Matrix operator+(const Matrix &o) const {
        Matrix r;
        for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] + o.d_[i];
        return r;
    }
    Matrix operator-(const Matrix &o) const {
        Matrix r;
        for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] - o.d_[i];
        return r;
    }
    Matrix operator*(Scalar s) const {
        Matrix r;
        for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] * s;
        return r;
    }
    template <int K>
    Matrix<Scalar, R, K> operator*(const Matrix<Scalar, C, K> &o) const {
        Matrix<Scalar, R, K> r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < K; ++j) {
                Scalar s = 0;
                for (int k = 0; k < C; ++k) s += (*this)(i, k) * o(k, j);
                r(i, j) = s;
            }
        return r;
    }
    Matrix<Scalar, C, R> transpose() const {
        Matrix<Scalar, C, R> r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j) r(j, i) = (*this)(i, j);
        return r;
    }
    Scalar norm() const {  // Frobenius
        Scalar s = 0;
        for (int i = 0; i < R * C; ++i) s += d_[i] * d_[i];
        return std::sqrt(s);
    }
    template <typename T2>
    Matrix<T2, R, C> cast() const {
        Matrix<T2, R, C> r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j) r(i, j) = static_cast<T2>((*this)(i, j));
        return r;
    }

 private:
    Scalar d_[R * C];
};

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 4, 4> Matrix4f;

enum TransformTraits { Affine = 2 };

// 4x4 homogeneous transform with an affine (bottom row 0 0 0 1) interpretation
class Affine3d {
 public:
    Affine3d() : m_(Matrix4d::Identity()) {}
    explicit Affine3d(const Matrix4d &m) : m_(m) {}
    static Affine3d Identity() { return Affine3d(); }
    Matrix4d &matrix() { return m_; }
    const Matrix4d &matrix() const { return m_; }
    Affine3d &operator=(const Matrix4d &m) {
        m_ = m;
        return *this;
    }

    // writable view of the translation column
    class TranslationRef {
     public:
        static constexpr int Rows = 3, Cols = 1;
        explicit TranslationRef(Matrix4d *m) : m_(m) {}
        double &coeffRef(int i, int) { return (*m_)(i, 3); }
        double &operator()(int i) { return (*m_)(i, 3); }
        double &x() { return (*m_)(0, 3); }
        double &y() { return (*m_)(1, 3); }
        double &z() { return (*m_)(2, 3); }
        CommaInit<TranslationRef> operator<<(double v) { return CommaInit<TranslationRef>(this, v); }
        operator Vector3d() const {
            Vector3d v;
            for (int i = 0; i < 3; ++i) v(i) = (*m_)(i, 3);
            return v;
        }

     private:
        Matrix4d *m_;
    };
    TranslationRef translation() { return TranslationRef(&m_); }
    Vector3d translation() const {
        Vector3d v;
        for (int i = 0; i < 3; ++i) v(i) = m_(i, 3);
        return v;
    }
    Matrix3d rotation() const {
        Matrix3d r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r(i, j) = m_(i, j);
        return r;
    }
    Matrix3d linear() const { return rotation(); }
    Affine3d operator*(const Affine3d &o) const { return Affine3d(m_ * o.m_); }
    Vector3d operator*(const Vector3d &p) const {
        Vector3d r;
        for (int i = 0; i < 3; ++i) r(i) = m_(i, 0) * p(0) + m_(i, 1) * p(1) + m_(i, 2) * p(2) + m_(i, 3);
        return r;
    }

 private:
    Matrix4d m_;
};

template <class T>
using aligned_allocator = std::allocator<T>;

}  // namespace Eigen
#endif
