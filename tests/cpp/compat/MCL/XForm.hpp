// Adapter for the boundary proof (see MCL/Vec.hpp beside this file): mcl::XForm<T> and the two factory functions the reference's test calls
// (samples/tests/test_lineartet.cpp:88, :98, :139) -- an affine transform applied with `xform * point` and read with `xform(r, c)`.
#ifndef ADMM_COMPAT_MCL_XFORM_HPP
#define ADMM_COMPAT_MCL_XFORM_HPP 1
#include <cmath>
#include <Eigen/Dense>
#include <Eigen/Geometry>
namespace mcl {
template <typename T> using XForm = Eigen::Transform<T, 3, Eigen::Affine>;
namespace xform {
template <typename T> inline XForm<T> make_rot(T angle_deg, const Eigen::Matrix<T, 3, 1> &axis) {
    XForm<T> r; r.setIdentity();
    r.rotate(Eigen::AngleAxis<T>(angle_deg * T(M_PI / 180.0), axis.normalized()));
    return r;
}
template <typename T> inline XForm<T> make_scale(T x, T y, T z) {
    XForm<T> r; r.setIdentity();
    r.scale(Eigen::Matrix<T, 3, 1>(x, y, z));
    return r;
}
}
}
#endif
