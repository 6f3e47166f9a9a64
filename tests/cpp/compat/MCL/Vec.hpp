// Adapter for the boundary proof (tests/cpp/build_reference_test.sh): the reference's own test includes "MCL/Vec.hpp" from mclscene, a
// dependency that is absent from /root/reference (deps/mclscene is an empty submodule).  The test uses two names of it -- mcl::Vec3d and
// mcl::Vec4i, Eigen typedefs in mclscene -- which this header provides over the Eigen found at build time.  It is part of THIS repository's
// test of ITS OWN mirror (admm-elastic_amd/host), not a piece of a reference build.
#ifndef ADMM_COMPAT_MCL_VEC_HPP
#define ADMM_COMPAT_MCL_VEC_HPP 1
#include <Eigen/Dense>
namespace mcl {
typedef Eigen::Matrix<double, 3, 1> Vec3d;
typedef Eigen::Matrix<float, 3, 1> Vec3f;
typedef Eigen::Matrix<int, 3, 1> Vec3i;
typedef Eigen::Matrix<int, 4, 1> Vec4i;
typedef Eigen::Matrix<double, 4, 1> Vec4d;
}
#endif
