// ExplicitForce.hpp -- explicit (pre-loop) forces, reference src/ExplicitForce.hpp:30-51.  Applied on the
// host before the state is uploaded, exactly where Solver::step calls them (src/Solver.cpp:54): they are
// O(n_tris) work outside the ADMM loop.
#ifndef ADMM_EXPLICITFORCE_HPP
#define ADMM_EXPLICITFORCE_HPP 1

#include <cmath>
#include <vector>
#include "MiniLinAlg.hpp"

namespace admm {
class ExplicitForce {
public:
    virtual ~ExplicitForce() {}
    virtual void project(double dt, VecX &x, VecX &v, VecX &masses) const = 0;
};

// Wind on a set of triangles (src/ExplicitForce.hpp:39-46, src/ExplicitForce.cpp:47-104; Wejchert & Haumann,
// "Animation aerodynamics", 1991): for every triangle the velocity relative to the wind, projected on the unit
// normal, gives the force  -alpha_n area v_n |v_n| n  (alpha_n = 1000), which is scaled by 0.33 dt and ADDED TO THE
// VELOCITY of each of the three nodes (the reference does not divide by the node mass; kept).
class WindForce : public ExplicitForce {
public:
    // Input is a list of all triangles the wind force affects (3 node indices per triangle).
    WindForce(std::vector<int> &tris_) : tris(tris_), direction(0, 0, 0) {}
    void project(double dt, VecX &x, VecX &v, VecX &masses) const {
        (void)masses;
        const int n_tris = (int)tris.size() / 3;
        for (int i = 0; i < n_tris; ++i) {
            const int idx[3] = {tris[i * 3 + 0] * 3, tris[i * 3 + 1] * 3, tris[i * 3 + 2] * 3};
            const Vec3 curr_v = (Vec3(v[idx[0]], v[idx[0] + 1], v[idx[0] + 2]) + Vec3(v[idx[1]], v[idx[1] + 1], v[idx[1] + 2]) +
                                 Vec3(v[idx[2]], v[idx[2] + 1], v[idx[2] + 2])) * (1.0 / 3.0);
            const Vec3 v_r = curr_v - direction;
            const Vec3 p0(x[idx[0]], x[idx[0] + 1], x[idx[0] + 2]), p1(x[idx[1]], x[idx[1] + 1], x[idx[1] + 2]), p2(x[idx[2]], x[idx[2] + 1], x[idx[2] + 2]);
            const Vec3 n = (p1 - p0).cross(p2 - p0);
            const double len = n.norm();
            if (!(len > 0.0)) continue;   // degenerate triangle: no area, no force (the reference would produce NaN)
            const Vec3 normal = n * (1.0 / len);
            const double area = 0.5 * len, alpha_n = 1000.0;
            const double v_n = normal.dot(v_r);
            const Vec3 force = normal * (-alpha_n * area * v_n * std::fabs(v_n) * 0.33 * dt);
            for (int j = 0; j < 3; ++j)
                for (int a = 0; a < 3; ++a) v[idx[j] + a] += force[a];
        }
    }
    std::vector<int> tris;
    Vec3 direction;
};
} // namespace admm
#endif
