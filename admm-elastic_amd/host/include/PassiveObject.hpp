// PassiveObject.hpp -- analytic obstacles (reference: src/PassiveObject.hpp:32-64).
#ifndef ADMM_PASSIVEOBJECT_HPP
#define ADMM_PASSIVEOBJECT_HPP 1

#include "Collider.hpp"

namespace admm {

class Floor : public PassiveCollision {
public:
    double m_y;
    Floor(double y) : m_y(y) {}
    void signed_distance(const Vec3 &x, Payload &p) const { // :37-43
        const double dx = x[1] - m_y;
        if (dx > p.dx) return;
        p.dx = dx; p.point = Vec3(x[0], m_y, x[2]); p.normal = Vec3(0, 1, 0);
    }
    bool flatten(int &kind, double *q) const { kind = 0; q[0] = m_y; q[1] = q[2] = q[3] = 0.0; return true; }
};

class Sphere : public PassiveCollision {
public:
    Vec3 center; double rad;
    Sphere(const Vec3 &c, double r) : center(c), rad(r) {}
    void signed_distance(const Vec3 &x, Payload &p) const { // :55-62
        Vec3 dir = x - center;
        const double l = dir.norm(), dx = l - rad;
        if (dx > p.dx) return;
        dir = dir * (1.0 / l);
        p.dx = dx; p.point = center + dir * rad; p.normal = dir;
    }
    bool flatten(int &kind, double *q) const { kind = 1; q[0] = center[0]; q[1] = center[1]; q[2] = center[2]; q[3] = rad; return true; }
};

// Not in the reference: a user-style PassiveCollision with a kernel of its own (ADMM_OBJ_PLANE) -- the half space n.x < d is solid.
class Plane : public PassiveCollision {
public:
    Vec3 n; double d;
    Plane(const Vec3 &normal, double offset) : n(normal * (1.0 / normal.norm())), d(offset / normal.norm()) {}
    void signed_distance(const Vec3 &x, Payload &p) const {
        const double dx = n.dot(x) - d;
        if (dx > p.dx) return;
        p.dx = dx; p.point = x - n * dx; p.normal = n;
    }
    bool flatten(int &kind, double *q) const { kind = 2; q[0] = n[0]; q[1] = n[1]; q[2] = n[2]; q[3] = d; return true; }
};

} // namespace admm
#endif
