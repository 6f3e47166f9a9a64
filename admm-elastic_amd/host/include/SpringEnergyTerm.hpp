// SpringEnergyTerm.hpp -- the hard pin "spring" (reference: src/SpringEnergyTerm.hpp:31-73).
#ifndef ADMM_SPRINGENERGYTERM_HPP
#define ADMM_SPRINGENERGYTERM_HPP 1

#include "EnergyTerm.hpp"

namespace admm {

class SpringPin : public EnergyTerm {
public:
    SpringPin(int idx_, const Vec3 &pin_) : idx(idx_), pin(pin_), active(true) {
        weight = std::sqrt(Lame::rubber().bulk_modulus() * 2.0); // :47-52
    }
    int get_dim() const { return 6; }  // :42 (rows 3..5 are never populated)
    double get_weight() const { return weight; }
    void set_pin(const Vec3 &p) { pin = p; }
    void set_active(bool a) { active = a; }
    int vertex() const { return idx; }
    const Vec3 &location() const { return pin; }
    bool is_active() const { return active; }
    bool flatten(FlatTerm &out) const;
protected:
    void get_reduction(std::vector<Triplet> &triplets) {
        for (int j = 0; j < 3; ++j) triplets.emplace_back(j, 3 * idx + j, 1.0); // :54-59
    }
    double energy(const VecX &) { throw std::runtime_error("**SpringPin Error: Energy not implemented"); }
    double gradient(const VecX &, VecX &) { throw std::runtime_error("**SpringPin Error: No gradient for hard constraint"); }
    int idx;
    Vec3 pin;
    bool active;
    double weight;
};

// SLIDE constraint (the reference's README lists "slide constraints" as a TODO, README.md:23-28; no reference code).  A SpringPin whose prox
// projects D_i x + u_i onto the PLANE through `pin` with normal `normal` instead of onto the point (src/SpringEnergyTerm.hpp:61): the
// vertex slides freely in the plane.  Same D-block, weight and dim-6 row layout as SpringPin.
class SlidePin : public SpringPin {
public:
    SlidePin(int idx_, const Vec3 &point_, const Vec3 &normal_) : SpringPin(idx_, point_), normal(normal_) {
        const double l = normal.norm();
        if (!(l > 0.0)) throw std::runtime_error("**SlidePin Error: zero normal");
        normal = normal * (1.0 / l);
    }
    void set_normal(const Vec3 &n) { const double l = n.norm(); if (!(l > 0.0)) throw std::runtime_error("**SlidePin Error: zero normal"); normal = n * (1.0 / l); }
    const Vec3 &plane_normal() const { return normal; }
    bool flatten(FlatTerm &out) const;
protected:
    Vec3 normal;
};

} // namespace admm
#endif
