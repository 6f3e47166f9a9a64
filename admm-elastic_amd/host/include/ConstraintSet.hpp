// ConstraintSet.hpp -- pins + collider + constraint stiffness (reference: src/ConstraintSet.hpp:28-52).
// The constraint matrix itself (make_matrix, :59-116) is assembled on the device per ADMM iteration.
#ifndef ADMM_CONSTRAINTSET_HPP
#define ADMM_CONSTRAINTSET_HPP 1

#include <unordered_map>
#include "Collider.hpp"

namespace admm {

class ConstraintSet {
public:
    double constraint_w;
    std::shared_ptr<Collider> collider;
    std::unordered_map<int, Vec3> pins; // index -> location
    std::unordered_map<int, std::pair<Vec3, Vec3> > slides; // index -> (point, unit normal): slide constraints (README.md:23-28 TODO; not in the reference)
    ConstraintSet() : constraint_w(1.0), collider(std::make_shared<Collider>()) {}
};

} // namespace admm
#endif
