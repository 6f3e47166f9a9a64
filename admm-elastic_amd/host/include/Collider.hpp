// Collider.hpp -- collision interfaces (reference: src/Collider.hpp).  On the MI355X build passive objects
// are evaluated inside the HIP kernels, so only the analytic obstacles (PassiveObject.hpp: Floor, Sphere)
// are accepted by Solver::initialize; the interfaces are kept so scene code compiles unchanged.
#ifndef ADMM_COLLIDER_HPP
#define ADMM_COLLIDER_HPP 1

#include <limits>
#include <memory>
#include <vector>
#include "MiniLinAlg.hpp"

namespace admm {

// src/Collider.hpp:32-61.  Self collision needs the BVH of the absent mclscene: out of scope.
class DynamicCollision {
public:
    virtual ~DynamicCollision() {}
};

// src/Collider.hpp:66-83
class PassiveCollision {
public:
    struct Payload {
        int vert_idx; double dx; Vec3 point, normal;
        Payload(int idx) : vert_idx(idx), dx(std::numeric_limits<double>::max()) {}
    };
    virtual ~PassiveCollision() {}
    virtual void signed_distance(const Vec3 &x, Payload &p) const = 0;
    // GPU description: kind (ADMM_OBJ_*) + 4 parameters; false = no kernel for this obstacle type
    virtual bool flatten(int &kind, double *params4) const { (void)kind; (void)params4; return false; }
};

// src/Collider.hpp:88-135 (the hit lists live on the device)
class Collider {
public:
    void add_passive_obj(std::shared_ptr<PassiveCollision> obj) { passive_objs.emplace_back(obj); }
    void add_dynamic_obj(std::shared_ptr<DynamicCollision> obj) { dynamic_objs.emplace_back(obj); }
    std::vector<std::shared_ptr<PassiveCollision> > passive_objs;
    std::vector<std::shared_ptr<DynamicCollision> > dynamic_objs;
};

} // namespace admm
#endif
