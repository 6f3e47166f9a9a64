// Collider.hpp -- collision interfaces (reference: src/Collider.hpp).  On the MI355X build passive objects are evaluated inside
// the HIP kernels: Floor, Sphere and Plane (PassiveObject.hpp) analytically, ANY OTHER PassiveCollision subclass from a grid that
// Solver::initialize samples from the object's own signed_distance (Solver::obstacle_grid_*; include/admm_hip.h: ADMM_OBJ_GRID).
// Dynamic colliders: tet-mesh self-collision proxies (DynamicObject.hpp: TetMeshCollision).
#ifndef ADMM_COLLIDER_HPP
#define ADMM_COLLIDER_HPP 1

#include <limits>
#include <memory>
#include <vector>
#include "MiniLinAlg.hpp"

namespace admm {

// src/Collider.hpp:32-61.  On the MI355X build detection runs on the GPU (csrc/dyn_collide.hpp), so an object is
// accepted by Solver::initialize when it can describe itself as a tet mesh (TetMeshCollision, DynamicObject.hpp);
// update() / signed_distance() are kept so scene code that overrides them compiles, but they are never called.
class DynamicCollision {
public:
    struct Payload {                       // src/Collider.hpp:40-53
        int vert_idx; Vec4i self_tet; double dx; Vec3 normal; Vec3i face; Vec3 barys;
        Payload(int idx) : vert_idx(idx), self_tet(-1, -1, -1, -1), dx(std::numeric_limits<double>::max()), normal(0, 0, 0), face(-1, -1, -1), barys(0, 0, 0) {}
    };
    // flat description for admm_hip_add_dynamic_tetmesh (include/admm_hip.h)
    struct DynFlat { int vert_offset; std::vector<double> rest; std::vector<int> tets, faces; };
    virtual ~DynamicCollision() {}
    virtual void update(const VecX &x) { (void)x; }
    virtual void signed_distance(const Vec3 &x, Payload &p) const { (void)x; (void)p; }
    virtual bool flatten(DynFlat &f) const { (void)f; return false; }   // false = no GPU kernel for this object type
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
    // GPU description: kind (ADMM_OBJ_*) + 4 parameters; false = no analytic kernel: Solver::initialize samples the object on a grid
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
