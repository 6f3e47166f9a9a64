// DynamicObject.hpp -- admm::TetMeshCollision of the MI355X build (reference: src/DynamicObject.hpp:31-121).
// The reference object owns two mclscene AABB trees and answers signed_distance() on the host; here it only carries the
// description (rest vertices, tets, surface faces, vertex offset) that Solver::initialize hands to
// admm_hip_add_dynamic_tetmesh -- trees, refit and queries live on the GPU (csrc/dyn_collide.hpp).
#ifndef ADMM_DYNAMICCOLLISION_HPP
#define ADMM_DYNAMICCOLLISION_HPP 1

#include <memory>
#include <stdexcept>
#include <vector>
#include "Collider.hpp"
#include "Meshes.hpp"

namespace admm {

class TetMeshCollision : public DynamicCollision {
public:
    // Constructor with vertex offset (to index into the global vertex array) -- src/DynamicObject.hpp:46-64
    TetMeshCollision(const std::shared_ptr<TetMesh> mesh, int v_offset) : vert_offset(v_offset) {
        if (mesh->faces.size() == 0) throw std::runtime_error("**TetMeshCollision Error: TetMesh needs surface faces");
        mesh_faces = mesh->faces;
        mesh_rest_verts = mesh->vertices;
        mesh_tets = mesh->tets;
    }
    bool flatten(DynFlat &f) const override {
        f.vert_offset = vert_offset;
        f.rest.resize(3 * mesh_rest_verts.size());
        for (size_t i = 0; i < mesh_rest_verts.size(); ++i) for (int a = 0; a < 3; ++a) f.rest[3 * i + a] = mesh_rest_verts[i][a];
        f.tets.resize(4 * mesh_tets.size());
        for (size_t i = 0; i < mesh_tets.size(); ++i) for (int a = 0; a < 4; ++a) f.tets[4 * i + a] = mesh_tets[i][a];
        f.faces.resize(3 * mesh_faces.size());
        for (size_t i = 0; i < mesh_faces.size(); ++i) for (int a = 0; a < 3; ++a) f.faces[3 * i + a] = mesh_faces[i][a];
        return true;
    }

private:
    int vert_offset;
    std::vector<Vec4i> mesh_tets;          // local indices (the reference stores them with the offset added, :60-62)
    std::vector<Vec3i> mesh_faces;
    std::vector<Vec3> mesh_rest_verts;
};

} // namespace admm
#endif
