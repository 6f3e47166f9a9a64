// AddMeshes.hpp -- glue between the mesh layer (Meshes.hpp) and admm::Solver, mirroring the helper functions
// the reference's samples use (samples/utils/AddMeshes.hpp:43-62, 97-235): same names, argument meaning,
// defaults and error messages, on this repository's own mesh types.
#ifndef ADMM_ADDMESHES_HPP
#define ADMM_ADDMESHES_HPP 1

#include <iostream>
#include <numeric>
#include "Meshes.hpp"
#include "Solver.hpp"
#include "DynamicObject.hpp"
#include "TetEnergyTerm.hpp"
#include "TriEnergyTerm.hpp"

namespace binding {

// Flags that can be added to the mesh->flags member (AddMeshes.hpp:56-61)
enum MeshFlags {
    NOSELFCOLLISION = 1 << 1,
    LINEAR = 1 << 2, // default when mesh->flags==0
    NEOHOOKEAN = 1 << 3,
    STVK = 1 << 4,
};

// Adds the mesh's vertices as solver nodes (masses: rubber, 1522 kg/m^3) and one energy term per tet.
// Defaults as in the reference: Lame::rubber(), linear model when no model flag is set.
inline void add_tetmesh(admm::Solver *solver, std::shared_ptr<admm::TetMesh> &mesh,
                        const admm::Lame &lame = admm::Lame::rubber(), bool verbose = true) {
    const int num_tet_verts = (int)mesh->vertices.size();
    const int prev_tet_verts = solver->m_x.rows() / 3;
    const int num_tets = (int)mesh->tets.size();
    std::vector<double> masses;
    mesh->weighted_masses(masses, 1522.0);
    for (double m : masses)
        if (m <= 0.0) throw std::runtime_error("TetMesh Error: Zero mass");
    std::vector<double> x(3 * (size_t)num_tet_verts), m3(3 * (size_t)num_tet_verts);
    for (int i = 0; i < num_tet_verts; ++i)
        for (int a = 0; a < 3; ++a) { x[3 * i + a] = mesh->vertices[i][a]; m3[3 * i + a] = masses[i]; }
    solver->add_nodes(x.data(), m3.data(), num_tet_verts);
    if (!(mesh->flags & NOSELFCOLLISION)) {
        // Add a dynamic collider (samples/utils/AddMeshes.hpp:125-131)
        mesh->need_faces();
        std::shared_ptr<admm::TetMeshCollision> collision_mesh(new admm::TetMeshCollision(mesh, prev_tet_verts));
        solver->add_dynamic_collider(collision_mesh);
        std::vector<int> surf;
        mesh->surface_inds(surf);
        for (int s : surf) solver->surface_inds.emplace_back(s + prev_tet_verts);
    }
    std::vector<int> inds(4 * (size_t)num_tets);
    for (int t = 0; t < num_tets; ++t) for (int c = 0; c < 4; ++c) inds[4 * t + c] = mesh->tets[t][c];
    if ((mesh->flags & LINEAR) || ((mesh->flags & ~NOSELFCOLLISION) == 0)) {
        admm::create_tets_from_mesh<double, admm::TetEnergyTerm>(solver->energyterms, x.data(), inds.data(), num_tets, lame, prev_tet_verts);
    } else if (mesh->flags & NEOHOOKEAN) {
        admm::create_tets_from_mesh<double, admm::NeoHookeanTet>(solver->energyterms, x.data(), inds.data(), num_tets, lame, prev_tet_verts);
    } else if (mesh->flags & STVK) {
        admm::create_tets_from_mesh<double, admm::StVKTet>(solver->energyterms, x.data(), inds.data(), num_tets, lame, prev_tet_verts);
    }
    if (verbose) {
        std::cout << "Added mesh: "
                  << "\n\tmass: " << std::accumulate(masses.begin(), masses.end(), 0.0) << "kg"
                  << "\n\tvertices: " << num_tet_verts << "\n\ttets: " << num_tets
                  << "\n\t(total) verts: " << solver->m_x.size() / 3 << std::endl;
    }
}

// Triangle mesh: unit area density, strain-limited linear elastic triangles (the only triangle model).
inline void add_trimesh(admm::Solver *solver, std::shared_ptr<admm::TriangleMesh> &mesh,
                        const admm::Lame &lame = admm::Lame::rubber(), bool verbose = true) {
    const int num_tri_verts = (int)mesh->vertices.size();
    const int prev_tri_verts = solver->m_x.rows() / 3;
    const int num_tris = (int)mesh->faces.size();
    std::vector<double> masses;
    mesh->weighted_masses(masses, 1.0);
    for (double m : masses)
        if (m <= 0.0) throw std::runtime_error("TriMesh Error: Zero mass");
    std::vector<double> x(3 * (size_t)num_tri_verts), m3(3 * (size_t)num_tri_verts);
    for (int i = 0; i < num_tri_verts; ++i)
        for (int a = 0; a < 3; ++a) { x[3 * i + a] = mesh->vertices[i][a]; m3[3 * i + a] = masses[i]; }
    solver->add_nodes(x.data(), m3.data(), num_tri_verts);
    std::vector<int> inds(3 * (size_t)num_tris);
    for (int t = 0; t < num_tris; ++t) for (int c = 0; c < 3; ++c) inds[3 * t + c] = mesh->faces[t][c];
    if ((mesh->flags & LINEAR) || ((mesh->flags & ~NOSELFCOLLISION) == 0)) {
        admm::create_tris_from_mesh<double, admm::TriEnergyTerm>(solver->energyterms, x.data(), inds.data(), num_tris, lame, prev_tri_verts);
    } else {
        throw std::runtime_error("**binding::add_trimesh Error: Unknown triangle mesh material type");
    }
    if (verbose) {
        std::cout << "Added mesh: "
                  << "\n\tmass: " << std::accumulate(masses.begin(), masses.end(), 0.0) << "kg"
                  << "\n\tvertices: " << num_tri_verts << "\n\ttris: " << num_tris
                  << "\n\t(total) verts: " << solver->m_x.size() / 3 << std::endl;
    }
}

} // namespace binding

// Used for pinning portions of a mesh (AddMeshes.hpp:64-91):  GrabbySphere gs(center, radius); gs.get_indices(solver.m_x, pins);
class GrabbySphere {
public:
    admm::Vec3 c; // center
    double r;     // radius
    GrabbySphere(admm::Vec3 c_, double r_) : c(c_), r(r_) {}
    // Appends the indices of the nodes inside the sphere
    void get_indices(const admm::VecX &x, std::vector<int> &inds) const {
        const int n_verts = x.size() / 3;
        for (int i = 0; i < n_verts; ++i)
            if ((x.segment<3>(i * 3) - c).norm() < r) inds.push_back(i);
    }
};

#endif
