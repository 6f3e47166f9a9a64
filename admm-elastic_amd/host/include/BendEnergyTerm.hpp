// BendEnergyTerm.hpp -- bending term for cloth.  The reference's README lists "bending force" as a TODO (README.md:23-28) and ships no code for
// it; this is the term in the mould of TriEnergyTerm (src/TriEnergyTerm.{hpp,cpp}): one EnergyTerm per HINGE (an interior edge v0 v1 and the
// two vertices v2, v3 opposite to it), dim 3, D-block = (c0, c1, c2, c3) (x) I3 with the cotangent stencil of the REST shape (Bergou et al.
// 2006, "A Quadratic Bending Model for Inextensible Surfaces": D_i x = sum_k c_k x_k is the hinge's discrete mean-curvature normal, zero for
// any flat configuration), energy E(z) = stiffness / 2 |z|^2 with stiffness = k_bend * 3 / (A0 + A1), weight = sqrt(stiffness) -- so that the
// prox is z = q / 2, the idiom of src/TriEnergyTerm.cpp:77-83.  The arithmetic runs in k_local_bends (csrc/kernels.hpp).
#ifndef ADMM_BENDENERGYTERM_HPP
#define ADMM_BENDENERGYTERM_HPP 1

#include "EnergyTerm.hpp"

namespace admm {

class BendEnergyTerm : public EnergyTerm {
public:
    // hinge = (v0, v1, v2, v3): v0 v1 the shared edge, v2 / v3 the opposite vertices; verts = their four REST positions
    BendEnergyTerm(const Vec4i &hinge, const std::vector<Vec3> &verts, double k_bend);
    // ... or from a stencil computed elsewhere (create_bends_from_mesh)
    BendEnergyTerm(const Vec4i &hinge, const double *coef4, double rest_area, double k_bend);
    int get_dim() const { return 3; }
    double get_weight() const { return weight; }
    bool flatten(FlatTerm &out) const;
    const double *stencil() const { return coef; }
protected:
    void get_reduction(std::vector<Triplet> &triplets) {
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 3; ++j) triplets.emplace_back(j, 3 * hinge[k] + j, coef[k]);
    }
    double energy(const VecX &F) { return 0.5 * stiffness * (F[0] * F[0] + F[1] * F[1] + F[2] * F[2]); }
    double gradient(const VecX &F, VecX &grad) { grad.resize(3); for (int j = 0; j < 3; ++j) grad[j] = stiffness * F[j]; return energy(F); }
    Vec4i hinge;
    double coef[4], area, stiffness, weight;
};

// One BendEnergyTerm per interior edge of a triangle mesh, like create_tris_from_mesh creates the stretch terms (src/TriEnergyTerm.hpp:31-46).
// Returns the number of hinges.  (Defined in Solver.cpp: the hinges come from admm_host_bend_hinges.)
int create_bends_from_mesh_d(std::vector<std::shared_ptr<EnergyTerm> > &energyterms, const double *verts, int n_verts, const int *inds, int n_tris,
                             double k_bend, int vertex_offset);
template <typename IN_SCALAR>
inline int create_bends_from_mesh(std::vector<std::shared_ptr<EnergyTerm> > &energyterms, const IN_SCALAR *verts, int n_verts, const int *inds,
                                  int n_tris, double k_bend, const int vertex_offset) {
    std::vector<double> v(3 * (size_t)n_verts);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (double)verts[i];
    return create_bends_from_mesh_d(energyterms, v.data(), n_verts, inds, n_tris, k_bend, vertex_offset);
}

} // namespace admm
#endif
