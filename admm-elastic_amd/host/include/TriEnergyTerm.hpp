// TriEnergyTerm.hpp -- cloth triangle term (reference: src/TriEnergyTerm.{hpp,cpp}).
#ifndef ADMM_TRIENERGYTERM_HPP
#define ADMM_TRIENERGYTERM_HPP 1

#include "EnergyTerm.hpp"

namespace admm {

class TriEnergyTerm : public EnergyTerm {
public:
    TriEnergyTerm(const Vec3i &tri, const std::vector<Vec3> &verts, const Lame &lame); // src/TriEnergyTerm.cpp:29-52
    int get_dim() const { return 6; }
    double get_weight() const { return weight; }
    bool flatten(FlatTerm &out) const;
protected:
    void get_reduction(std::vector<Triplet> &triplets);  // src/TriEnergyTerm.cpp:54-69
    double energy(const VecX &F);                         // src/TriEnergyTerm.cpp:104-114
    double gradient(const VecX &F, VecX &grad);
    Vec3i tri;
    Lame lame;
    double area, weight;
    double rest_pose[4]; // column-major 2x2
};

// src/TriEnergyTerm.hpp:31-46
template <typename IN_SCALAR, typename TYPE>
inline void create_tris_from_mesh(std::vector<std::shared_ptr<EnergyTerm> > &energyterms, const IN_SCALAR *verts,
                                  const int *inds, int n_tris, const Lame &lame, const int vertex_offset) {
    for (int i = 0; i < n_tris; ++i) {
        Vec3i tri(inds[i * 3 + 0], inds[i * 3 + 1], inds[i * 3 + 2]);
        std::vector<Vec3> tv;
        for (int c = 0; c < 3; ++c) tv.push_back(Vec3(verts[tri[c] * 3 + 0], verts[tri[c] * 3 + 1], verts[tri[c] * 3 + 2]));
        for (int c = 0; c < 3; ++c) tri[c] += vertex_offset;
        energyterms.emplace_back(std::make_shared<TYPE>(tri, tv, lame));
    }
}

} // namespace admm
#endif
