// TetEnergyTerm.hpp -- tetrahedral energy terms (reference: src/TetEnergyTerm.{hpp,cpp}).
#ifndef ADMM_TETENERGYTERM_HPP
#define ADMM_TETENERGYTERM_HPP 1

#include <memory>
#include "EnergyTerm.hpp"
#include "XuSpline.hpp"

namespace admm {

// Linear ("ARAP-like") tet -- src/TetEnergyTerm.hpp:57-81, ctor src/TetEnergyTerm.cpp:31-48
class TetEnergyTerm : public EnergyTerm {
public:
    TetEnergyTerm(const Vec4i &tet, const std::vector<Vec3> &verts, const Lame &lame);
    int get_dim() const { return 9; }
    double get_weight() const { return weight; }
    virtual bool flatten(FlatTerm &out) const;
    virtual int kind() const { return 0; } // ADMM_TET_LINEAR
protected:
    void get_reduction(std::vector<Triplet> &triplets);
    double energy(const VecX &F);
    double gradient(const VecX &F, VecX &grad);
    Vec4i tet;
    Lame lame;
    double volume, weight;
    double edges_inv[9]; // column-major
};

// src/TetEnergyTerm.hpp:116-136
class NeoHookeanTet : public TetEnergyTerm {
public:
    NeoHookeanTet(const Vec4i &tet, const std::vector<Vec3> &verts, const Lame &lame) : TetEnergyTerm(tet, verts, lame) {}
    int kind() const { return 1; }
protected:
    double energy(const VecX &F);
};

// src/TetEnergyTerm.hpp:142-164
class StVKTet : public TetEnergyTerm {
public:
    StVKTet(const Vec4i &tet, const std::vector<Vec3> &verts, const Lame &lame) : TetEnergyTerm(tet, verts, lame) {}
    int kind() const { return 2; }
protected:
    double energy(const VecX &F);
};

// STABLE NEO-HOOKEAN (the reference's README lists it as a TODO, README.md:23-28; it would be one more HyperElasticTet beside
// NeoHookeanTet, src/TetEnergyTerm.hpp:116-136).  Smith, de Goes, Kim 2018: Psi = mu_s/2 (I_C - 3) + la_s/2 (J - alpha)^2 - mu_s/2 log(I_C + 1),
// mu_s = 4/3 mu, la_s = lambda + 5/6 mu, alpha = 1 + 3 mu_s / (4 la_s); finite and smooth for inverted elements (ADMM_TET_STABLE_NH).
class StableNeoHookeanTet : public TetEnergyTerm {
public:
    StableNeoHookeanTet(const Vec4i &tet, const std::vector<Vec3> &verts, const Lame &lame) : TetEnergyTerm(tet, verts, lame) {}
    int kind() const { return 7; }
protected:
    double energy(const VecX &F);
};

// src/TetEnergyTerm.hpp:176-206.  Defaults to xu::NeoHookean(mu, lambda, 0) like the reference (:191-195); the second
// constructor takes any xu::Spline (XuSpline.hpp): the reference's three with or without compression term run closed-form
// kernels, a user-defined one is tabulated by Solver::initialize and runs the table kernel (ADMM_TET_SPLINE_TABLE).
class SplineTet : public NeoHookeanTet {
public:
    SplineTet(const Vec4i &tet, const std::vector<Vec3> &verts, const Lame &lame)
        : NeoHookeanTet(tet, verts, lame), spline(std::make_shared<xu::NeoHookean>(lame.mu, lame.lambda, 0.0)) {}
    SplineTet(const Vec4i &tet, const std::vector<Vec3> &verts, const Lame &lame, std::shared_ptr<xu::Spline> spline_)
        : NeoHookeanTet(tet, verts, lame), spline(spline_) {}
    int kind() const { int kd = 3; double m, l, kp; return spline->flatten(kd, m, l, kp) ? kd : 6; }
    bool flatten(FlatTerm &out) const;
    std::shared_ptr<xu::Spline> spline;
protected:
    double energy(const VecX &F);   // src/TetEnergyTerm.cpp:243-247 on the signed stretches, times the volume
};

// src/TetEnergyTerm.hpp:35-51
template <typename IN_SCALAR, typename TYPE>
inline void create_tets_from_mesh(std::vector<std::shared_ptr<EnergyTerm> > &energyterms, const IN_SCALAR *verts,
                                  const int *inds, int n_tets, const Lame &lame, const int vertex_offset) {
    for (int i = 0; i < n_tets; ++i) {
        Vec4i tet(inds[i * 4 + 0], inds[i * 4 + 1], inds[i * 4 + 2], inds[i * 4 + 3]);
        std::vector<Vec3> tv;
        for (int c = 0; c < 4; ++c) tv.push_back(Vec3(verts[tet[c] * 3 + 0], verts[tet[c] * 3 + 1], verts[tet[c] * 3 + 2]));
        for (int c = 0; c < 4; ++c) tet[c] += vertex_offset;
        energyterms.emplace_back(std::make_shared<TYPE>(tet, tv, lame));
    }
}

} // namespace admm
#endif
