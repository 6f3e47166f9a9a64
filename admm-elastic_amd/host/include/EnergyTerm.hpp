// EnergyTerm.hpp -- admm::Lame and the abstract admm::EnergyTerm of the MI355X build.
// Mirrors the reference's src/EnergyTerm.hpp (Lame :34-59, EnergyTerm :65-107, update :130-140) with the
// per-term arithmetic executed by the HIP kernels behind include/admm_hip.h.  A term describes itself to
// the solver through flatten(); user-defined subclasses that cannot (no GPU kernel) are rejected by
// Solver::initialize -- there is no CPU fallback on the hot path.
#ifndef ADMM_ENERGYTERM_HPP
#define ADMM_ENERGYTERM_HPP 1

#include <iostream>      // (the reference's EnergyTerm.hpp brings it to every user, src/EnergyTerm.hpp:26)
#include <memory>
#include <vector>
#include "MiniLinAlg.hpp"

namespace admm {

// src/EnergyTerm.hpp:34-59
class Lame {
public:
    static Lame rubber() { return Lame(10000000, 0.499); }
    static Lame soft_rubber() { return Lame(10000000, 0.399); }
    static Lame very_soft_rubber() { return Lame(1000000, 0.299); }
    double mu, lambda;
    double bulk_modulus() const { return lambda + (2.0 / 3.0) * mu; }
    double limit_min, limit_max; // hard strain limiting (triangles), default: none
    Lame(double youngs, double poisson)
        : mu(youngs / (2.0 * (1.0 + poisson))), lambda(youngs * poisson / ((1.0 + poisson) * (1.0 - 2.0 * poisson))),
          limit_min(-100.0), limit_max(100.0) {}
    Lame() : mu(0), lambda(0), limit_min(-100.0), limit_max(100.0) {}
};

// What Solver::initialize needs from one term, in the layout of admm_hip_desc.
struct FlatTerm {
    enum Type { TET = 0, TRI = 1, PIN = 2, BEND = 3 };   // BEND: idx = the hinge's four vertices, mat[0..3] = its stencil, k = its stiffness
    int type;
    int idx[4];
    double mat[9];      // tet: edges_inv (col-major 3x3); tri: rest_pose (col-major 2x2)
    double weight;
    int kind;           // ADMM_TET_*
    double mu, lambda, k, limit_min, limit_max;
    double kappa = 0.0; // SplineTet: compression term of the xu:: spline (src/XuSpline.hpp:43-45)
    const void *user_spline = nullptr;   // SplineTet with a user-defined xu::Spline (kind ADMM_TET_SPLINE_TABLE): the object to tabulate
    double pin[3];
    int active;
    double nrm[3];      // PIN: non-zero = a slide pin (SlidePin): the vertex may move in the plane through `pin` with this normal
};

class EnergyTerm {
public:
    virtual ~EnergyTerm();
    // Appends this term's rows of the reduction matrix D and `get_dim()` copies of its weight
    // (src/EnergyTerm.hpp:113-128); also fixes the term's first row (g_index).
    void get_reduction(std::vector<Triplet> &triplets, std::vector<double> &weights);
    // Local step of this single term (src/EnergyTerm.hpp:130-140), executed on the GPU through a
    // one-term context: z_i = prox(D_i x + u_i), u_i += D_i x - z_i.
    void update(const SparseMat &D, const VecX &x, VecX &z, VecX &u);
    // Debugging aids of the reference (host arithmetic, not on the hot path).
    double energy(const SparseMat &D, const VecX &x);
    double gradient(const SparseMat &D, const VecX &x, VecX &grad);

    virtual int get_dim() const = 0;
    virtual double get_weight() const = 0;
    // Describe the term for the GPU; return false if the type has no kernel.
    virtual bool flatten(FlatTerm &out) const { (void)out; return false; }
    int global_index() const { return g_index; }

protected:
    EnergyTerm() : g_index(0), one_ctx_(nullptr) {}
    virtual void get_reduction(std::vector<Triplet> &triplets) = 0;
    virtual double energy(const VecX &F) = 0;
    virtual double gradient(const VecX &F, VecX &grad) = 0;

private:
    int g_index;
    void *one_ctx_; // cached one-term admm_hip_ctx for update()
};

// signed SVD on the host (used by energy() only): F col-major -> U, S, V with U,V in SO(3)
void host_signed_svd3(const double *F, double *U, double *S, double *V);

} // namespace admm
#endif
