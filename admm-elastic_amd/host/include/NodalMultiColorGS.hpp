// NodalMultiColorGS.hpp -- the nodal multi-colour Gauss-Seidel solver object (reference: src/NodalMultiColorGS.hpp).  Carries the
// reference's public tuning members (30 sweeps, tolerance 1e-10, omega 1.9: NodalMultiColorGS.hpp:40-43); the sweeps -- pins first,
// over-relaxed rows, plane projection of rows inside a passive obstacle, residual test per sweep -- run in the Solver's HIP context
// (csrc/gs_persist.hpp: one persistent launch per solve; csrc/kernels.hpp: k_gs_color* as fall-back).
#ifndef ADMM_NODALMULTICOLORGS_HPP
#define ADMM_NODALMULTICOLORGS_HPP 1

#include "LinearSolver.hpp"

namespace admm {

// src/NodalMultiColorGS.hpp:33-59
class NodalMultiColorGS : public LinearSolver {
public:
    int max_iters; double m_tol, m_omega;
    std::shared_ptr<ConstraintSet> constraints;
    NodalMultiColorGS(std::shared_ptr<ConstraintSet> c) : max_iters(30), m_tol(1e-10), m_omega(1.9), constraints(c) {}
    NodalMultiColorGS() : NodalMultiColorGS(std::make_shared<ConstraintSet>()) {}
    int kind() const { return 1; }
};

} // namespace admm
#endif
