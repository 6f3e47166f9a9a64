// UzawaCG.hpp -- the Schur-complement CG solver object (reference: src/UzawaCG.hpp).  Carries the reference's public tuning members
// (max_iters 20, m_tol 1e-10: UzawaCG.hpp:44-45) next to the tolerance of the GPU PCG that stands for the prefactored solve; the
// arithmetic -- constraint rows, cached columns of K^-1, the Schur CG as one persistent launch -- runs in the Solver's HIP context
// (csrc/admm_hip.hip: launch_uzawa, csrc/uz_persist.hpp).
#ifndef ADMM_UZAWACG_HPP
#define ADMM_UZAWACG_HPP 1

#include "LinearSolver.hpp"

namespace admm {

// src/UzawaCG.hpp:33-55
class UzawaCG : public LinearSolver {
public:
    int max_iters; double m_tol;
    int pcg_max_iters; double pcg_tol;
    std::shared_ptr<ConstraintSet> constraints;
    UzawaCG(std::shared_ptr<ConstraintSet> c) : max_iters(20), m_tol(1e-10), pcg_max_iters(500), pcg_tol(1e-10), constraints(c) {}
    UzawaCG() : UzawaCG(std::make_shared<ConstraintSet>()) {}
    int kind() const { return 2; }
};

} // namespace admm
#endif
