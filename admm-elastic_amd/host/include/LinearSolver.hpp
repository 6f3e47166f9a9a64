// LinearSolver.hpp -- global-step solvers (reference: src/LinearSolver.hpp, src/UzawaCG.hpp,
// src/NodalMultiColorGS.hpp).  The objects carry the reference's public tuning members; the arithmetic
// runs in the HIP kernels of the Solver's context, which the Solver attaches after initialize().
#ifndef ADMM_LINEARSOLVER_HPP
#define ADMM_LINEARSOLVER_HPP 1

#include <memory>
#include "ConstraintSet.hpp"

namespace admm {

class LinearSolver {
public:
    virtual ~LinearSolver() {}
    // src/LinearSolver.hpp:43: the GPU context owns the matrix (assembled from the energy terms); a solver
    // object cannot be re-pointed at an arbitrary matrix.
    virtual void update_system(const SparseMat &A_) { A = A_; }
    // src/LinearSolver.hpp:46: x is in/out (warm start); returns the inner iteration count.
    virtual int solve(VecX &x, const VecX &b);
    virtual int kind() const = 0; // ADMM_LS_*
    const SparseMat &matrix() const { return A; }
    void attach(void *ctx) { ctx_ = ctx; }
protected:
    LinearSolver() : ctx_(nullptr) {}
    SparseMat A;
    void *ctx_;
};

// src/LinearSolver.hpp:59-92 -- the prefactored LDLT becomes a GPU PCG iterated to pcg_tol
class LDLTSolver : public LinearSolver {
public:
    int pcg_max_iters; double pcg_tol;
    LDLTSolver() : pcg_max_iters(500), pcg_tol(1e-12) {} // stands for an exact factorisation: converge tightly
    int kind() const { return 0; }
};

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
