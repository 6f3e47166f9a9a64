// LinearSolver.hpp -- the interface of the global-step solvers and the prefactored solve (reference: src/LinearSolver.hpp;
// UzawaCG and NodalMultiColorGS: UzawaCG.hpp, NodalMultiColorGS.hpp).  The objects carry the reference's public tuning members; the arithmetic
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
    void push_params();      // the object's public tuning members -> the context (called before every solve / step)
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

} // namespace admm

// (the reference's users include LinearSolver.hpp and get the solvers it needs; the classes themselves live in their own headers,
// as in src/)
#include "UzawaCG.hpp"
#include "NodalMultiColorGS.hpp"
#endif
