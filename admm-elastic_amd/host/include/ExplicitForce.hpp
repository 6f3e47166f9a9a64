// ExplicitForce.hpp -- explicit (pre-loop) forces, reference src/ExplicitForce.hpp:30-51.  Applied on the
// host before the state is uploaded, exactly where Solver::step calls them (src/Solver.cpp:54); the
// reference's WindForce is outside the hot path.
#ifndef ADMM_EXPLICITFORCE_HPP
#define ADMM_EXPLICITFORCE_HPP 1

#include "MiniLinAlg.hpp"

namespace admm {
class ExplicitForce {
public:
    virtual ~ExplicitForce() {}
    virtual void project(double dt, VecX &x, VecX &v, VecX &masses) const = 0;
};
} // namespace admm
#endif
