// XuSpline.hpp -- the spline family of SplineTet (reference: src/XuSpline.hpp; Xu, Sin, Zhu, Barbic 2015,
// "Nonlinear Material Design Using Principal Stretches").  Psi = sum f(s_i) + sum g(s_i s_j) + h(s_0 s_1 s_2).
// The interface (xu::Spline with f / g / h and their derivatives) and the three named splines keep the reference's
// names and constructor arguments.  On the GPU each named spline with kappa = 0 IS one of the closed-form stretch models
// (NeoHookean -> NH, StVK -> StVK, CoRotated -> co-rotated linear), which is how SplineTet::flatten hands it to the
// kernels; the host-side f / g / h below only serve EnergyTerm::energy.  A USER-DEFINED subclass (any object with the six
// functions, as in the reference) is sampled once by Solver::initialize (admm_host_tabulate_spline over stretches in
// [table_min, table_max]) and evaluated on the device from its tables: no CPU fallback, no restriction to the three named splines.
#ifndef ADMM_XUSPLINE_HPP
#define ADMM_XUSPLINE_HPP 1

#include <cmath>

namespace admm {
namespace xu {

class Spline {
public:
    virtual ~Spline() {}
    virtual double f(double x) const = 0;
    virtual double g(double x) const = 0;
    virtual double h(double x) const = 0;
    virtual double df(double x) const = 0;
    virtual double dg(double x) const = 0;
    virtual double dh(double x) const = 0;
    // compression term of the paper's Eq. 16 and its derivative as the reference evaluates them (src/XuSpline.hpp:44-45)
    static double compress_term(double kappa, double x) { const double t = (1.0 - x) / 6.0; return kappa * t * t * t / 12.0; }
    static double d_compress_term(double kappa, double x) { const double t = (1.0 - x) / 6.0; return -kappa * t * t / 24.0; }
    // GPU description: ADMM_TET_SPLINE_* kind, the spline's Lame constants and compression term; false = not one of the named
    // splines: the solver tabulates it (ADMM_TET_SPLINE_TABLE)
    virtual bool flatten(int &kind, double &mu_out, double &lambda_out, double &kappa_out) const {
        (void)kind; (void)mu_out; (void)lambda_out; (void)kappa_out; return false;
    }
    // range of principal stretches the table of a user-defined spline covers (outside it the end nodes' quadratics continue)
    double table_min = 0.02, table_max = 50.0;
};

namespace detail {

// The three shipped splines differ only in the polynomial / logarithmic pieces below; one class evaluates all of them.
enum Model { kNeoHookean = 3, kStVK = 4, kCoRotated = 5 };   // = the ADMM_TET_SPLINE_* kinds of include/admm_hip.h

template <Model M>
class LameSpline : public Spline {
public:
    LameSpline(double mu_, double lambda_, double kappa_) : mu(mu_), lambda(lambda_), kappa(kappa_) {}
    const double mu, lambda, kappa;

    double f(double s) const {
        const double s2 = s * s;
        if (M == kNeoHookean) return mu * (s2 - 1.0) / 2.0;
        if (M == kStVK) return lambda * (s2 * s2 - 6.0 * s2 + 5.0) / 8.0 + mu * (s2 - 1.0) * (s2 - 1.0) / 4.0;
        return lambda * (s2 - 6.0 * s + 5.0) / 2.0 + mu * (s - 1.0) * (s - 1.0);
    }
    double df(double s) const {
        if (M == kNeoHookean) return mu * s;
        if (M == kStVK) return lambda * (s * s * s - 3.0 * s) / 2.0 + mu * s * (s * s - 1.0);
        return lambda * (s - 3.0) + 2.0 * mu * (s - 1.0);
    }
    double g(double p) const { return M == kNeoHookean ? 0.0 : M == kStVK ? lambda * (p * p - 1.0) / 4.0 : lambda * (p - 1.0); }
    double dg(double p) const { return M == kNeoHookean ? 0.0 : M == kStVK ? lambda * p / 2.0 : lambda; }
    double h(double J) const {
        double v = compress_term(kappa, J);
        if (M == kNeoHookean) { const double lJ = std::log(J); v += lJ * (lambda * lJ / 2.0 - mu); }
        return v;
    }
    double dh(double J) const {
        double v = d_compress_term(kappa, J);
        if (M == kNeoHookean) v += (lambda * std::log(J) - mu) / J;
        return v;
    }
    bool flatten(int &kind, double &mu_out, double &lambda_out, double &kappa_out) const {
        kind = (int)M; mu_out = mu; lambda_out = lambda; kappa_out = kappa;
        return true;
    }
};

} // namespace detail

// src/XuSpline.hpp:48-62, :64-82, :84-96 -- same names and (mu, lambda, kappa) constructors
struct NeoHookean : detail::LameSpline<detail::kNeoHookean> { NeoHookean(double mu_, double lambda_, double kappa_) : LameSpline(mu_, lambda_, kappa_) {} };
struct StVK : detail::LameSpline<detail::kStVK> { StVK(double mu_, double lambda_, double kappa_) : LameSpline(mu_, lambda_, kappa_) {} };
struct CoRotated : detail::LameSpline<detail::kCoRotated> { CoRotated(double mu_, double lambda_, double kappa_) : LameSpline(mu_, lambda_, kappa_) {} };

} // namespace xu
} // namespace admm
#endif
