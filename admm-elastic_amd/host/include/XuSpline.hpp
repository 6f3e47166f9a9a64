// XuSpline.hpp -- the spline family of SplineTet (reference: src/XuSpline.hpp; Xu, Sin, Zhu, Barbic 2015,
// "Nonlinear Material Design Using Principal Stretches").  Psi = sum f(s_i) + sum g(s_i s_j) + h(s_0 s_1 s_2).
// The three classes the reference ships keep their names, constructors and f/g/h formulas (host-side energy
// evaluation); on the GPU each of them, with kappa = 0, IS one of the closed-form models (NeoHookean -> NH, StVK -> StVK,
// CoRotated -> co-rotated linear), which is how SplineTet::flatten hands it to the kernels.  A spline with
// kappa != 0, or a user-defined subclass, has no kernel: Solver::initialize rejects it (no CPU fallback).
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
    // Eq. 16: compression term (src/XuSpline.hpp:44-45)
    static double compress_term(double kappa, double x) { return (kappa / 12.0) * std::pow((1.0 - x) / 6.0, 3.0); }
    static double d_compress_term(double kappa, double x) { return (-kappa / 24.0) * (std::pow((1.0 - x) / (6.0), 2.0)); }
    // GPU description: the ADMM_TET_SPLINE_* kind and the spline's constants; false = no kernel
    virtual bool flatten(int &kind, double &mu, double &lambda) const { (void)kind; (void)mu; (void)lambda; return false; }
};

class NeoHookean : public Spline {   // src/XuSpline.hpp:48-62
public:
    NeoHookean(double mu_, double lambda_, double kappa_) : mu(mu_), lambda(lambda_), kappa(kappa_) {}
    const double mu, lambda, kappa;
    double f(double x) const { return 0.5 * mu * (x * x - 1.0); }
    double g(double) const { return 0.0; }
    double h(double x) const { const double l = std::log(x); return -mu * l + 0.5 * lambda * l * l + compress_term(kappa, x); }
    double df(double x) const { return mu * x; }
    double dg(double) const { return 0.0; }
    double dh(double x) const { return -mu / x + lambda * std::log(x) / x + d_compress_term(kappa, x); }
    bool flatten(int &kind, double &m, double &l) const { kind = 3; m = mu; l = lambda; return kappa == 0.0; }
};

class StVK : public Spline {         // src/XuSpline.hpp:64-82
public:
    StVK(double mu_, double lambda_, double kappa_) : mu(mu_), lambda(lambda_), kappa(kappa_) {}
    const double mu, lambda, kappa;
    double f(double x) const { const double x2 = x * x; return 0.125 * lambda * (x2 * x2 - 6.0 * x2 + 5.0) + 0.25 * mu * (x2 - 1.0) * (x2 - 1.0); }
    double g(double x) const { return 0.25 * lambda * (x * x - 1.0); }
    double h(double x) const { return compress_term(kappa, x); }
    double df(double x) const { const double x2 = x * x; return 0.125 * lambda * (4.0 * x2 * x - 12.0 * x) + mu * x * (x2 - 1.0); }
    double dg(double x) const { return 0.5 * lambda * x; }
    double dh(double x) const { return d_compress_term(kappa, x); }
    bool flatten(int &kind, double &m, double &l) const { kind = 4; m = mu; l = lambda; return kappa == 0.0; }
};

class CoRotated : public Spline {    // src/XuSpline.hpp:84-96
public:
    CoRotated(double mu_, double lambda_, double kappa_) : mu(mu_), lambda(lambda_), kappa(kappa_) {}
    const double mu, lambda, kappa;
    double f(double x) const { return 0.5 * lambda * (x * x - 6.0 * x + 5.0) + mu * (x - 1.0) * (x - 1.0); }
    double g(double x) const { return lambda * (x - 1.0); }
    double h(double x) const { return compress_term(kappa, x); }
    double df(double x) const { return 0.5 * lambda * (2.0 * x - 6.0) + 2.0 * mu * (x - 1.0); }
    double dg(double) const { return lambda; }
    double dh(double x) const { return d_compress_term(kappa, x); }
    bool flatten(int &kind, double &m, double &l) const { kind = 5; m = mu; l = lambda; return kappa == 0.0; }
};

} // namespace xu
} // namespace admm
#endif
