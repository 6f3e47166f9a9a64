// SplineTet with the reference's splines (src/XuSpline.hpp) on the C++ class mirror:
//   * xu::StVK(mu, lambda, 0) is the StVK model: SplineTet(..., StVK spline) and StVKTet give the same trajectory;
//   * xu::NeoHookean likewise against NeoHookeanTet; xu::CoRotated runs and pulls a stretched cube back;
//   * the spline's own constants are used (a stiffer spline than the tet's Lame changes the result);
//   * a compression term (kappa != 0) changes the result and stays finite; a USER-DEFINED spline is tabulated by
//     Solver::initialize and runs the table kernel -- checked against the closed-form kernel of the same functions.
#include <cmath>
#include <cstdio>
#include <iostream>
#include "AddMeshes.hpp"

using namespace admm;

template <class MAKE>
static VecX run(MAKE make_term, int frames, bool &threw) {
    threw = false;
    auto mesh = factory::make_tet_blocks(2, 2, 2);
    mesh->scale(0.5, 0.5, 0.5);
    Solver solver;
    std::vector<double> m; mesh->weighted_masses(m, 1522.0);
    std::vector<double> x, m3;
    for (size_t i = 0; i < mesh->vertices.size(); ++i) for (int a = 0; a < 3; ++a) { x.push_back(mesh->vertices[i][a] * (a == 0 ? 1.2 : 1.0)); m3.push_back(m[i]); }
    solver.add_nodes(x.data(), m3.data(), (int)mesh->vertices.size());
    for (const Vec4i &t : mesh->tets) {
        std::vector<Vec3> tv;
        for (int c = 0; c < 4; ++c) tv.push_back(mesh->vertices[t[c]]);
        solver.energyterms.emplace_back(make_term(t, tv));
    }
    Solver::Settings s; s.verbose = 0; s.gravity = 0; s.admm_iters = 15;
    try {
        if (!solver.initialize(s)) { threw = true; return VecX(); }
        for (int f = 0; f < frames; ++f) solver.step();
    } catch (const std::exception &e) { threw = true; std::cout << "  (threw: " << e.what() << ")" << std::endl; return VecX(); }
    return solver.m_x;
}

static double maxdiff(const VecX &a, const VecX &b) { double d = 0; for (int i = 0; i < a.size(); ++i) d = std::max(d, std::fabs(a[i] - b[i])); return d; }

int main() {
    const Lame lame = Lame::soft_rubber();
    bool t0, t1;
    int fail = 0;
    auto check = [&](bool ok, const char *what) { std::printf("%s: %s\n", ok ? "ok  " : "FAIL", what); if (!ok) ++fail; };
    {
        VecX a = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<StVKTet>(t, v, lame); }, 5, t0);
        VecX b = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<xu::StVK>(lame.mu, lame.lambda, 0.0)); }, 5, t1);
        check(!t0 && !t1 && maxdiff(a, b) < 1e-12, "SplineTet(xu::StVK, kappa 0) == StVKTet");
    }
    {
        VecX a = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<NeoHookeanTet>(t, v, lame); }, 5, t0);
        VecX b = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame); }, 5, t1);
        check(!t0 && !t1 && maxdiff(a, b) < 1e-12, "SplineTet (default xu::NeoHookean) == NeoHookeanTet");
        VecX c = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<xu::NeoHookean>(3.0 * lame.mu, lame.lambda, 0.0)); }, 5, t1);
        check(!t1 && maxdiff(a, c) > 1e-6, "the spline's own constants are used");
    }
    {
        VecX a = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<xu::CoRotated>(lame.mu, lame.lambda, 0.0)); }, 40, t0);
        bool finite = !t0;
        double xmax = 0;
        for (int i = 0; finite && i < a.size(); ++i) { finite = std::isfinite(a[i]); if (i % 3 == 0) xmax = std::max(xmax, a[i]); }
        double xmin = 1e300; for (int i = 0; finite && i < a.size(); i += 3) xmin = std::min(xmin, a[i]);
        std::printf("  co-rotated: finite %d, extent along x after 40 frames %.4f (rest 1.0, start 1.2)\n", (int)finite, xmax - xmin);
        check(finite && std::fabs((xmax - xmin) - 1.0) < 0.05, "SplineTet(xu::CoRotated): the cube stretched by 1.2 along x is back near its rest length");
    }
    {
        VecX a = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<xu::StVK>(lame.mu, lame.lambda, 0.0)); }, 5, t0);
        VecX b = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<xu::StVK>(lame.mu, lame.lambda, 50.0 * lame.mu)); }, 5, t1);
        bool finite = !t0 && !t1;
        for (int i = 0; finite && i < b.size(); ++i) finite = std::isfinite(b[i]);
        check(finite && maxdiff(a, b) > 1e-9, "SplineTet(xu::StVK, kappa != 0): runs on the GPU, finite, and the compression term changes the result");
    }
    {
        // A USER-DEFINED xu::Spline (src/XuSpline.hpp:34-46: any object with the six functions; src/TetEnergyTerm.hpp:197-204): sampled by
        // Solver::initialize into device tables.  This one re-states xu::StVK with a compression term behind an unknown dynamic type, so the
        // table kernel can be held against the closed-form kernel of the named spline.
        struct MyStVK : xu::Spline {
            double mu, la, ka;
            MyStVK(double m, double l, double k) : mu(m), la(l), ka(k) {}
            double f(double s) const { const double s2 = s * s; return la * (s2 * s2 - 6.0 * s2 + 5.0) / 8.0 + mu * (s2 - 1.0) * (s2 - 1.0) / 4.0; }
            double df(double s) const { return la * (s * s * s - 3.0 * s) / 2.0 + mu * s * (s * s - 1.0); }
            double g(double p) const { return la * (p * p - 1.0) / 4.0; }
            double dg(double p) const { return la * p / 2.0; }
            double h(double J) const { return compress_term(ka, J); }
            double dh(double J) const { return d_compress_term(ka, J); }
        };
        const double kap = 50.0 * lame.mu;
        VecX a = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<xu::StVK>(lame.mu, lame.lambda, kap)); }, 5, t0);
        auto mine = std::make_shared<MyStVK>(lame.mu, lame.lambda, kap);
        VecX b = run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, mine); }, 5, t1);
        std::printf("  user-defined spline (tables) vs the named one (closed form): max position difference %.3e\n", (!t0 && !t1) ? maxdiff(a, b) : -1.0);
        check(!t0 && !t1 && maxdiff(a, b) < 1e-7, "user-defined spline: tabulated at initialize, runs on the GPU, matches the closed-form kernel of the same functions");
        struct Bad : xu::Spline {   // a spline that returns NaN inside the table range: refused with the reason
            double f(double) const { return std::nan(""); } double g(double) const { return 0; } double h(double) const { return 0; }
            double df(double) const { return 0; } double dg(double) const { return 0; } double dh(double) const { return 0; }
        };
        run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<Bad>()); }, 1, t0);
        check(t0, "a spline that is not finite on its table range: initialize refuses");
    }
    std::printf(fail ? "FAILURE\n" : "SUCCESS\n");
    return fail ? 1 : 0;
}
