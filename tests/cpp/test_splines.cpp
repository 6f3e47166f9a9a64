// SplineTet with the reference's splines (src/XuSpline.hpp) on the C++ class mirror:
//   * xu::StVK(mu, lambda, 0) is the StVK model: SplineTet(..., StVK spline) and StVKTet give the same trajectory;
//   * xu::NeoHookean likewise against NeoHookeanTet; xu::CoRotated runs and pulls a stretched cube back;
//   * the spline's own constants are used (a stiffer spline than the tet's Lame changes the result);
//   * a compression term (kappa != 0) changes the result and stays finite; a user-defined spline has no kernel:
//     Solver::initialize refuses instead of running something else.
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
        struct MySpline : xu::Spline {   // a user-defined spline: no kernel
            double f(double x) const { return x * x; } double g(double) const { return 0; } double h(double) const { return 0; }
            double df(double x) const { return 2 * x; } double dg(double) const { return 0; } double dh(double) const { return 0; }
        };
        run([&](const Vec4i &t, const std::vector<Vec3> &v) { return std::make_shared<SplineTet>(t, v, lame, std::make_shared<MySpline>()); }, 1, t0);
        check(t0, "user-defined spline: initialize refuses (no kernel, no CPU fallback)");
    }
    std::printf(fail ? "FAILURE\n" : "SUCCESS\n");
    return fail ? 1 : 0;
}
