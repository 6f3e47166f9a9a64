// CPU-only checks of the mesh layer (Meshes.hpp / AddMeshes.hpp): generators, lumped masses, surface
// extraction, TetGen round trip, GrabbySphere, and that binding::add_tetmesh / add_trimesh fill the solver's
// node vectors and energy-term list the way the reference's helpers do (AddMeshes.hpp:97-235).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "AddMeshes.hpp"
#include "ExplicitForce.hpp"
#include "XuSpline.hpp"

using namespace admm;

#define CHECK(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp/admm_test_mesh";
    auto m = factory::make_tet_blocks(4, 3, 2);
    CHECK(m->vertices.size() == 5 * 4 * 3 && m->tets.size() == 6 * 4 * 3 * 2);
    double vol = 0.0;
    for (int t = 0; t < (int)m->tets.size(); ++t) { CHECK(m->signed_volume(t) > 0.0); vol += m->signed_volume(t); }
    CHECK(std::fabs(vol - 24.0) < 1e-12);
    std::vector<double> masses;
    m->weighted_masses(masses, 1522.0);
    double ms = 0.0; for (double x : masses) { CHECK(x > 0.0); ms += x; }
    CHECK(std::fabs(ms - 1522.0 * 24.0) < 1e-8);
    std::vector<Vec3i> faces; m->surface_faces(faces);
    CHECK(faces.size() == 2 * 2 * (4 * 3 + 4 * 2 + 3 * 2));            // two triangles per boundary cell face
    std::vector<int> surf; m->surface_inds(surf);
    CHECK(surf.size() == 5 * 4 * 3 - 3 * 2 * 1);                       // all but the interior vertices
    // outward orientation: the closed surface has the mesh volume
    double sv = 0.0;
    for (const Vec3i &f : faces) sv += m->vertices[f[0]].dot(m->vertices[f[1]].cross(m->vertices[f[2]])) / 6.0;
    CHECK(std::fabs(sv - 24.0) < 1e-10);
    Vec3 lo, hi; m->bounds(lo, hi);
    CHECK(lo[0] == 0 && hi[0] == 4 && hi[1] == 3 && hi[2] == 2);
    {   // renumber_for_locality: a scrambled numbering is replaced (same geometry per tet), a good one is left alone
        auto g = factory::make_tet_blocks(6, 6, 6);
        const std::vector<int> keep = g->renumber_for_locality();
        for (size_t i = 0; i < keep.size(); ++i) CHECK(keep[i] == (int)i);
        auto sh = factory::make_tet_blocks(6, 6, 6);
        const int nv = (int)sh->vertices.size();
        std::vector<int> p(nv);
        for (int i = 0; i < nv; ++i) p[i] = (int)(((long long)i * 7919) % nv);      // 7919 is coprime with 343: a permutation
        std::vector<Vec3> vv(nv);
        for (int i = 0; i < nv; ++i) vv[p[i]] = sh->vertices[i];
        sh->vertices = vv;
        for (Vec4i &t : sh->tets) for (int c = 0; c < 4; ++c) t[c] = p[t[c]];
        auto span = [](const TetMesh &mm) { double s = 0; for (const Vec4i &t : mm.tets) for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) s += std::abs(t[a] - t[b]); return s / (6.0 * mm.tets.size()); };
        const double before = span(*sh);
        const std::vector<int> nid = sh->renumber_for_locality();
        CHECK(span(*sh) < 0.5 * before);
        std::vector<char> seen(nv, 0);
        for (int i = 0; i < nv; ++i) { CHECK(nid[i] >= 0 && nid[i] < nv && !seen[nid[i]]); seen[nid[i]] = 1; }
        for (size_t t = 0; t < sh->tets.size(); ++t) {
            CHECK(std::fabs(sh->signed_volume((int)t) - g->signed_volume((int)t)) < 1e-12);
            for (int c = 0; c < 4; ++c) CHECK((sh->vertices[sh->tets[t][c]] - g->vertices[g->tets[t][c]]).norm() == 0.0);
        }
    }
    {   // xu:: splines (XuSpline.hpp): the header's shared evaluation against the formulas of the paper written out per
        // spline (f, g, h and derivatives, kappa included), and the kinds / constants flatten() reports
        const double mu = 3.0, la = 7.0, ka = 2.0;
        xu::NeoHookean sn(mu, la, ka); xu::StVK ss(mu, la, ka); xu::CoRotated sc(mu, la, ka);
        for (double x = 0.3; x < 2.5; x += 0.37) {
            const double x2 = x * x, l = std::log(x), ct = (ka / 12.0) * std::pow((1.0 - x) / 6.0, 3.0), dct = (-ka / 24.0) * std::pow((1.0 - x) / 6.0, 2.0);
            const double e[18] = {0.5 * mu * (x2 - 1), 0, -mu * l + 0.5 * la * l * l + ct, mu * x, 0, -mu / x + la * l / x + dct,
                                  0.125 * la * (x2 * x2 - 6 * x2 + 5) + 0.25 * mu * (x2 - 1) * (x2 - 1), 0.25 * la * (x2 - 1), ct,
                                  0.125 * la * (4 * x2 * x - 12 * x) + mu * x * (x2 - 1), 0.5 * la * x, dct,
                                  0.5 * la * (x2 - 6 * x + 5) + mu * (x - 1) * (x - 1), la * (x - 1), ct, 0.5 * la * (2 * x - 6) + 2 * mu * (x - 1), la, dct};
            const double g[18] = {sn.f(x), sn.g(x), sn.h(x), sn.df(x), sn.dg(x), sn.dh(x), ss.f(x), ss.g(x), ss.h(x), ss.df(x), ss.dg(x), ss.dh(x),
                                  sc.f(x), sc.g(x), sc.h(x), sc.df(x), sc.dg(x), sc.dh(x)};
            for (int i = 0; i < 18; ++i) CHECK(std::fabs(e[i] - g[i]) <= 1e-12 * (1 + std::fabs(e[i])));
        }
        int kd; double m_, l_, k_;
        CHECK(sn.flatten(kd, m_, l_, k_) && kd == 3 && k_ == ka);                   // the compression term travels with the spline
        xu::StVK s0(mu, la, 0.0); CHECK(s0.flatten(kd, m_, l_, k_) && kd == 4 && m_ == mu && l_ == la && k_ == 0.0);
        xu::CoRotated c0(mu, la, 0.0); CHECK(c0.flatten(kd, m_, l_, k_) && kd == 5);
        xu::NeoHookean n0(mu, la, 0.0); CHECK(n0.flatten(kd, m_, l_, k_) && kd == 3);
        CHECK(n0.mu == mu && n0.lambda == la && n0.kappa == 0.0);                   // public constants as in the reference
    }
    // TetGen round trip (0-based) and a 1-based file with a flipped tet
    meshio::save_tetgen(tmp, *m);
    auto r = meshio::load_tetgen(tmp);
    CHECK(r->vertices.size() == m->vertices.size() && r->tets.size() == m->tets.size());
    for (size_t i = 0; i < m->tets.size(); ++i) for (int c = 0; c < 4; ++c) CHECK(r->tets[i][c] == m->tets[i][c]);
    for (size_t i = 0; i < m->vertices.size(); ++i) CHECK((r->vertices[i] - m->vertices[i]).norm() == 0.0);
    {
        FILE *f = fopen((tmp + "_b.node").c_str(), "w");
        fprintf(f, "# one tet, 1-based\n4 3 0 0\n1 0 0 0\n2 1 0 0\n3 0 1 0\n4 0 0 1\n"); fclose(f);
        f = fopen((tmp + "_b.ele").c_str(), "w");
        fprintf(f, "1 4 0\n1 1 3 2 4\n"); fclose(f);   // negatively oriented as written
        auto b = meshio::load_tetgen(tmp + "_b");
        CHECK(b->tets.size() == 1 && b->signed_volume(0) > 0.0);
    }
    auto p = factory::make_plane(5, 2.0, 0.5);
    CHECK(p->vertices.size() == 36 && p->faces.size() == 50);
    p->weighted_masses(masses, 1.0);
    ms = 0.0; for (double x : masses) ms += x;
    CHECK(std::fabs(ms - 4.0) < 1e-12);
    // binding: nodes, masses (x3), energy terms, flags
    Solver solver;
    m->flags = binding::NOSELFCOLLISION | binding::NEOHOOKEAN;
    binding::add_tetmesh(&solver, m, Lame::soft_rubber(), false);
    CHECK(solver.m_x.size() == 3 * 60 && solver.m_masses.size() == 3 * 60 && solver.energyterms.size() == 144);
    CHECK(solver.m_masses[0] == solver.m_masses[1] && solver.m_masses[1] == solver.m_masses[2] && solver.m_masses[0] > 0.0);
    CHECK(dynamic_cast<NeoHookeanTet *>(solver.energyterms[0].get()) != nullptr);
    GrabbySphere gs(Vec3(0, 0, 0), 1.01);
    std::vector<int> inds; gs.get_indices(solver.m_x, inds);
    CHECK(inds.size() == 4);   // (0,0,0), (1,0,0), (0,1,0), (0,0,1) of the tet block
    p->flags = binding::NOSELFCOLLISION;
    binding::add_trimesh(&solver, p, Lame(100, 0.1), false);
    CHECK(solver.m_x.size() == 3 * (60 + 36) && solver.energyterms.size() == 144 + 50);
    CHECK(dynamic_cast<TriEnergyTerm *>(solver.energyterms[144].get()) != nullptr);
    CHECK(std::fabs(solver.m_x[3 * 60 + 1] - 0.5) < 1e-15);
    p->flags = binding::STVK;
    bool threw = false;
    try { binding::add_trimesh(&solver, p, Lame(100, 0.1), false); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    {   // WindForce (ExplicitForce.cpp:47-104) on one triangle, worked by hand: n = (0,0,1), area 1/2, v_r = (0,0,1.5)
        //   force = -1000 * 0.5 * 1.5 * 1.5 * n = (0,0,-1125);  dv = 0.33 * dt * force = (0,0,-3.7125) at dt = 0.01
        VecX x(9), v(9), m(9, 1.0);
        const double px[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0};
        for (int i = 0; i < 9; ++i) { x[i] = px[i]; v[i] = (i % 3 == 2) ? 2.0 : 0.0; }
        std::vector<int> tri = {0, 1, 2};
        WindForce wind(tri);
        wind.direction = Vec3(0.0, 0.0, 0.5);
        wind.project(0.01, x, v, m);
        for (int i = 0; i < 3; ++i) { CHECK(v[3 * i] == 0.0 && v[3 * i + 1] == 0.0); CHECK(std::fabs(v[3 * i + 2] - (2.0 - 3.7125)) < 1e-12); }
        // plugged into the solver like the reference does (Solver.hpp:72, Solver.cpp:54)
        solver.ext_forces.push_back(std::make_shared<WindForce>(tri));
        CHECK(solver.ext_forces.size() == 1);
    }
    printf("SUCCESS\n");
    return 0;
}
