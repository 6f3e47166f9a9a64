// The LinearSolver objects' public tuning members changed AFTER Solver::initialize: the reference reads them on every solve
// (src/NodalMultiColorGS.hpp:40-46 used at :100; src/UzawaCG.hpp:44-45 used at :92), so user code may cast the solver and change
// them between steps.  Multi-colour GS at tolerance 1e-10 never converges in 30 sweeps, so the inner iteration
// count of a step is exactly admm_iters x max_iters: halving max_iters must halve it.
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <memory>
#include <vector>
#include "Solver.hpp"
#include "../../include/admm_hip.h"
#include "TetEnergyTerm.hpp"

using namespace admm;

int main() {
    const int n = 3;
    std::vector<double> verts; std::vector<int> tets;
    auto vid = [&](int i, int j, int k) { return (i * (n + 1) + j) * (n + 1) + k; };
    for (int i = 0; i <= n; ++i) for (int j = 0; j <= n; ++j) for (int k = 0; k <= n; ++k) { verts.push_back(0.5 * i / n); verts.push_back(0.5 * j / n + 0.02); verts.push_back(0.5 * k / n); }
    const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k)
        for (int p = 0; p < 6; ++p) {
            int c[3] = {i, j, k}, id[4];
            id[0] = vid(c[0], c[1], c[2]);
            for (int s = 0; s < 3; ++s) { c[perms[p][s]] += 1; id[s + 1] = vid(c[0], c[1], c[2]); }
            if (p == 1 || p == 2 || p == 5) std::swap(id[2], id[3]);
            for (int s = 0; s < 4; ++s) tets.push_back(id[s]);
        }
    const int nv = (int)verts.size() / 3, nt = (int)tets.size() / 4;
    std::vector<double> m(3 * nv, 0.0);
    const double cell = 0.5 / n, vol = cell * cell * cell / 6.0;
    for (int t = 0; t < nt; ++t) for (int s = 0; s < 4; ++s) for (int a = 0; a < 3; ++a) m[3 * tets[4 * t + s] + a] += 1522.0 * vol / 4.0;
    int failures = 0;
    {   // ---- NodalMultiColorGS: max_iters 30 -> 15 after initialize ----
        Solver solver;
        solver.add_nodes(verts.data(), m.data(), nv);
        create_tets_from_mesh<double, NeoHookeanTet>(solver.energyterms, verts.data(), tets.data(), nt, Lame::soft_rubber(), 0);
        std::vector<int> pins;      // a cantilever (x = 0 face pinned): the sweeps never meet 1e-10, every solve runs max_iters of them
        for (int j = 0; j <= n; ++j) for (int k = 0; k <= n; ++k) pins.push_back(vid(0, j, k));
        solver.set_pins(pins);
        Solver::Settings st; st.verbose = 0; st.admm_iters = 8; st.linsolver = 1;
        if (!solver.initialize(st)) return 2;
        solver.step();
        const int full = solver.runtime_data().inner_iters;
        auto gs = std::dynamic_pointer_cast<NodalMultiColorGS>(solver.linear_solver());
        if (!gs) { fprintf(stderr, "linear_solver() is not a NodalMultiColorGS\n"); return 3; }
        gs->max_iters = 15;
        solver.step();
        const int half = solver.runtime_data().inner_iters;
        printf("GS: inner_iters %d with max_iters 30, %d with max_iters 15\n", full, half);
        if (full != 8 * 30 || half != 8 * 15) { fprintf(stderr, "FAILURE: GS max_iters changed after initialize was not honoured\n"); ++failures; }
        gs->max_iters = 30; gs->m_omega = 1.0;      // plain Gauss-Seidel: another trajectory than with omega 1.9
        VecX before = solver.m_x;
        solver.step();
        if (solver.runtime_data().inner_iters != 8 * 30) { fprintf(stderr, "FAILURE: GS max_iters restored\n"); ++failures; }
    }
    {   // ---- UzawaCG: max_iters 20 -> 2 caps the Schur iterations of a contact step ----
        Solver solver;
        solver.add_nodes(verts.data(), m.data(), nv);
        create_tets_from_mesh<double, NeoHookeanTet>(solver.energyterms, verts.data(), tets.data(), nt, Lame::soft_rubber(), 0);
        solver.add_obstacle(std::make_shared<Floor>(0.0));
        Solver::Settings st; st.verbose = 0; st.admm_iters = 8; st.linsolver = 2;
        if (!solver.initialize(st)) return 2;
        for (int f = 0; f < 3; ++f) solver.step();      // the cube starts 0.02 above the floor and lands
        const int full = solver.runtime_data().inner_iters;
        auto uz = std::dynamic_pointer_cast<UzawaCG>(solver.linear_solver());
        if (!uz) { fprintf(stderr, "linear_solver() is not a UzawaCG\n"); return 3; }
        uz->max_iters = 2;
        solver.step();
        const int capped = solver.runtime_data().inner_iters;
        printf("UzawaCG: inner_iters %d with max_iters 20, %d with max_iters 2\n", full, capped);
        if (!(full > 8 * 2 && capped <= 8 * 2 && capped > 0)) { fprintf(stderr, "FAILURE: UzawaCG max_iters changed after initialize was not honoured\n"); ++failures; }
    }
    {   // ---- Settings::soft_modes (GPU build): the end projection changes a converged trajectory by no more than the tolerance ----
        VecX xs[2];
        for (int pass = 0; pass < 2; ++pass) {
            Solver solver;
            solver.add_nodes(verts.data(), m.data(), nv);
            create_tets_from_mesh<double, NeoHookeanTet>(solver.energyterms, verts.data(), tets.data(), nt, Lame::soft_rubber(), 0);
            std::vector<int> pins;
            for (int j = 0; j <= n; ++j) for (int k = 0; k <= n; ++k) pins.push_back(vid(0, j, k));
            solver.set_pins(pins);
            Solver::Settings st; st.verbose = 0; st.admm_iters = 8; st.linsolver = 0; st.soft_modes = pass ? 6 : 0;
            if (!solver.initialize(st)) return 2;
            int32_t k = -1;
            if (admm_hip_get_soft_modes((admm_hip_ctx *)solver.context(), &k, nullptr) != ADMM_HIP_OK || k != st.soft_modes) { fprintf(stderr, "FAILURE: %d soft modes installed, %d asked for\n", (int)k, st.soft_modes); ++failures; }
            for (int f = 0; f < 3; ++f) solver.step();
            xs[pass] = solver.m_x;
        }
        double dmax = 0.0, moved = 0.0;
        for (int i = 0; i < 3 * nv; ++i) { dmax = std::max(dmax, std::fabs(xs[0][i] - xs[1][i])); moved = std::max(moved, std::fabs(xs[0][i] - verts[i])); }
        printf("soft modes: trajectories differ by %.2e (moved %.2e)\n", dmax, moved);
        if (!(dmax < 1e-8 && moved > 1e-3)) { fprintf(stderr, "FAILURE: Settings::soft_modes changed the converged trajectory\n"); ++failures; }
    }
    if (failures) return 1;
    printf("SUCCESS\n");
    return 0;
}
