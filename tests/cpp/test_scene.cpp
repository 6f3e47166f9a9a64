// A small scene driven ONLY through the C++ mirror of the reference API (create_tets_from_mesh, set_pins,
// add_obstacle, Settings, initialize, step, runtime_data), in the style of the reference's samples
// (samples/utils/AddMeshes.hpp:97-177, samples/tvcg2017/boxes.cpp).  Prints the final positions; the
// pytest wrapper compares them with the Python binding driving the same C ABI.
//   usage: test_scene <linsolver 0|1|2> <frames>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "Solver.hpp"
#include "TetEnergyTerm.hpp"

using namespace admm;

// A subclass that reads the protected global matrices of src/Solver.hpp:115-121 the way user code of the reference may:
// dt^2 D^T W^T W D must be the assembled system matrix (solver_termA = Ahat, one scalar block for the three axes).
struct ProbeSolver : public Solver {
    bool matrices_consistent() const {
        const int dof = (int)m_x.size(), n_row = (int)m_W_diag.size();
        if (m_D.rows() != n_row || m_D.cols() != dof || m_Dt.rows() != dof || solver_Dt_Wt_W.rows() != dof || solver_Dt_Wt_W.cols() != n_row) return false;
        VecX v(dof);
        for (int i = 0; i < dof; ++i) v[i] = std::sin(0.37 * i) + 0.1 * (i % 7);
        const VecX lhs = solver_Dt_Wt_W * (m_D * v);
        double worst = 0.0, scale = 0.0;
        for (int i = 0; i < dof / 3; ++i)
            for (int a = 0; a < 3; ++a) {
                double r = 0.0;
                for (int k = solver_termA.rowptr()[i]; k < solver_termA.rowptr()[i + 1]; ++k) r += solver_termA.values()[k] * v[3 * solver_termA.colind()[k] + a];
                worst = std::max(worst, std::fabs(r - lhs[3 * i + a])); scale = std::max(scale, std::fabs(r));
            }
        return n_row > 0 && worst <= 1e-10 * scale;
    }
};

int main(int argc, char **argv) {
    const int ls = argc > 1 ? atoi(argv[1]) : 0, frames = argc > 2 ? atoi(argv[2]) : 3;
    const int n = 3; // cells per edge, Kuhn triangulation built inline (6 tets per cell)
    std::vector<double> verts; std::vector<int> tets;
    auto vid = [&](int i, int j, int k) { return (i * (n + 1) + j) * (n + 1) + k; };
    for (int i = 0; i <= n; ++i) for (int j = 0; j <= n; ++j) for (int k = 0; k <= n; ++k) {
        verts.push_back(0.5 * i / n); verts.push_back(0.5 * j / n + 0.02); verts.push_back(0.5 * k / n);
    }
    const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k)
        for (int p = 0; p < 6; ++p) {
            int c[3] = {i, j, k}, id[4];
            id[0] = vid(c[0], c[1], c[2]);
            for (int s = 0; s < 3; ++s) { c[perms[p][s]] += 1; id[s + 1] = vid(c[0], c[1], c[2]); }
            // orientation = parity of the permutation
            const bool odd = (p == 1 || p == 2 || p == 5);
            if (odd) std::swap(id[2], id[3]);
            for (int s = 0; s < 4; ++s) tets.push_back(id[s]);
        }
    const int nv = (int)verts.size() / 3, nt = (int)tets.size() / 4;
    // lumped masses rho vol / 4 (AddMeshes.hpp:113-122), rho = 1522
    std::vector<double> m(3 * nv, 0.0);
    const double cell = 0.5 / n, vol = cell * cell * cell / 6.0;
    for (int t = 0; t < nt; ++t) for (int s = 0; s < 4; ++s) for (int a = 0; a < 3; ++a) m[3 * tets[4 * t + s] + a] += 1522.0 * vol / 4.0;

    ProbeSolver solver;
    solver.add_nodes(verts.data(), m.data(), nv);
    create_tets_from_mesh<double, NeoHookeanTet>(solver.energyterms, verts.data(), tets.data(), nt, Lame::soft_rubber(), 0);
    Solver::Settings st;
    st.verbose = 0; st.admm_iters = 8; st.linsolver = ls;
    std::vector<int> pins; std::vector<Vec3> pts;
    if (ls == 0) { // cantilever: pin the x = 0 face, then drag it upwards every frame
        for (int j = 0; j <= n; ++j) for (int k = 0; k <= n; ++k) pins.push_back(vid(0, j, k));
        solver.set_pins(pins);
    } else {
        solver.add_obstacle(std::make_shared<Floor>(0.0));
    }
    if (!solver.initialize(st)) return 2;
    if (!solver.matrices_consistent()) { fprintf(stderr, "m_D / m_W_diag / solver_Dt_Wt_W do not reproduce solver_termA\n"); return 3; }
    for (int f = 0; f < frames; ++f) {
        if (ls == 0) {
            pts.clear();
            for (int v : pins) pts.push_back(Vec3(verts[3 * v], verts[3 * v + 1] + 0.01 * (f + 1), verts[3 * v + 2]));
            solver.set_pins(pins, pts);
        }
        solver.step();
    }
    printf("inner_iters %d\n", solver.runtime_data().inner_iters);
    for (int i = 0; i < 3 * nv; ++i) printf("%.17g\n", solver.m_x[i]);
    return 0;
}
