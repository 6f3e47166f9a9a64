// The reference's only shipped test (samples/tests/test_lineartet.cpp) re-stated against the C++ mirror
// of the reference API in admm-elastic_amd/host (admm::Solver, admm::TetEnergyTerm, admm::Lame) -- i.e.
// through the drop-in boundary: every update()/step() below executes in the HIP kernels.
// Checks and their reference lines are listed in tests/test_known_answers.py.
#include <cmath>
#include <cstdio>
#include <iostream>
#include "Solver.hpp"
#include "TetEnergyTerm.hpp"

using namespace admm;

static bool near(double a, double b, double eps = 1e-12) { return std::fabs(a - b) < eps; }

static double volume(const VecX &x) {
    const Vec3 a = x.segment<3>(0), e0 = Vec3(x.segment<3>(3)) - a, e1 = Vec3(x.segment<3>(6)) - a, e2 = Vec3(x.segment<3>(9)) - a;
    return e0.dot(e1.cross(e2)) / 6.0;
}

struct SingleTet { // samples/tests/test_lineartet.cpp:343-396
    std::vector<Vec3> verts;
    Vec4i tet;
    SparseMat D;
    std::shared_ptr<EnergyTerm> t;
    bool init(const Lame &lame) {
        tet = Vec4i(0, 1, 2, 3);
        verts = {Vec3(0, 0, 0), Vec3(0, 1, 0), Vec3(0, 0, 1), Vec3(1, 0, 0)};
        t = std::make_shared<TetEnergyTerm>(tet, verts, lame);
        if (t->get_weight() <= 0) return false;
        std::vector<Triplet> trips; std::vector<double> w;
        t->get_reduction(trips, w);
        if (w.size() != 9 || trips.size() != 36) { std::cerr << "Bad num weights/triplets" << std::endl; return false; }
        D.resize(9, 12);
        D.setFromTriplets(trips.begin(), trips.end());
        return true;
    }
    VecX x() const { VecX r(12); for (int i = 0; i < 4; ++i) r.segment<3>(3 * i) = verts[i]; return r; }
    double vol() const { return volume(x()); }
};

static Vec3 rotate(const Vec3 &p, double deg, Vec3 axis) { // Rodrigues
    axis = axis * (1.0 / axis.norm());
    const double t = deg * M_PI / 180.0;
    return p * std::cos(t) + axis.cross(p) * std::sin(t) + axis * (axis.dot(p) * (1 - std::cos(t)));
}

static bool test_energy() {
    Lame lame; lame.mu = 0; lame.lambda = 1;
    if (!near(lame.bulk_modulus(), 1.0)) return false;
    SingleTet tet;
    if (!tet.init(lame)) return false;
    const double w = tet.t->get_weight();
    if (!near(lame.bulk_modulus() * tet.vol(), w * w)) { std::cerr << "weight function changed" << std::endl; return false; }
    if (!near(tet.t->energy(tet.D, tet.x()), 0.0)) { std::cerr << "Energy not zero at rest" << std::endl; return false; }
    tet.init(lame);
    for (auto &v : tet.verts) v = rotate(v, 45.0, Vec3(1, 1, 1));
    if (!near(tet.t->energy(tet.D, tet.x()), 0.0)) { std::cerr << "Energy not zero after rotation" << std::endl; return false; }
    tet.init(lame);
    for (auto &v : tet.verts) v = v * 2.0;
    double energy = tet.t->energy(tet.D, tet.x());
    if (!near(energy, 0.25)) { std::cerr << "Energy not correct after deformation: " << energy << std::endl; return false; }
    lame.lambda = 2.123;
    tet.init(lame);
    for (auto &v : tet.verts) v = v * 2.0;
    const double prev = energy;
    energy = tet.t->energy(tet.D, tet.x());
    if (!near(energy, prev * lame.lambda) || energy <= 0.0) { std::cerr << "Energy does not scale with lambda" << std::endl; return false; }
    // prox at rest satisfies W (Dx - z) = 0  -- this update() runs on the GPU
    tet.init(lame);
    VecX z(9), u = VecX::Zero(9);
    for (int i = 0; i < 9; ++i) z[i] = std::sin(1.0 + i);
    VecX Dx = tet.D * tet.x();
    tet.t->update(tet.D, tet.x(), z, u);
    if (!near(tet.t->get_weight() * (Dx - z).norm(), 0.0)) { std::cerr << "Prox doesn't satisfy constraint" << std::endl; return false; }
    tet.init(lame);
    const double sc[3] = {3.1, 4.2, 5.3};
    for (auto &v : tet.verts) v = Vec3(v[0] * sc[0], v[1] * sc[1], v[2] * sc[2]);
    Dx = tet.D * tet.x();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (!near(Dx[c * 3 + r], r == c ? sc[r] : 0.0)) { std::cerr << "Bad deform grad" << std::endl; return false; }
    return true;
}

static bool test_solver_iters() { // :165-230
    SingleTet tet;
    Lame lame(500000, 0.25);
    if (!tet.init(lame)) return false;
    Solver solver;
    Solver::Settings settings;
    settings.gravity = 0; settings.verbose = 0; settings.timestep_s = 1.f / 24.f; settings.linsolver = 0;
    solver.energyterms.push_back(tet.t);
    solver.m_x = tet.x();
    solver.m_masses = VecX::Ones(12);
    const VecX init_x = solver.m_x;
    double last_error = -1;
    const double true_x = 52.2321;
    for (int i = 5; i < 100; ++i) {
        settings.admm_iters = i;
        solver.m_x = init_x;
        if (!solver.initialize(settings)) return false;
        solver.m_x.segment<3>(9) = Vec3(200, 0, 0);
        solver.step();
        const double new_x = solver.m_x[9];
        if (i > 20) {
            if (!near(true_x, new_x, 1e-4)) { std::cerr << "Did not converge with iters (" << i << "): " << new_x << std::endl; return false; }
        } else if (last_error >= 1e-8) {
            if ((true_x - new_x) * (true_x - new_x) > last_error) { std::cerr << "Problem converging with increased iterations (" << i << ")" << std::endl; return false; }
        }
        last_error = (true_x - new_x) * (true_x - new_x);
    }
    return true;
}

static bool test_inversion() { // :236-323
    Lame soft; soft.mu = 100; soft.lambda = 100;
    SingleTet tet;
    if (!tet.init(soft)) return false;
    Solver solver;
    Solver::Settings settings;
    settings.gravity = 0; settings.verbose = 0; settings.timestep_s = 0.7; settings.linsolver = 0;
    solver.energyterms.push_back(tet.t);
    solver.m_x = tet.x();
    solver.m_masses = VecX::Ones(12);
    const VecX init_x = solver.m_x;
    Vec3 last_x;
    const double target_v = tet.vol();
    for (int i = 10; i < 100; ++i) {
        settings.admm_iters = i;
        solver.m_x = init_x;
        if (!solver.initialize(settings)) return false;
        if (!near(volume(solver.m_x), target_v)) return false;
        solver.m_x.segment<3>(0) = Vec3(1, 1, 1);
        if (volume(solver.m_x) > 0) { std::cerr << "Didn't invert the tet" << std::endl; return false; }
        for (int j = 0; j < 10; ++j) solver.step();
        const Vec3 curr_x = solver.m_x.segment<3>(0);
        const double new_v = volume(solver.m_x);
        if (new_v <= 0.0) { std::cerr << "Invert test: Did not fix inversion" << std::endl; return false; }
        if (!near(target_v, new_v, 1e-6)) { std::cerr << "Invert test: volume " << new_v << " at iters " << i << std::endl; return false; }
        if (i > 10 && !near((last_x - curr_x).norm(), 0.0, 1e-6)) { std::cerr << "Invert test: position differs with iters " << i << std::endl; return false; }
        last_x = curr_x;
    }
    return true;
}

int main() {
    bool ok = true;
    try {
        ok &= test_energy();
        ok &= test_solver_iters();
        ok &= test_inversion();
    } catch (std::exception &e) {
        std::cerr << "exception: " << e.what() << std::endl;
        ok = false;
    }
    if (!ok) { std::cerr << "\n**FAILURE**\n" << std::endl; return 1; }
    std::cout << "SUCCESS" << std::endl;
    return 0;
}
