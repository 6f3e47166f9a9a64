// Solver.hpp -- admm::Solver of the MI355X build (reference: src/Solver.hpp:32-124).
// Same public surface; initialize() flattens the scene into an admm_hip_desc and step() runs the whole
// ADMM loop on the GPU through include/admm_hip.h.  m_x / m_v stay valid host-side after every step.
#ifndef ADMM_SOLVER_HPP
#define ADMM_SOLVER_HPP 1

#include <string>
#include <unordered_map>
#include "ConstraintSet.hpp"
#include "EnergyTerm.hpp"
#include "SpringEnergyTerm.hpp"
#include "ExplicitForce.hpp"
#include "LinearSolver.hpp"
#include "PassiveObject.hpp"

namespace admm {

class Solver {
public:
    // src/Solver.hpp:39-50
    struct Settings {
        bool parse_args(int argc, char **argv); // returns true if help() was printed
        void help();
        double timestep_s;   // -dt
        int verbose;         // -v
        int admm_iters;      // -it
        double gravity;      // -g
        int linsolver;       // -ls  0=LDLT(->GPU PCG), 1=NCMCGS, 2=UzawaCG
        double constraint_w; // -ck
        Settings() : timestep_s(1.0 / 24.0), verbose(1), admm_iters(10), gravity(-9.8), linsolver(0), constraint_w(-1) {}
    };
    // src/Solver.hpp:54-61
    struct RuntimeData {
        double global_ms, local_ms, collision_ms;
        int inner_iters;
        RuntimeData() : global_ms(0), local_ms(0), collision_ms(0), inner_iters(0) {}
        void print(const Settings &settings);
    };

    Solver();
    virtual ~Solver();

    VecX m_x, m_v, m_masses;          // per-node x3 (src/Solver.hpp:66-68)
    std::vector<int> surface_inds;
    std::vector<std::shared_ptr<ExplicitForce> > ext_forces;
    std::vector<std::shared_ptr<EnergyTerm> > energyterms;

    template <typename T> int add_nodes(T *x, T *m, int n_verts); // src/Solver.hpp:127-141
    virtual void set_pins(const std::vector<int> &inds, const std::vector<Vec3> &points = std::vector<Vec3>());
    virtual void add_obstacle(std::shared_ptr<PassiveCollision> obj);
    virtual void add_dynamic_collider(std::shared_ptr<DynamicCollision> obj);
    virtual bool initialize(const Settings &settings_ = Settings());
    virtual void step();
    virtual const RuntimeData &runtime_data() { return m_runtime; }
    virtual void save_matrix(const std::string &filename);
    const Settings &settings() { return m_settings; }
    // GPU-side extras
    std::shared_ptr<LinearSolver> linear_solver() { return m_linsolver; }
    int device; // HIP device ordinal used by initialize() (default 0)

protected:
    Settings m_settings;
    RuntimeData m_runtime;
    bool initialized;
    std::shared_ptr<LinearSolver> m_linsolver;
    std::shared_ptr<ConstraintSet> m_constraints;
    std::unordered_map<int, std::shared_ptr<SpringPin> > m_pin_energies;
    SparseMat solver_termA; // Ahat: A = diag(m) + Ahat (x) I3
    void *m_ctx;            // admm_hip_ctx
    void release();
};

template <typename T>
int Solver::add_nodes(T *x, T *m, int n_verts) {
    const int prev_n = m_x.size(), n3 = n_verts * 3;
    m_x.conservativeResize(prev_n + n3);
    m_v.conservativeResize(prev_n + n3);
    m_masses.conservativeResize(prev_n + n3);
    for (int i = 0; i < n3; ++i) { m_x[prev_n + i] = x[i]; m_v[prev_n + i] = 0.0; m_masses[prev_n + i] = m[i]; }
    return (prev_n + n3) / 3;
}

} // namespace admm
#endif
