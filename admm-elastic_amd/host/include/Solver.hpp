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
#include "BendEnergyTerm.hpp"
#include "ExplicitForce.hpp"
#include "LinearSolver.hpp"
#include "PassiveObject.hpp"

namespace admm {

namespace solver_detail {

// Solver::Settings (src/Solver.hpp:39-50): command-line switches in the comments
struct SolverSettings {
    SolverSettings() : timestep_s(1.0 / 24.0), verbose(1), admm_iters(10), gravity(-9.8), linsolver(0), constraint_w(-1), soft_modes(0) {}
    double timestep_s;   // -dt
    int verbose;         // -v
    int admm_iters;      // -it
    double gravity;      // -g
    int linsolver;       // -ls  0 = LDLT (here: GPU PCG), 1 = NCMCGS, 2 = UzawaCG
    double constraint_w; // -ck  (-1 = automatic)
    int soft_modes;      // -sm  (GPU build, appended: the reference's fields keep their order) every PCG solve ends with an exact Galerkin step on
                         //      the k softest modes of the system matrix (admm_hip_compute_soft_modes at initialize); 0 = off
    void help();
    bool parse_args(int argc, char **argv);   // true when help() was printed
};

// Solver::RuntimeData (src/Solver.hpp:54-61): filled from admm_hip_stats after every step
struct SolverRuntimeData {
    SolverRuntimeData() : global_ms(0), local_ms(0), collision_ms(0), inner_iters(0) {}
    double global_ms, local_ms, collision_ms;
    int inner_iters;
    void print(const SolverSettings &settings);
};

} // namespace solver_detail

class Solver {
public:
    typedef solver_detail::SolverSettings Settings;
    typedef solver_detail::SolverRuntimeData RuntimeData;

    Solver();
    virtual ~Solver();

    // ---- scene data the callers fill directly (src/Solver.hpp:66-73) ----
    VecX m_x, m_v, m_masses;                                      // three entries per node
    std::vector<std::shared_ptr<EnergyTerm> > energyterms;
    std::vector<std::shared_ptr<ExplicitForce> > ext_forces;
    std::vector<int> surface_inds;                                // collision candidates (empty = every node)

    // ---- the life cycle (all virtual in the reference as well) ----
    virtual bool initialize(const Settings &settings_ = Settings());
    virtual void step();
    virtual void set_pins(const std::vector<int> &inds, const std::vector<Vec3> &points = std::vector<Vec3>());
    // Slide constraints (README.md:23-28 TODO of the reference; the counterpart of set_pins for normal-only constraints): node inds[i] may move
    // in the plane through points[i] with normal normals[i].  Replaces the current set; after initialize() only nodes that had a slide
    // constraint at initialize() may be given again (like pins with linsolver 0 / 2, src/Solver.cpp:147-151).
    virtual void set_slide_pins(const std::vector<int> &inds, const std::vector<Vec3> &points, const std::vector<Vec3> &normals);
    virtual void add_obstacle(std::shared_ptr<PassiveCollision> obj);
    virtual void add_dynamic_collider(std::shared_ptr<DynamicCollision> obj);
    virtual void save_matrix(const std::string &filename);
    virtual const RuntimeData &runtime_data() { return m_runtime; }
    const Settings &settings() { return m_settings; }

    // appends n_verts nodes (positions x, masses m, three values each); returns the new node count (src/Solver.hpp:127-141)
    template <typename T> int add_nodes(T *x, T *m, int n_verts) {
        const int old_size = m_x.size(), extra = 3 * n_verts;
        m_x.conservativeResize(old_size + extra);
        m_v.conservativeResize(old_size + extra);
        m_masses.conservativeResize(old_size + extra);
        for (int i = 0; i < extra; ++i) { m_x[old_size + i] = x[i]; m_masses[old_size + i] = m[i]; m_v[old_size + i] = 0.0; }
        return (old_size + extra) / 3;
    }

    // ---- additions of the GPU build ----
    int device;                                                   // HIP device ordinal used by initialize() (default 0)
    // User-defined PassiveCollision subclasses are sampled at initialize() on obstacle_grid_nodes^3 nodes over the box
    // [obstacle_grid_lo, obstacle_grid_hi]; an empty box (lo >= hi, the default) = the bounding box of m_x grown by half its diagonal.
    int obstacle_grid_nodes; Vec3 obstacle_grid_lo, obstacle_grid_hi;
    bool build_global_matrices;                                   // initialize() fills m_D / m_Dt / m_W_diag / solver_Dt_Wt_W (default true, like
                                                                  // the reference; the GPU path itself never reads them -- switch off for very large scenes)
    std::shared_ptr<LinearSolver> linear_solver() { return m_linsolver; }
    void *context() { return m_ctx; }                             // the admm_hip_ctx behind this solver (include/admm_hip.h), for the C ABI's extras

protected:
    void release();
    void *m_ctx;                                                  // admm_hip_ctx
    bool initialized;
    Settings m_settings;
    RuntimeData m_runtime;
    // Global matrices of src/Solver.hpp:115-121, for subclasses that read them (the reference's own step() is their only other user;
    // here the device holds its own copies in kernel layouts): reduction matrix D (rows x dof) and its transpose, the weights W,
    // dt^2 D^T W^T W (dof x rows).
    SparseMat m_D, m_Dt;
    VecX m_W_diag;
    SparseMat solver_Dt_Wt_W;
    SparseMat solver_termA;                                       // Ahat (n_verts x n_verts): the reference's solver_termA = diag(m) + Ahat (x) I3
    std::shared_ptr<ConstraintSet> m_constraints;
    std::shared_ptr<LinearSolver> m_linsolver;
    std::unordered_map<int, std::shared_ptr<SpringPin> > m_pin_energies;
    std::unordered_map<int, std::shared_ptr<SlidePin> > m_slide_energies;
    void push_pins();                                            // pins + slide constraints -> the context
};

} // namespace admm
#endif
