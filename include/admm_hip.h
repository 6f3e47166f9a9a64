/*
 * admm_hip.h -- C ABI of the MI355X-native ADMM elastic hot path (libadmm_hip.so).
 *
 * The reference (mattoverby/admm-elastic) has no FFI of its own: its extension points are the C++
 * classes admm::Solver / EnergyTerm / LinearSolver.  This header is the boundary those classes'
 * MI355X-native re-implementation (admm-elastic_amd/host/) sits on, and the only thing the Python
 * harness binds (ctypes).  Plain pointers and sizes, no C++/torch types.  Each entry point cites the
 * reference interface it replaces (file:line relative to the reference repo root).
 *
 * Conventions
 *   - all floating point data is FP64, all indices int32
 *   - node vectors (x, v, b, masses) are [3*n_verts], interleaved xyz, exactly like Solver::m_x
 *     (src/Solver.hpp:66-68)
 *   - 3x3 / 2x2 matrices are column-major (Eigen default)
 *   - energy-term order inside a context is: tets (given order), then tris, then bending hinges, then pins; the ADMM
 *     vectors z,u use the reference's row layout (9 rows per tet, 6 per tri, 3 per hinge, 6 per pin:
 *     src/TetEnergyTerm.hpp:68, src/TriEnergyTerm.hpp:66, src/SpringEnergyTerm.hpp:42)
 *   - every function returns 0 on success or a negative admm_hip_status; the message of the last
 *     failure on the calling thread is available from admm_hip_last_error()
 *   - a context is bound to one HIP device, owns all of its device memory, is not thread-safe
 *   - functions named admm_host_* do not touch the GPU (set-up arithmetic shared by every caller)
 */
#ifndef ADMM_HIP_H
#define ADMM_HIP_H 1

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ADMM_HIP_OK = 0,
    ADMM_HIP_ERR_ARG = -1,      /* bad argument / inconsistent description  (reference: std::runtime_error) */
    ADMM_HIP_ERR_DEVICE = -2,   /* no usable HIP device, or a HIP runtime call failed */
    ADMM_HIP_ERR_GEOMETRY = -3, /* inverted rest element (src/TetEnergyTerm.cpp:42-44, TriEnergyTerm.cpp:45-47) */
    ADMM_HIP_ERR_STATE = -4,    /* call out of order (e.g. step before set_state) */
    ADMM_HIP_ERR_COMM = -5      /* RCCL failure */
} admm_hip_status;

/* tet constitutive models -- src/TetEnergyTerm.hpp:57 (linear), :116 (NeoHookean), :142 (StVK), :176 (SplineTet).
 * SplineTet carries an xu::Spline (src/XuSpline.hpp); the three splines the reference ships are accepted with
 * : xu::NeoHookean (the default; with kappa = 0 algebraically the NH model), xu::StVK (with kappa = 0 algebraically the
 * StVK model) and xu::CoRotated (co-rotated linear: mu sum (s_i-1)^2 + lambda/2 (sum s_i - 3)^2).  tet_mu / tet_lambda are
 * the SPLINE's constants, tet_kappa its compression term (src/XuSpline.hpp:44-45; 0 = none), tet_k the tet's bulk modulus
 * (src/TetEnergyTerm.hpp:192-204).  User-defined splines have no kernel. */
enum { ADMM_TET_LINEAR = 0, ADMM_TET_NEOHOOKEAN = 1, ADMM_TET_STVK = 2, ADMM_TET_SPLINE_NH = 3, ADMM_TET_SPLINE_STVK = 4,
       ADMM_TET_SPLINE_COROTATED = 5,
       ADMM_TET_SPLINE_TABLE = 6,     /* SplineTet with a USER-DEFINED xu::Spline (src/TetEnergyTerm.hpp:197-204, src/XuSpline.hpp:34-46):
                                       * the spline's six functions sampled by admm_host_tabulate_spline, desc.tet_spline = its table */
       ADMM_TET_STABLE_NH = 7 };      /* stable Neo-Hookean (README.md:23-28 lists it as a TODO the reference never shipped; it would sit beside
                                       * NeoHookeanTet, src/TetEnergyTerm.hpp:116-136, as one more HyperElasticTet): Smith, de Goes, Kim 2018,
                                       * Psi = mu_s/2 (I_C - 3) + la_s/2 (J - alpha)^2 - mu_s/2 log(I_C + 1) with mu_s = 4/3 mu, la_s = lambda + 5/6 mu,
                                       * alpha = 1 + 3 mu_s / (4 la_s) from the tet's tet_mu / tet_lambda; finite for inverted elements */
#define ADMM_SPLINE_TABLE_DOUBLES 9228   /* doubles of one table: 3 functions x (4 header + 3 x 1024 node values) */

/* global solvers -- Solver::Settings::linsolver, src/Solver.hpp:46 ("0=LDLT, 1=NCMCGS, 2=UzawaCG").
 * 0: the prefactored LDLT (src/LinearSolver.hpp:59-92) is replaced by a preconditioned CG (Jacobi, or a block-local
 *    symmetric Gauss-Seidel sweep on 2-colourable meshes) on
 *    the GPU iterated to pcg_tol; 1: nodal multi-colour SOR (src/NodalMultiColorGS.hpp); 2: Schur-
 *    complement CG (src/UzawaCG.hpp) whose inner LDLT solves are the same GPU PCG. */
enum { ADMM_LS_LDLT_AS_PCG = 0, ADMM_LS_NCMCGS = 1, ADMM_LS_UZAWACG = 2 };

/* passive obstacles -- src/PassiveObject.hpp:32-45 (Floor: params[0]=y), :48-64 (Sphere: cx,cy,cz,r).
 * User-defined PassiveCollision subclasses (src/Collider.hpp:66-83 is an interface of one virtual function) reach the kernels in two
 * forms: ADMM_OBJ_PLANE, the half space n.x < d (params nx, ny, nz, d; n is normalised by the library; signed distance n.x - d, contact
 * point x - dx n), and ADMM_OBJ_GRID, ANY object sampled at Solver::initialize (params[0] = index of its grid in
 * desc.obstacle_grid_meta; admm_host_sample_obstacle fills the grid from the object's own signed_distance): the kernels interpolate
 * distance and normal trilinearly and take the contact point as x - dx n (exact for a true signed distance function, O(h^2)
 * otherwise); outside its grid an object is never hit. */
enum { ADMM_OBJ_FLOOR = 0, ADMM_OBJ_SPHERE = 1, ADMM_OBJ_PLANE = 2, ADMM_OBJ_GRID = 3 };

typedef struct admm_hip_ctx admm_hip_ctx;

/* Everything Solver::initialize (src/Solver.cpp:167-261) consumes, flattened to arrays. */
typedef struct {
    int32_t struct_size;          /* = sizeof(admm_hip_desc), for forward compatibility */
    int32_t device;               /* HIP device ordinal */

    int32_t n_verts;
    const double *masses;         /* [3*n_verts]  Solver::m_masses (src/Solver.hpp:68) */
    double dt;                    /* Settings::timestep_s (src/Solver.hpp:42); <=0 -> 1/24 like Solver.cpp:175-179 */

    /* TetEnergyTerm family (src/TetEnergyTerm.cpp:31-71) */
    int32_t n_tets;
    const int32_t *tet_idx;       /* [4*n_tets] */
    const double *tet_Binv;       /* [9*n_tets] edges_inv, column-major */
    const double *tet_weight;     /* [n_tets]   w = sqrt(k*vol) */
    const int32_t *tet_kind;      /* [n_tets]   ADMM_TET_* */
    const double *tet_mu;         /* [n_tets]   Lame::mu */
    const double *tet_lambda;     /* [n_tets]   Lame::lambda */
    const double *tet_k;          /* [n_tets]   Lame::bulk_modulus() */

    /* TriEnergyTerm (src/TriEnergyTerm.cpp:29-101) */
    int32_t n_tris;
    const int32_t *tri_idx;       /* [3*n_tris] */
    const double *tri_rest;       /* [4*n_tris] rest_pose (2x2, column-major) */
    const double *tri_weight;     /* [n_tris] */
    const double *tri_limit_min;  /* [n_tris] Lame::limit_min */
    const double *tri_limit_max;  /* [n_tris] Lame::limit_max */

    /* Pins.  linsolver 0/2: every pin becomes a SpringPin energy term (src/Solver.cpp:190-196,
     * src/SpringEnergyTerm.hpp:31-73) with weight pin_weight (<=0 -> sqrt(2*k_rubber)).
     * linsolver 1: pins are applied inside the GS sweeps (src/NodalMultiColorGS.hpp:111-117). */
    int32_t n_pins;
    const int32_t *pin_vert;      /* [n_pins] */
    const double *pin_xyz;        /* [3*n_pins] */
    const int32_t *pin_active;    /* [n_pins] or NULL (= all active) */
    double pin_weight;

    int32_t linsolver;            /* ADMM_LS_* */
    double constraint_w;          /* Settings::constraint_w; <=0 -> auto (src/Solver.cpp:235,239,245) */

    int32_t pcg_max_iters;        /* <=0 -> 500.  Iterations are launched in chunks, one chunk ahead of the GPU,
                                   * and stop as soon as the device signals convergence (no stream sync). */
    double pcg_tol;               /* relative residual of the (Jacobi-scaled) system; <=0 -> 1e-10 */
    int32_t gs_max_iters;         /* <=0 -> 30   (NodalMultiColorGS::max_iters, NodalMultiColorGS.hpp:45-46) */
    double gs_tol;                /* <0 -> 1e-10 (m_tol); 0 disables the residual test like the reference */
    double gs_omega;              /* <=0 -> 1.9  (m_omega) */
    int32_t uzawa_max_iters;      /* <=0 -> 20   (UzawaCG::max_iters, UzawaCG.hpp:44-45) */
    double uzawa_tol;             /* <=0 -> 1e-10 */

    int32_t n_obstacles;          /* Solver::add_obstacle (src/Solver.cpp:159-161); needs linsolver 1 or 2 */
    const int32_t *obstacle_kind; /* [n_obstacles] ADMM_OBJ_* */
    const double *obstacle_params;/* [4*n_obstacles] */

    const int32_t *gs_colors;     /* optional [n_verts] colour per node; NULL -> greedy (admm_host_greedy_coloring) */

    /* multi-GPU: this context is rank `rank` of `world_size` (0 or 1 = single GPU).  Every rank is
     * given the FULL description; the library keeps the element block admm_host_partition assigns to
     * the rank (tets and tris separately), assembles the full matrix, and replicates the global solve. */
    int32_t rank;
    int32_t world_size;

    /* SplineTet with a spline constructed with a compression term (src/XuSpline.hpp:43-45, the `kappa` of xu::NeoHookean /
     * xu::StVK / xu::CoRotated): [n_tets] or NULL (= 0 everywhere, what the reference's own SplineTet constructors pass,
     * src/TetEnergyTerm.hpp:194).  Read for the ADMM_TET_SPLINE_* kinds only. */
    const double *tet_kappa;

    /* Optional [3 * n_verts]: any SMOOTH coordinates of the vertices -- Solver::m_x as it stands at Solver::initialize
     * (src/Solver.cpp:167), which is what both host mirrors pass.  Used ONLY to build the coarse space of the on-chip PCG's
     * two-level preconditioner: with coordinates every block carries the four functions {1, x, y, z} (the low-energy modes of
     * the Laplacian-like Ahat) instead of four piecewise-constant aggregates -- same 4 unknowns per block, the smooth error
     * modes that dominate the POSITION error of an iterate are represented to second order.  NULL = piecewise constants.
     * The solution of the system does not depend on it (a preconditioner). */
    const double *vert_xyz;

    /* Tabulated user splines (ADMM_TET_SPLINE_TABLE): n_spline_tables tables of ADMM_SPLINE_TABLE_DOUBLES doubles each, made by
     * admm_host_tabulate_spline; tet_spline [n_tets]: table of every tet (read for that kind only).  tet_mu / tet_lambda of such
     * tets are not used; tet_k is the tet's bulk modulus as for every kind. */
    int32_t n_spline_tables;
    const double *spline_tables;
    const int32_t *tet_spline;

    /* Sampled obstacles (ADMM_OBJ_GRID): obstacle_grid_meta [10 * n_obstacle_grids] = origin xyz, spacing xyz, nodes nx ny nz
     * (each >= 2), index of the grid's first node in obstacle_grid_data; obstacle_grid_data [4 per node, x fastest] = signed
     * distance, normal xyz.  Both made by admm_host_sample_obstacle. */
    int32_t n_obstacle_grids;
    const double *obstacle_grid_meta;
    const double *obstacle_grid_data;

    /* SLIDE constraints (README.md:23-28 lists them as a TODO the reference never shipped; no reference code).  [3*n_pins] or NULL:
     * a pin with a non-zero normal n (normalised by the library) constrains only n . (x - pin_xyz) = 0 -- the vertex slides in the
     * plane through the pin's point.  linsolver 0 / 2: a SpringPin term (src/SpringEnergyTerm.hpp:31-73: same D-block I3, same
     * weight) whose prox projects q = D x + u onto that plane instead of onto the point (:61); linsolver 1: inside the sweeps
     * (src/NodalMultiColorGS.hpp:111-117) the node takes the plane-constrained Jacobi value of :218-262 on its own plane.  A zero
     * normal = an ordinary pin.  admm_hip_set_pins moves the points and keeps the normals; admm_hip_set_pin_normals changes them. */
    const double *pin_normal;

    /* BENDING terms for cloth (README.md:23-28 TODO "bending force"; no reference code).  One EnergyTerm per hinge -- an interior edge
     * (v0, v1) and the two vertices v2, v3 opposite to it -- beside TriEnergyTerm (src/TriEnergyTerm.cpp:54-101): dim 3, D-block =
     * (c0, c1, c2, c3) (x) I3, i.e. D_i x = sum_k c_k x_{v_k} (the discrete mean-curvature normal of the hinge, Bergou et al. 2006, "A
     * Quadratic Bending Model for Inextensible Surfaces"), energy E(z) = bend_stiffness / 2 |z|^2, weight bend_weight, so that
     * prox(q) = q w^2 / (stiffness + w^2).  bend_idx [4*n_bends], bend_coef [4*n_bends], bend_weight [n_bends] (> 0), bend_stiffness
     * [n_bends] (>= 0).  admm_host_bend_hinges builds idx / coef / rest areas from a triangle mesh; the mirrors then use
     * stiffness = k_bend * 3 / area and weight = sqrt(stiffness) (prox = q / 2, the idiom of src/TriEnergyTerm.cpp:77-83).
     * Rows of z / u: after the triangles, before the pins, 3 per hinge. */
    int32_t n_bends;
    const int32_t *bend_idx;
    const double *bend_coef;
    const double *bend_weight;
    const double *bend_stiffness;
} admm_hip_desc;

/* Maps 1:1 onto Solver::RuntimeData (src/Solver.hpp:54-61) plus GPU-side extras. */
typedef struct {
    double global_ms;
    double local_ms;              /* the per-element prox kernels only (HIP events on the context's stream) */
    double collision_ms;
    int32_t inner_iters;
    int32_t admm_iters;
    double step_ms;               /* whole step, HIP events */
    int32_t last_solve_converged; /* PCG: residual test met within pcg_max_iters in the last ADMM iteration */
    int32_t n_constraints;        /* rows of C in the last ADMM iteration (UzawaCG) */
    double rhs_ms;                /* part of global_ms spent assembling b = M x_bar + dt^2 D^T W^2 (z-u) */
    int32_t unconverged_solves;   /* PCG solves of this step that did not meet pcg_tol (0 = every solve converged) */
    int32_t pcg_launched_iters;   /* PCG iterations launched for the last solve (converged ones exit early on the device) */
    int32_t pcg_iters_per_solve[64]; /* linsolver 0: PCG iterations of each ADMM iteration's solve (first 64) */
    double local_kernel_ms;       /* sum of the tet local-step KERNEL durations of this step, from the device wall clock
                                   * (every wave stamps its entry and, after its stores have drained, its exit; duration =
                                   * max exit - min entry): what rocprofv3 --kernel-trace reports.  local_ms above is the
                                   * PHASE between two event records, dispatch gaps included.  Opt-in (ADMM_HIP_KERNEL_CLOCK=1:
                                   * the stamps cost the launch ~2 %); 0 otherwise, or when there are no tets. */
} admm_hip_stats;

const char *admm_hip_last_error(void);

/* Number of visible HIP devices (0 when there is no GPU / no driver). Never fails. */
int admm_hip_device_count(void);

/* Solver::initialize (src/Solver.cpp:167-261): validates, uploads, assembles D / W / A on the host,
 * builds the SpMV / gather structures, colours the graph (linsolver 1). */
int admm_hip_create(const admm_hip_desc *desc, admm_hip_ctx **out);
void admm_hip_destroy(admm_hip_ctx *ctx);

/* Solver::m_x / m_v access (src/Solver.hpp:66-67).  Host <-> device copies. */
int admm_hip_set_state(admm_hip_ctx *ctx, const double *x, const double *v);
int admm_hip_get_state(admm_hip_ctx *ctx, double *x, double *v);

/* Solver::set_pins after initialize (src/Solver.cpp:113-157): move / activate / deactivate pins.
 * linsolver 0/2: idx must be among the pins given at create (else ADMM_HIP_ERR_ARG, like the throw
 * at Solver.cpp:147-151); all other pins become inactive.  linsolver 1: the set is replaced freely. */
int admm_hip_set_pins(admm_hip_ctx *ctx, int32_t n, const int32_t *vert, const double *xyz);

/* Slide pins after initialize (see desc.pin_normal): normals [3*n] of the pins at `vert` (a zero normal makes it an ordinary pin again).
 * linsolver 0 / 2: vert must be among the pins given at create; linsolver 1: among the current pins.  Points and activation are
 * admm_hip_set_pins' business. */
int admm_hip_set_pin_normals(admm_hip_ctx *ctx, int32_t n, const int32_t *vert, const double *normals);

/* Solver::ext_forces with a WindForce (src/ExplicitForce.hpp:39-46, src/ExplicitForce.cpp:47-104), applied on the DEVICE at the
 * start of every admm_hip_step, where Solver::step calls ExplicitForce::project (src/Solver.cpp:54): tris [3*n_tris] = the node
 * indices of the triangles the wind acts on (WindForce's constructor argument), direction [3] = WindForce::direction.  Every
 * triangle sees the velocities of the start of the step (the reference's OpenMP loop reads whatever its critical sections have
 * already written).  n_tris = 0 removes the force.  For callers that keep m_x / m_v on the host the C++ mirror applies the same
 * force on the host, like the reference. */
int admm_hip_set_wind(admm_hip_ctx *ctx, int32_t n_tris, const int32_t *tris, const double *direction);

/* Solver::surface_inds (src/Solver.hpp:70; filled by binding::add_tetmesh, samples/utils/AddMeshes.hpp:131-137): the
 * vertices Collider::detect tests (src/Collider.hpp:157,163).  n = 0 -> every vertex (the reference's rule for an
 * empty list).  Applies to passive detection of the UzawaCG path and to dynamic detection. */
int admm_hip_set_surface_inds(admm_hip_ctx *ctx, int32_t n, const int32_t *inds);

/* Solver::add_dynamic_collider(std::make_shared<TetMeshCollision>(mesh, vert_offset)) -- src/Solver.cpp:163-165,
 * src/DynamicObject.hpp:45-64.  rest_verts [3*n_verts], tets [4*n_tets] and faces [3*n_faces] (the surface
 * triangles, outward) index the mesh's OWN vertices; node id = local id + vert_offset.  At every ADMM iteration
 * Collider::detect (src/Collider.hpp:166-168,192-201) refits the tet tree and runs TetMeshCollision::signed_distance
 * (src/DynamicObject.hpp:72-119) for every candidate vertex; hits become the dynamic rows of
 * ConstraintSet::make_matrix (src/ConstraintSet.hpp:92-110): hard constraints of the Schur CG with linsolver 2
 * (src/UzawaCG.hpp), the penalty A + C^T C swept by the multi-colour GS with linsolver 1
 * (src/NodalMultiColorGS.hpp:75-86; the matrix is not formed and not re-coloured as a whole: only the nodes of the
 * hits are re-coloured, see csrc/dyn_collide.hpp).  linsolver 0 fails with the reference's "No collisions with LDLT
 * solver" (src/Solver.cpp:249-254).  Call after admm_hip_create, in add_dynamic_collider order. */
int admm_hip_add_dynamic_tetmesh(admm_hip_ctx *ctx, int32_t vert_offset, int32_t n_verts, const double *rest_verts,
                                 int32_t n_tets, const int32_t *tets, int32_t n_faces, const int32_t *faces);

/* Kernel-level entry point (parity tests): Collider::detect of the dynamic objects at x [3*n_verts] (host).
 * Up to cap hits are written in candidate order: vert[h], face[3h] (node ids), barys[3h], normal[3h], dx[h]
 * = DynamicCollision::Payload (src/Collider.hpp:40-53); *n_hits = number found (may exceed cap). */
int admm_hip_detect_dynamic(admm_hip_ctx *ctx, const double *x, int32_t cap, int32_t *n_hits, int32_t *vert, int32_t *face,
                            double *barys, double *normal, double *dx);

/* Solver::step (src/Solver.cpp:35-110) on the device-resident state: gravity, x_bar, admm_iters x
 * {local step, collision detection, RHS, global solve}, velocity update.  No host<->device traffic
 * besides the launch stream; stats may be NULL. */
int admm_hip_step(admm_hip_ctx *ctx, int32_t admm_iters, double gravity, admm_hip_stats *stats);

/* Kernel-level entry points (parity tests).
 * local step = the OpenMP loop at src/Solver.cpp:84-87 over EnergyTerm::update
 * (src/EnergyTerm.hpp:130-140): given x and u (in/out) produce z, u in the reference row layout.
 * If Mxbar and b_out are non-NULL also returns b = Mxbar + dt^2 D^T W^2 (z-u) (src/Solver.cpp:98).
 * Multi-GPU contexts fill only the rows of their own element block; before admm_hip_comm_init they
 * return their partial b (rank 0 carries Mxbar and the pin terms), afterwards the all-reduced one. */
int admm_hip_local_step(admm_hip_ctx *ctx, const double *x, double *u_inout, double *z_out,
                        const double *Mxbar, double *b_out);
/* LinearSolver::solve (src/LinearSolver.hpp:46): x is in/out (warm start), returns inner iterations
 * through *iters.  Uses the context's linsolver, pins and obstacles. */
int admm_hip_global_solve(admm_hip_ctx *ctx, const double *b, double *x_inout, int32_t *iters);

/* Totals of the on-chip PCG since admm_hip_create (no reference counterpart): solves launched, solves that met pcg_tol,
 * inner iterations.  Synchronises the context's stream.  Lets a caller run admm_hip_step WITHOUT per-step statistics
 * (asynchronously, no events between the kernels) and still check afterwards that every solve converged.  -1 when the context
 * does not use the general-mesh on-chip PCG. */
int admm_hip_solve_totals(admm_hip_ctx *ctx, int64_t *solves, int64_t *converged, int64_t *inner_iters);   /* returns ADMM_HIP_OK; the three values are -1 when this context's solver keeps no totals */

/* The LinearSolver objects' public tuning members, changed AFTER Solver::initialize.  The reference reads them on every solve --
 * NodalMultiColorGS::max_iters / m_tol / m_omega (src/NodalMultiColorGS.hpp:40-46, used at :100 and :136-140), UzawaCG::max_iters /
 * m_tol (src/UzawaCG.hpp:44-45, used at :92 and :109) -- so a caller may cast Solver::m_linsolver and change them between steps.
 * kind = ADMM_LS_NCMCGS (max_iters, tol, omega), ADMM_LS_UZAWACG (max_iters, tol) or ADMM_LS_LDLT_AS_PCG (max_iters, tol of the GPU
 * PCG that stands for the prefactored solve, with linsolver 0 or 2); ADMM_HIP_ERR_ARG when the context does not run that solver.
 * max_iters <= 0, tol < 0 (GS: tol = 0 switches the residual test off, like the reference; the others: tol <= 0) and omega <= 0 keep
 * the current value.  In effect from the next solve.  admm_hip_get_solver_params reads the values in effect (pointers may be NULL). */
int admm_hip_set_solver_params(admm_hip_ctx *ctx, int32_t kind, int32_t max_iters, double tol, double omega);
int admm_hip_get_solver_params(const admm_hip_ctx *ctx, int32_t kind, int32_t *max_iters, double *tol, double *omega);

/* End projection of every PCG solve on SOFT MODES (linsolver 0 / 2; no reference counterpart -- the reference solves exactly).  Z [k][n_verts]:
 * k <= 64 smooth scalar fields, normally the lowest eigenvectors of K = diag(m) + Ahat (one field serves the three axes).  After every solve
 * of the ADMM loop the iterate is corrected by the exact Galerkin step x += Z (Z^T K Z)^-1 Z^T (b - A x), which removes the error a
 * residual-norm stop leaves in span(Z) -- the soft modes in which the error of an inexact solve is largest and accumulates from frame to
 * frame.  The same step is also taken IN FRONT of the second solve of every frame (ADMM's transient: what the recycled warm start leaves of its
 * entry error lies in these modes; ADMM_HIP_DEFL_START=mask at create chooses other solves, 0 none).  The result of a converged solve changes
 * only within the solver's tolerance.  k = 0 removes the modes. */
int admm_hip_set_soft_modes(admm_hip_ctx *ctx, int32_t k, const double *Z);
/* ... with the modes computed by the library: the k lowest eigenvectors of K by `iters` (<= 0: 8) steps of inverse subspace iteration on
 * the context's own PCG (a few seconds at 1 M tets, once per scene -- A never changes after Solver::initialize, src/Solver.cpp:225-226).
 * With the on-chip PCG and k <= 32 the projection is part of the solve's one persistent launch (one more all-to-all).  Multi-rank contexts:
 * every rank computes the modes of the system it solves (element blocks: the replicated system; components: its own bodies); refused with the
 * distributed solve (ADMM_HIP_DIST_SOLVE=1). */
int admm_hip_compute_soft_modes(admm_hip_ctx *ctx, int32_t k, int32_t iters);
/* the modes in effect: *k, and Z [k][n_verts] when Z is not NULL */
int admm_hip_get_soft_modes(admm_hip_ctx *ctx, int32_t *k, double *Z);

/* Contact work since admm_hip_create (measurement: that a timed region really exercised the collision path).  linsolver 2: rows of C
 * (ConstraintSet::make_matrix, src/ConstraintSet.hpp:59-116) summed over all UzawaCG solves; linsolver 1: node updates replaced by the
 * plane-constrained update of src/NodalMultiColorGS.hpp:218-262 (a row projected onto a passive obstacle), summed over all sweeps
 * (passive obstacles; dynamic hits of the GS path are penalty rows and not counted); linsolver 0: 0.  Synchronises the stream (GS). */
int admm_hip_contact_totals(admm_hip_ctx *ctx, int64_t *rows);

/* Launches of the three persistent solver kernels since admm_hip_create (no reference counterpart): the on-chip PCG (one per solve of
 * linsolver 0 / 2 and per batch of K^-1 columns), the multi-colour GS, the Schur CG of UzawaCG.  After a barrier / hand-off time-out
 * the context falls back to the launch-per-iteration kernels for good: the counts then stop growing (tests). */
int admm_hip_persistent_launches(const admm_hip_ctx *ctx, int64_t *pcg, int64_t *gs, int64_t *schur);

/* What the on-chip PCG (k_pcg2) has found out about this context since admm_hip_create and acts on for good (no reference counterpart; diagnostics,
 * tests): *smoother_given_up -- a pipelined pass broke off with negative or non-finite sums far from convergence, later solves run with the Jacobi
 * smoother instead of the block-local Chebyshev polynomial (slower, as exact); *trust_revoked -- a sampled verification of a short first pass
 * failed, every later solve verifies its residual; *failed_checks -- how many such checks failed.  Any pointer may be NULL.  The solves of
 * admm_hip_compute_soft_modes (right-hand sides that are nearly eigenvectors) do not count: their findings are put back (round 6: until then
 * they switched the smoother off for every context that computed its modes).  Synchronises the stream. */
int admm_hip_pcg_findings(admm_hip_ctx *ctx, int32_t *smoother_given_up, int32_t *trust_revoked, int64_t *failed_checks);

/* Diagnostics of the on-chip PCG (linsolver 0 / 2; no reference counterpart): the latency floor of the two
 * synchronisations one CG iteration consists of, measured on this context's grid with the kernel's own primitives and
 * payloads but no arithmetic, as microseconds per repetition over n repetitions: the all-to-all (block record -> grid barrier
 * -> every block reads every record) and the vector exchange (publish 32 B per row -> neighbour flags -> halo fetch).  0 when
 * the context has no on-chip plan.  plan_stats [6] (may be NULL): off-diagonal non-zeros, stored SELL entries, entries held in
 * LDS, block-local non-zeros, most neighbour blocks of a block, coarse unknowns of the two-level preconditioner. */
int admm_hip_probe_sync(admm_hip_ctx *ctx, int32_t n, double *us_all_to_all, double *us_exchange, int64_t *plan_stats);

/* Measurement only (bench.py `roofline`; no counterpart in the reference, whose MicroTimer brackets the whole local loop,
 * src/Solver.cpp:83-88).  on = 1: every later admm_hip_step WITHOUT statistics records a hipEvent pair around the launches of
 * its local step (all constitutive models of one ADMM iteration) on the context's stream -- nothing else changes, no
 * synchronisation is added.  admm_hip_local_launch_times synchronises the stream and returns the number of pairs recorded since
 * the last call and the sum of their intervals in milliseconds (kernel + the two dispatch gaps of the pair).
 * on = 2: the pair is attached to the dispatch of the dominant local-step kernel instead (hipExtLaunchKernelGGL start / stop events:
 * the kernel's own begin and end time stamps, the duration a profiler reports for it). */
int admm_hip_time_local_launches(admm_hip_ctx *ctx, int32_t on);
int admm_hip_local_launch_times(admm_hip_ctx *ctx, int64_t *n_pairs, double *sum_ms);

/* Sizes: R = rows of D (9*n_tets + 6*n_tris + 6*n_pin_terms).  A context of the component partition (whole bodies per rank) holds
 * only its rank's bodies: it returns the rows of THOSE (the kernel-level entry points that use this layout refuse such a context);
 * step statistics are per rank as well. */
int admm_hip_num_rows(const admm_hip_ctx *ctx);

/* The assembled scalar system matrix Ahat (A = M + Ahat (x) I3, src/Solver.cpp:225-226 with the
 * x/y/z replication of src/TetEnergyTerm.cpp:66-68 factored out), CSR with sorted columns.
 * Pass NULL pointers to query nnz only. */
int admm_hip_get_matrix(const admm_hip_ctx *ctx, int32_t *rowptr, int32_t *col, double *val, int32_t *nnz);
/* colour per node used by linsolver 1 */
int admm_hip_get_colors(const admm_hip_ctx *ctx, int32_t *color, int32_t *n_colors);

/* ---- multi-GPU (one context per rank; new work, no reference counterpart -- SURVEY.md 8e) ----
 * Element-block partition: rank r owns the contiguous block of tets/tris admm_host_partition gives
 * it; every rank holds all vertices and replicates the global solve; the only exchange is one RCCL
 * sum all-reduce of the [3*n_verts] partial right-hand side per ADMM iteration. */
int admm_hip_comm_unique_id(char *id128);                                   /* ncclGetUniqueId */
int admm_hip_comm_init(admm_hip_ctx *ctx, const char *id128, int rank, int world_size);
/* What the context's RCCL communicator says about itself (ncclCommCount / ncclCommUserRank: 0 / -1 without a communicator) and which
 * device the context sits on (device_id64: at least 64 chars, "pci <bus id> uuid <hex>") -- printed by bench.py in its N > 1 line so
 * that a scaling run can be checked for N distinct devices in ONE communicator.  Pointers may be NULL. */
int admm_hip_comm_info(const admm_hip_ctx *ctx, int32_t *n_ranks, int32_t *rank, char *device_id64);
/* The same exchange over the CALLER's transport instead of RCCL (MPI, gloo, a test harness that puts several ranks on one device
 * -- which RCCL refuses): once per ADMM iteration the library copies the partial right-hand side to pinned host memory, calls
 * fn(user, host_buf, 3 * n_verts) -- which must sum host_buf in place over all ranks and return 0 -- and copies the result back.
 * Two PCIe copies and one stream synchronisation per ADMM iteration: a functional path, not the fast one.  A context with a
 * communicator (admm_hip_comm_init) ignores it.  fn = NULL removes it. */
typedef int (*admm_allreduce_fn)(void *user, double *host_buf, int64_t n);
int admm_hip_set_rhs_allreduce(admm_hip_ctx *ctx, admm_allreduce_fn fn, void *user);
void admm_host_partition(int32_t n_items, int world_size, int rank, int32_t *begin, int32_t *end);

/* A user-defined xu::Spline on the device (src/XuSpline.hpp:34-46 is an interface of six virtual functions; the reference calls them
 * inside its L-BFGS, src/TetEnergyTerm.cpp:243-265).  `fn(user, which, x)` returns f, g, h, df, dg, dh (which = 0..5) at x > 0;
 * the functions are sampled over stretches in [s_min, s_max] (f), their pairwise products (g) and triple products (h) on grids
 * uniform in ln x, second derivatives by central differences of df / dg / dh, into table_out [ADMM_SPLINE_TABLE_DOUBLES].  The
 * kernels evaluate the C2 quintic Hermite interpolant (relative error of the energy gradient ~1e-10 for splines as smooth as
 * the reference's).  Returns ADMM_HIP_ERR_ARG for a bad range or a non-finite sample.  admm_host_spline_table_eval evaluates a
 * table as the device does: out3 = {F, F', F''} of function which (0 f, 1 g, 2 h) at x (tests). */
typedef double (*admm_spline_fn)(void *user, int which, double x);
int admm_host_tabulate_spline(admm_spline_fn fn, void *user, double s_min, double s_max, double *table_out);
void admm_host_spline_table_eval(const double *table, int which, double x, double *out3);

/* Multi-GPU, component-aware partition (the reference has no distributed layer).  When a scene has at least
 * world_size connected components (bodies that share no vertex), admm_hip_create gives every rank WHOLE bodies: its context
 * holds only them, renumbered locally, and steps them like a single-GPU scene -- no exchange inside a step, the global solve
 * is the rank's block of the block-diagonal system matrix.  The caller keeps the global numbering (set_state / get_state /
 * set_pins translate; admm_hip_get_state merges the ranks' parts over RCCL when admm_hip_comm_init was called -- it is then a
 * COLLECTIVE call: every rank must make it, each runs both merges whatever pointers it passes).  Loose vertices (no element) are not
 * bodies; a scene whose largest body holds more than twice a rank's fair share of the elements uses element blocks instead.  Components go
 * to ranks by decreasing element count, each to the least loaded rank so far.  Otherwise (a single body, or
 * ADMM_HIP_PARTITION=elements) the element-block partition above is used.  This host-only function returns the number of
 * components and, in vertex_rank [n_verts], the rank that would own every vertex. */
int32_t admm_host_component_partition(const admm_hip_desc *desc, int world_size, int32_t *vertex_rank);

/* ---- host-side set-up arithmetic (no GPU) ---- */
/* The matrix admm_hip_create would assemble for this description (same code path), without a GPU:
 * Ahat in CSR (see admm_hip_get_matrix).  Pass NULL arrays to query nnz. */
int admm_host_assemble_matrix(const admm_hip_desc *desc, int32_t *rowptr, int32_t *col, double *val, int32_t *nnz);
/* TetEnergyTerm ctor (src/TetEnergyTerm.cpp:31-48): Binv [9*n], vol [n]; ADMM_HIP_ERR_GEOMETRY if a
 * rest volume is negative.  verts [3*nv], idx [4*n]. */
int admm_host_tet_rest(int32_t n, const int32_t *idx, const double *verts, double *Binv, double *vol);
/* Rest positions behind a set of edges_inv (no reference counterpart; src/TetEnergyTerm.cpp:31-48 builds every Binv from the
 * vertex positions it is handed).  admm_hip_create calls this on the tets it is given: when ONE set of vertex positions
 * reproduces every tet's Binv to 1e-11, the local step gathers those positions (vertex data, cache-resident) and recomputes
 * Binv instead of streaming 72 bytes per tet and launch.  candidate [3*n_verts] or NULL is tried first (admm_hip_create
 * passes desc.vert_xyz); otherwise positions are propagated from tet to tet through inv(Binv), one translation per connected
 * component.  Returns 1 (candidate), 2 (propagated), 0 (the tets do not share one set of rest positions: Binv is streamed),
 * -1 bad arguments; x0_out [3*n_verts].  ADMM_HIP_TET_REST=0 in the environment keeps Binv streamed (A/B, tests). */
int admm_host_tet_rest_positions(int32_t n_verts, int32_t n_tets, const int32_t *idx, const double *Binv,
                                 const double *candidate, double *x0_out);
/* UzawaCG's Schur iterations (src/UzawaCG.hpp:92-120) apply A^-1 to C^T d.  A never changes after initialize and C^T d is
 * non-zero only at constrained vertices, so the library keeps the columns of K^-1 (A = K (x) I3) of every vertex that has ever
 * carried a constraint row: solved once by the on-chip PCG at 1e-2 x pcg_tol (<= 1e-10), three per launch, kept in HBM
 * (grown on demand up to ADMM_HIP_UZ_CACHE_MB, default 16384).  The Schur iterations then run on the active vertices only (the
 * active x active block of K^-1; ADMM_HIP_UZ_COMPACT=0: one pass over the full-height active columns per iteration) and x is
 * updated once from the multiplier update.  ADMM_HIP_UZ_CACHE=0: every Schur iteration is a PCG solve.  columns: cached columns (-1: cache off); column_solves: PCG
 * launches spent on columns; schur_from_columns / schur_by_pcg: Schur iterations served either way since create; evicted:
 * columns given up because the cache was full (it then drops the columns of every vertex that is not active in the current solve). */
int admm_hip_uzawa_cache_stats(admm_hip_ctx *ctx, int64_t *columns, int64_t *column_solves, int64_t *schur_from_columns,
                               int64_t *schur_by_pcg, int64_t *evicted);
/* Column solves that did NOT meet their tolerance within max(pcg_max_iters, 2000) iterations since create: their batch is not cached
 * (an inexact column would be a wrong Schur operator for every later solve), the solve that asked for them applies A^-1 by inner PCG
 * solves instead.  0 in every healthy run. */
int admm_hip_uzawa_unconverged_columns(admm_hip_ctx *ctx, int64_t *n);
/* The column solves of a batch run side by side on up to 8 streams when several instances of the on-chip PCG kernel fit the chip at
 * once (small bodies: the kernel of a 20 k-vertex body holds 77 of 256 CUs and is bound by its grid barrier's latency).  The count is
 * the occupancy of the kernel, never more; ADMM_HIP_UZ_LANES=n (1..8) lowers it, 1 = the main stream only.  batches: batches solved on
 * more than one stream since create; lanes: streams set up (at create for scenes with colliders, else by the first such batch).
 * LOOK-AHEAD (opt-in: ADMM_HIP_UZ_AHEAD=frames, e.g. 4; default 0 = off; admm_hip_step, passive objects, bodies that leave room for the
 * lanes beside the loop's own solve): in the first solve of a step the vertices that would reach an object within that many frames at their current speed get their
 * columns solved on the lanes WHILE the ADMM loop goes on; they are committed when done, a touchdown that needs one still in flight
 * waits for it.  ahead_columns: columns committed that way; ahead_waits: solves that had to wait.  Same columns to the tolerance of a
 * column (the three vertices of a launch share its iteration count), same active sets.  Trades throughput (columns of vertices that never
 * touch: -2 % over 200 frames of the bench's contact scene) for the absence of a stall in the touchdown frame. */
int admm_hip_uzawa_column_lanes(admm_hip_ctx *ctx, int64_t *batches, int *lanes, int64_t *ahead_columns, int64_t *ahead_waits);
/* A user-defined PassiveCollision on the device (src/Collider.hpp:66-83; the reference calls signed_distance per vertex inside
 * Collider::detect_passive, :137-150).  fn(user, x, out7) evaluates the object at x on a FRESH payload: out7 = signed distance, contact
 * point xyz, normal xyz.  Sampled at the nx x ny x nz nodes of the box [lo, hi] (each n >= 2) into meta10_out (node offset 0: the
 * caller adds the offset of the grid inside its obstacle_grid_data) and data_out [4 * nx * ny * nz].  Returns ADMM_HIP_ERR_ARG for a bad
 * box or a non-finite sample. */
typedef void (*admm_obstacle_fn)(void *user, const double *x3, double *out7);
int admm_host_sample_obstacle(admm_obstacle_fn fn, void *user, const double *lo3, const double *hi3, const int32_t *dims3, double *meta10_out,
                              double *data_out);
/* which of the above this context's local step uses: 0 streamed Binv, 1 / 2 rest positions; -1 NULL context */
int admm_hip_tet_rest_mode(const admm_hip_ctx *ctx);
/* TriEnergyTerm ctor (src/TriEnergyTerm.cpp:29-52): rest [4*n], area [n]. */
int admm_host_tri_rest(int32_t n, const int32_t *idx, const double *verts, double *rest, double *area);
/* Bending hinges of a triangle mesh (desc.bend_*; no reference counterpart): every interior edge shared by exactly two triangles gives
 * one hinge (v0 < v1 the edge, v2 / v3 the opposite vertices of the first / second triangle in index order), hinges sorted by (v0, v1).
 * coef = (c03 + c04, c01 + c02, -(c01 + c03), -(c02 + c04)) with the cotangents of the REST angles at v0 (c01 in the first triangle, c02 in
 * the second) and at v1 (c03, c04): sum_k coef_k x_k = 0 for any flat configuration; area = rest area of the two triangles.
 * Returns the number of hinges (may exceed cap: then only cap are written; NULL arrays to count). */
int32_t admm_host_bend_hinges(int32_t n_verts, int32_t n_tris, const int32_t *tris, const double *verts, int32_t cap,
                              int32_t *hinge_idx, double *coef, double *area);
/* Lame (src/EnergyTerm.hpp:34-59) */
void admm_host_lame(double youngs, double poisson, double *mu, double *lambda, double *bulk);
/* Greedy nodal colouring on a CSR pattern (role of the absent mcl::graphcolor::color_matrix,
 * src/NodalMultiColorGS.hpp:57): the better of first-fit in index order and DSATUR; deterministic; returns the
 * number of colours. */
int admm_host_greedy_coloring(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t *color);
/* Mesh preprocessing (no counterpart in the reference, whose CPU solvers do not care): a vertex numbering with locality.
 * The kernels gather positions, corner forces and matrix rows by vertex index, and the on-chip PCG hands vector slices
 * between the blocks that own neighbouring index ranges, so the numbering a mesh arrives with decides how coherent all of
 * that is (1 M-tet cube: lexicographic numbering 1.37 ms per ADMM iteration, randomly renumbered 3.57 ms).  Reverse
 * Cuthill-McKee on the element graph; idx = [corners * n_elems] vertex ids (corners 3 or 4); new_id[v] = new index of
 * vertex v; span_before / span_after (may be NULL) = mean |i - j| over element edges, the figure to decide by.  The
 * caller renumbers its mesh (vertices, elements, pins) BEFORE building the solver: the library never renumbers behind
 * the API (m_x keeps the caller's order). */
void admm_host_locality_order(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t *new_id,
                              double *span_before, double *span_after);

/* The plan admm_hip_create makes for the on-chip PCG of linsolver 0 / 2 (the GPU replacement of LDLTSolver::solve,
 * src/LinearSolver.hpp:87-90), without a GPU -- diagnostics and tests.  The solve renumbers the rows internally (the
 * caller's numbering is kept at the API): n_blocks compact blocks (one per CU) of slices_per_block wavefronts, every block
 * split into 4 compact aggregates that carry the coarse space of the two-level preconditioner
 * M^-1 = D^-1 + P (P^T A P)^-1 P^T.  row_vertex [64 * n_blocks * slices_per_block]: vertex of every internal row (-1 =
 * unused slot); row_aggregate (same length): coarse unknown of the row (block * 4 + aggregate); row_weights (NULL to skip)
 * [4 * rows]: the row's weights in the four coarse functions of its block, i.e. row r of P -- one-hot on the aggregate
 * without desc.vert_xyz, (1, x, y, z) centred and scaled per block with it; coarse_inv [nc * nc], nc =
 * 4 * n_blocks: (P^T A P)^-1, row-major (NULL to skip); stats [11]: off-diagonal non-zeros, stored SELL entries, entries
 * held in LDS, block-local non-zeros, most neighbour blocks of a block (-1: more than 64), coarse unknowns (0 = two-level
 * off), largest halo list, LDS slab columns used, 1e9 x the estimate of lambda_max(D^-1 A_bb) the block-local smoother is built
 * on (A_bb = entries of M + Ahat inside one block), 1e6 x the lanes on the busiest LDS bank pair per half-wavefront column with the
 * entries in index order / as placed.  lds_bytes = LDS a block may spend on its local vector and matrix slab. */
int admm_host_oc_plan(const admm_hip_desc *desc, int32_t n_blocks, int32_t slices_per_block, int32_t lds_bytes,
                      int32_t *row_vertex, int32_t *row_aggregate, double *coarse_inv, int64_t *stats, float *row_weights);

/* The plan of the LAUNCH-PATH two-level PCG (csrc/pcg_big.hpp; csrc/oc_plan.cpp: build_big_plan) for this scene, no GPU -- bodies beyond the
 * chip's LDS and the fall-back of the on-chip solver; replaces the same prefactored solve, src/LinearSolver.hpp:87-90.  The vertices are
 * dealt to G compact aggregates of `ra` rows each (at most max_aggregates, <= 0: 1024); stats [6]: G, ra, rows (G ra), coarse unknowns nc
 * = 4 G, their padded count, SELL slices.  row_vertex [rows] (-1 = unused slot), row_weights [4 rows] (row r of P, energy-orthonormal per
 * aggregate), coarse_inv [nc nc] = (P^T A P)^-1 in single precision; any of the three may be NULL (call once for the sizes). */
int admm_host_big_plan(const admm_hip_desc *desc, int32_t max_aggregates, int32_t *stats, int32_t *row_vertex, double *row_weights, float *coarse_inv);

/* The plan of the PERSISTENT multi-colour Gauss-Seidel kernel (csrc/gs_persist.hpp; csrc/oc_plan.cpp: build_gs_plan) run ON THE HOST, no
 * GPU (tests): `sweeps` plain SOR sweeps (src/NodalMultiColorGS.hpp:180-216, no pins, no obstacles) over the plan's own data -- blocks, the ELL of
 * every (block, colour) with local columns, halo lists, outbox nodes -- in the kernel's order of phases, x [3 n_verts] in and out.  A plan that
 * drops or misroutes an entry, a halo value or a boundary row gives another x than the sweeps on the assembled matrix.  color [n_verts] as
 * from admm_host_greedy_coloring; max_blocks / rows_target <= 0: 256 / 384; stats [6] (may be NULL): blocks, colours, LDS bytes of the fullest
 * block, largest halo list, most neighbour blocks, most rows of a block. */
int admm_host_gs_plan_sweeps(const admm_hip_desc *desc, int32_t n_colors, const int32_t *color, int32_t max_blocks, int32_t rows_target,
                             const double *b, double *x, int32_t sweeps, double omega, int32_t *stats);

/* Mesh preprocessing, second ordering (no counterpart in the reference): hierarchical BLOCK order.  The vertices are split
 * into compact leaves of about `leaf` vertices by recursive graph bisection (the method the on-chip PCG uses for its blocks),
 * leaves numbered in recursion-tree order, vertices breadth-first inside a leaf.  Reverse Cuthill-McKee minimises the
 * bandwidth, i.e. the index distance of neighbours; this order minimises the ACTIVE WINDOW of the per-vertex gathers (corner
 * forces, positions): on an unstructured 1 M-tet body the right-hand-side gather fetches 2x less.  Same contract as
 * admm_host_locality_order: new_id[v] = new index of vertex v, the caller renumbers its mesh before building the solver. */
void admm_host_block_order(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t leaf, int32_t *new_id);

/* Test hook for the host-side plan of the local step's block-level reduction (csrc/host_setup.hpp: TetChunks; no counterpart
 * in the reference, whose right-hand side is the sparse product D^T W^2 (z - u), src/Solver.cpp:98).  Runs the plan on the host
 * exactly as the kernels do: every chunk of 256 consecutive tets (chunks do not straddle kind_begin[k], the starts of the five
 * constitutive-model groups, kind_begin[5] = n_tets) sums its corner forces [n_tets][4][3] per vertex into records of at most 8
 * corner forces, the per-vertex lists of records are summed into vertex_sums [n_verts][3].
 * stats[0..4] = chunks, records, most 256-record passes of a chunk, widest record list, stored list entries. */
int admm_host_chunk_reduce(int32_t n_verts, int32_t n_tets, const int32_t *tet_idx, const int32_t *kind_begin, const double *corner_forces,
                           double *vertex_sums, int64_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_HIP_H */
