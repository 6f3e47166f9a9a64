// Solver.cpp -- host side of the MI355X-native admm::Solver: flattens the scene, drives the C ABI.
// Reference call stack replaced: Solver::initialize (src/Solver.cpp:167-261), Solver::step (:35-110),
// Solver::set_pins (:113-157).
#include "Solver.hpp"
#include "TetEnergyTerm.hpp"
#include "TriEnergyTerm.hpp"
#include "../../../include/admm_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace admm {

namespace {
void check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + admm_hip_last_error());
}

// gathers flat arrays for admm_hip_desc
struct Flat {
    std::vector<int32_t> tet_idx, tet_kind, tri_idx, pin_vert, pin_active, tet_spline;
    std::vector<const void *> splines;      // distinct user-defined splines, in order of first use
    std::vector<double> spline_tables;
    std::vector<double> tet_Binv, tet_w, tet_mu, tet_la, tet_k, tet_kappa, tri_rest, tri_w, tri_lmin, tri_lmax, pin_xyz, pin_nrm, bend_coef, bend_w, bend_k;
    std::vector<int32_t> bend_idx; bool any_slide = false;
    double pin_weight = 0.0;
    void add(const FlatTerm &t) {
        if (t.type == FlatTerm::TET) {
            for (int i = 0; i < 4; ++i) tet_idx.push_back(t.idx[i]);
            for (int i = 0; i < 9; ++i) tet_Binv.push_back(t.mat[i]);
            tet_w.push_back(t.weight); tet_kind.push_back(t.kind); tet_mu.push_back(t.mu); tet_la.push_back(t.lambda); tet_k.push_back(t.k);
            tet_kappa.push_back(t.kappa);
            int tab = 0;
            if (t.kind == ADMM_TET_SPLINE_TABLE) {
                tab = (int)(std::find(splines.begin(), splines.end(), t.user_spline) - splines.begin());
                if (tab == (int)splines.size()) splines.push_back(t.user_spline);
            }
            tet_spline.push_back(tab);
        } else if (t.type == FlatTerm::TRI) {
            for (int i = 0; i < 3; ++i) tri_idx.push_back(t.idx[i]);
            for (int i = 0; i < 4; ++i) tri_rest.push_back(t.mat[i]);
            tri_w.push_back(t.weight); tri_lmin.push_back(t.limit_min); tri_lmax.push_back(t.limit_max);
        } else if (t.type == FlatTerm::BEND) {
            for (int i = 0; i < 4; ++i) { bend_idx.push_back(t.idx[i]); bend_coef.push_back(t.mat[i]); }
            bend_w.push_back(t.weight); bend_k.push_back(t.k);
        } else {
            pin_vert.push_back(t.idx[0]);
            for (int i = 0; i < 3; ++i) { pin_xyz.push_back(t.pin[i]); pin_nrm.push_back(t.nrm[i]); any_slide = any_slide || t.nrm[i] != 0.0; }
            pin_active.push_back(t.active); pin_weight = t.weight;
        }
    }
    // samples every distinct user-defined spline (src/XuSpline.hpp:34-46) into its device table
    void tabulate() {
        spline_tables.assign(splines.size() * (size_t)ADMM_SPLINE_TABLE_DOUBLES, 0.0);
        for (size_t i = 0; i < splines.size(); ++i) {
            const xu::Spline *sp = (const xu::Spline *)splines[i];
            auto cb = [](void *user, int which, double x) -> double {
                const xu::Spline *q = (const xu::Spline *)user;
                return which == 0 ? q->f(x) : which == 1 ? q->g(x) : which == 2 ? q->h(x) : which == 3 ? q->df(x) : which == 4 ? q->dg(x) : q->dh(x);
            };
            if (admm_host_tabulate_spline(cb, (void *)sp, sp->table_min, sp->table_max, &spline_tables[i * (size_t)ADMM_SPLINE_TABLE_DOUBLES]) != ADMM_HIP_OK)
                throw std::runtime_error(std::string("Solver::initialize: ") + admm_hip_last_error());
        }
    }
    void fill(admm_hip_desc &d) const {
        d.n_tets = (int32_t)tet_w.size();
        d.tet_idx = tet_idx.data(); d.tet_Binv = tet_Binv.data(); d.tet_weight = tet_w.data(); d.tet_kind = tet_kind.data();
        d.tet_mu = tet_mu.data(); d.tet_lambda = tet_la.data(); d.tet_k = tet_k.data(); d.tet_kappa = tet_kappa.data();
        d.n_spline_tables = (int32_t)splines.size(); d.spline_tables = spline_tables.data(); d.tet_spline = tet_spline.data();
        d.n_tris = (int32_t)tri_w.size();
        d.tri_idx = tri_idx.data(); d.tri_rest = tri_rest.data(); d.tri_weight = tri_w.data();
        d.tri_limit_min = tri_lmin.data(); d.tri_limit_max = tri_lmax.data();
        d.n_pins = (int32_t)pin_vert.size();
        d.pin_vert = pin_vert.data(); d.pin_xyz = pin_xyz.data(); d.pin_active = pin_active.data(); d.pin_weight = pin_weight;
        d.pin_normal = any_slide ? pin_nrm.data() : nullptr;
        d.n_bends = (int32_t)bend_w.size();
        d.bend_idx = bend_idx.data(); d.bend_coef = bend_coef.data(); d.bend_weight = bend_w.data(); d.bend_stiffness = bend_k.data();
    }
};

double det3(const double *B) { // column-major
    return B[0] * (B[4] * B[8] - B[7] * B[5]) - B[3] * (B[1] * B[8] - B[7] * B[2]) + B[6] * (B[1] * B[5] - B[4] * B[2]);
}
} // namespace

// ---------------------------------------------------------------- EnergyTerm -----------------------
EnergyTerm::~EnergyTerm() {
    if (one_ctx_) admm_hip_destroy((admm_hip_ctx *)one_ctx_);
}

void EnergyTerm::get_reduction(std::vector<Triplet> &triplets, std::vector<double> &weights) {
    std::vector<Triplet> tmp;
    get_reduction(tmp);
    g_index = (int)weights.size();
    for (const Triplet &t : tmp) triplets.emplace_back(t.row() + g_index, t.col(), t.value());
    const double w = get_weight();
    if (w <= 0.0) throw std::runtime_error("**EnergyTerm::get_reduction Error: Some weight leq 0");
    for (int i = 0; i < get_dim(); ++i) weights.emplace_back(w);
}

void EnergyTerm::update(const SparseMat &D, const VecX &x, VecX &z, VecX &u) {
    (void)D; // the kernel recomputes D_i x from the term's own rest data
    FlatTerm ft;
    std::memset(&ft, 0, sizeof(ft));
    if (!flatten(ft)) throw std::runtime_error("EnergyTerm::update: this term type has no GPU kernel");
    const int nv = x.rows() / 3, dim = get_dim();
    if (!one_ctx_) {
        Flat f; f.add(ft);
        std::vector<double> masses(x.rows(), 1.0);
        admm_hip_desc d;
        std::memset(&d, 0, sizeof(d));
        d.struct_size = sizeof(d); d.n_verts = nv; d.masses = masses.data(); d.dt = 1.0; d.linsolver = 0; d.gs_tol = -1.0;
        f.tabulate();
        f.fill(d);
        admm_hip_ctx *ctx = nullptr;
        check(admm_hip_create(&d, &ctx), "EnergyTerm::update");
        one_ctx_ = ctx;
    }
    std::vector<double> ui(dim, 0.0), zi(dim, 0.0);
    for (int i = 0; i < dim; ++i) ui[i] = u[g_index + i];
    if (ft.type == FlatTerm::PIN) { // pins may move between calls
        int32_t v = ft.idx[0];
        if (ft.active) check(admm_hip_set_pins((admm_hip_ctx *)one_ctx_, 1, &v, ft.pin), "EnergyTerm::update");
        else check(admm_hip_set_pins((admm_hip_ctx *)one_ctx_, 0, nullptr, nullptr), "EnergyTerm::update");
    }
    check(admm_hip_local_step((admm_hip_ctx *)one_ctx_, x.data(), ui.data(), zi.data(), nullptr, nullptr), "EnergyTerm::update");
    for (int i = 0; i < dim; ++i) { u[g_index + i] = ui[i]; z[g_index + i] = zi[i]; }
}

double EnergyTerm::energy(const SparseMat &D, const VecX &x) {
    const int dim = get_dim();
    VecX Dx = D * x, F(dim);
    for (int i = 0; i < dim; ++i) F[i] = Dx[g_index + i];
    return energy(F);
}

double EnergyTerm::gradient(const SparseMat &D, const VecX &x, VecX &grad) {
    const int dim = get_dim();
    VecX Dx = D * x, F(dim);
    for (int i = 0; i < dim; ++i) F[i] = Dx[g_index + i];
    return gradient(F, grad);
}

// One-sided Jacobi on the host -- debugging aid for energy(); not on the hot path.
void host_signed_svd3(const double *F, double *U, double *S, double *V) {
    double G[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(G, F, sizeof(G));
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rot = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; ++r) { al += G[3 * p + r] * G[3 * p + r]; be += G[3 * q + r] * G[3 * q + r]; ga += G[3 * p + r] * G[3 * q + r]; }
                if (ga == 0.0 || std::fabs(ga) <= 1e-17 * std::sqrt(al * be)) continue;
                rot = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    double a = G[3 * p + r], b = G[3 * q + r]; G[3 * p + r] = c * a - s * b; G[3 * q + r] = s * a + c * b;
                    a = W[3 * p + r]; b = W[3 * q + r]; W[3 * p + r] = c * a - s * b; W[3 * q + r] = s * a + c * b;
                }
            }
        if (!rot) break;
    }
    int ord[3] = {0, 1, 2};
    double sig[3];
    for (int j = 0; j < 3; ++j) sig[j] = std::sqrt(G[3 * j] * G[3 * j] + G[3 * j + 1] * G[3 * j + 1] + G[3 * j + 2] * G[3 * j + 2]);
    for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) if (sig[ord[j]] > sig[ord[i]]) std::swap(ord[i], ord[j]);
    for (int j = 0; j < 3; ++j) {
        S[j] = sig[ord[j]];
        for (int r = 0; r < 3; ++r) { V[3 * j + r] = W[3 * ord[j] + r]; U[3 * j + r] = sig[ord[j]] > 0 ? G[3 * ord[j] + r] / sig[ord[j]] : (r == j ? 1.0 : 0.0); }
    }
    if (det3(U) < 0) { for (int r = 0; r < 3; ++r) U[6 + r] = -U[6 + r]; S[2] = -S[2]; }
    if (det3(V) < 0) { for (int r = 0; r < 3; ++r) V[6 + r] = -V[6 + r]; S[2] = -S[2]; }
}

// ---------------------------------------------------------------- tets ------------------------------
TetEnergyTerm::TetEnergyTerm(const Vec4i &tet_, const std::vector<Vec3> &verts, const Lame &lame_)
    : tet(tet_), lame(lame_), volume(0.0), weight(0.0) {
    const int32_t idx[4] = {0, 1, 2, 3};
    double vv[12];
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 3; ++j) vv[3 * c + j] = verts[c][j];
    if (admm_host_tet_rest(1, idx, vv, edges_inv, &volume) != 0)
        throw std::runtime_error("**TetEnergyTerm Error: Inverted initial tet"); // src/TetEnergyTerm.cpp:42-44
    weight = std::sqrt(lame.bulk_modulus() * volume);                           // :46-47
}

void TetEnergyTerm::get_reduction(std::vector<Triplet> &triplets) { // src/TetEnergyTerm.cpp:50-71
    for (int r = 0; r < 3; ++r) {
        const double d[4] = {-(edges_inv[3 * r] + edges_inv[3 * r + 1] + edges_inv[3 * r + 2]), edges_inv[3 * r], edges_inv[3 * r + 1], edges_inv[3 * r + 2]};
        for (int c = 0; c < 4; ++c)
            for (int j = 0; j < 3; ++j) triplets.emplace_back(3 * r + j, 3 * tet[c] + j, d[c]);
    }
}

bool TetEnergyTerm::flatten(FlatTerm &o) const {
    o.type = FlatTerm::TET;
    for (int i = 0; i < 4; ++i) o.idx[i] = tet[i];
    for (int i = 0; i < 9; ++i) o.mat[i] = edges_inv[i];
    o.weight = weight; o.kind = kind(); o.mu = lame.mu; o.lambda = lame.lambda; o.k = lame.bulk_modulus();
    return true;
}

double TetEnergyTerm::energy(const VecX &F) { // src/TetEnergyTerm.cpp:94-100
    double U[9], S[3], V[9];
    host_signed_svd3(F.data(), U, S, V);
    double e = 0;
    for (int i = 0; i < 3; ++i) e += (std::fabs(S[i]) - 1.0) * (std::fabs(S[i]) - 1.0);
    return 0.5 * lame.bulk_modulus() * volume * e;
}
double TetEnergyTerm::gradient(const VecX &, VecX &) { throw std::runtime_error("**TetEnergyTerm TODO: gradient function"); }

double NeoHookeanTet::energy(const VecX &F) { // src/TetEnergyTerm.cpp:138-150, :173-182
    double U[9], S[3], V[9];
    host_signed_svd3(F.data(), U, S, V);
    if (S[2] < 0) S[2] = -S[2];
    const double J = S[0] * S[1] * S[2], I1 = S[0] * S[0] + S[1] * S[1] + S[2] * S[2], l = std::log(J * J);
    return (0.5 * lame.mu * (I1 - l - 3.0) + 0.125 * lame.lambda * l * l) * volume;
}
double StVKTet::energy(const VecX &F) { // src/TetEnergyTerm.cpp:220-226
    double U[9], S[3], V[9];
    host_signed_svd3(F.data(), U, S, V);
    double tr = 0, dd = 0;
    for (int i = 0; i < 3; ++i) { const double st = 0.5 * (S[i] * S[i] - 1.0); tr += st; dd += st * st; }
    return (lame.mu * dd + 0.5 * lame.lambda * tr * tr) * volume;
}

bool SplineTet::flatten(FlatTerm &o) const {
    if (!TetEnergyTerm::flatten(o)) return false;
    int kd = 0; double m = 0, l = 0, kp = 0;
    if (!spline) return false;
    if (!spline->flatten(kd, m, l, kp)) {     // a user-defined spline: Solver::initialize tabulates the object
        o.kind = ADMM_TET_SPLINE_TABLE; o.user_spline = spline.get(); o.kappa = 0.0;
        return true;
    }
    o.kind = kd; o.mu = m; o.lambda = l; o.kappa = kp;             // the spline's constants; k stays the tet's (TetEnergyTerm.hpp:192-204)
    return true;
}
double SplineTet::energy(const VecX &F) {
    double U[9], S[3], V[9];
    host_signed_svd3(F.data(), U, S, V);
    if (S[2] < 0) S[2] = -S[2];
    return (spline->f(S[0]) + spline->f(S[1]) + spline->f(S[2]) + spline->g(S[0] * S[1]) + spline->g(S[1] * S[2]) + spline->g(S[2] * S[0]) +
            spline->h(S[0] * S[1] * S[2])) * volume;
}

// ---------------------------------------------------------------- tris / pins -----------------------
TriEnergyTerm::TriEnergyTerm(const Vec3i &tri_, const std::vector<Vec3> &verts, const Lame &lame_)
    : tri(tri_), lame(lame_), area(0.0), weight(0.0) {
    if (lame.limit_min > 1.0) throw std::runtime_error("**TriEnergyTerm Error: Strain limit min should be -inf to 1");
    if (lame.limit_max < 1.0) throw std::runtime_error("**TriEnergyTerm Error: Strain limit max should be 1 to inf");
    const int32_t idx[3] = {0, 1, 2};
    double vv[9];
    for (int c = 0; c < 3; ++c) for (int j = 0; j < 3; ++j) vv[3 * c + j] = verts[c][j];
    if (admm_host_tri_rest(1, idx, vv, rest_pose, &area) != 0)
        throw std::runtime_error("**TriEnergyTerm Error: Inverted initial pose");
    weight = std::sqrt(lame.bulk_modulus() * area);
}
void TriEnergyTerm::get_reduction(std::vector<Triplet> &triplets) {
    const double D[3][2] = {{-(rest_pose[0] + rest_pose[1]), -(rest_pose[2] + rest_pose[3])}, {rest_pose[0], rest_pose[2]}, {rest_pose[1], rest_pose[3]}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            triplets.emplace_back(i, 3 * tri[j] + i, D[j][0]);
            triplets.emplace_back(3 + i, 3 * tri[j] + i, D[j][1]);
        }
}
bool TriEnergyTerm::flatten(FlatTerm &o) const {
    o.type = FlatTerm::TRI;
    for (int i = 0; i < 3; ++i) o.idx[i] = tri[i];
    for (int i = 0; i < 4; ++i) o.mat[i] = rest_pose[i];
    o.weight = weight; o.limit_min = lame.limit_min; o.limit_max = lame.limit_max;
    return true;
}
double TriEnergyTerm::energy(const VecX &) { throw std::runtime_error("TriEnergyTerm::energy: debugging aid not provided in the MI355X build"); }
double TriEnergyTerm::gradient(const VecX &, VecX &) { throw std::runtime_error("**TriEnergyTerm TODO: gradient function"); }

bool SpringPin::flatten(FlatTerm &o) const {
    o.type = FlatTerm::PIN; o.idx[0] = idx; o.weight = weight; o.active = active ? 1 : 0;
    for (int i = 0; i < 3; ++i) { o.pin[i] = pin[i]; o.nrm[i] = 0.0; }
    return true;
}
bool SlidePin::flatten(FlatTerm &o) const {
    if (!SpringPin::flatten(o)) return false;
    for (int i = 0; i < 3; ++i) o.nrm[i] = normal[i];
    return true;
}

// ---------------------------------------------------------------- bending / stable Neo-Hookean (README TODOs of the reference) ------
BendEnergyTerm::BendEnergyTerm(const Vec4i &hinge_, const std::vector<Vec3> &verts, double k_bend) : hinge(hinge_) {
    if (verts.size() != 4) throw std::runtime_error("**BendEnergyTerm Error: a hinge has four vertices");
    double vv[12];
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 3; ++j) vv[3 * c + j] = verts[c][j];
    const int32_t tris[6] = {0, 1, 2, 1, 0, 3};      // the two triangles of the hinge: shared edge (0, 1), opposite vertices 2 and 3
    int32_t idx[4];
    if (admm_host_bend_hinges(4, 2, tris, vv, 1, idx, coef, &area) != 1 || !(area > 0.0))
        throw std::runtime_error("**BendEnergyTerm Error: degenerate hinge");
    stiffness = k_bend * 3.0 / area; weight = std::sqrt(stiffness);
}
BendEnergyTerm::BendEnergyTerm(const Vec4i &hinge_, const double *coef4, double rest_area, double k_bend) : hinge(hinge_), area(rest_area) {
    for (int k = 0; k < 4; ++k) coef[k] = coef4[k];
    if (!(area > 0.0)) throw std::runtime_error("**BendEnergyTerm Error: degenerate hinge");
    stiffness = k_bend * 3.0 / area; weight = std::sqrt(stiffness);
}
bool BendEnergyTerm::flatten(FlatTerm &o) const {
    o.type = FlatTerm::BEND;
    for (int i = 0; i < 4; ++i) { o.idx[i] = hinge[i]; o.mat[i] = coef[i]; }
    o.weight = weight; o.k = stiffness;
    return true;
}
int create_bends_from_mesh_d(std::vector<std::shared_ptr<EnergyTerm> > &energyterms, const double *verts, int n_verts, const int *inds, int n_tris,
                             double k_bend, int vertex_offset) {
    std::vector<int32_t> tris(inds, inds + 3 * (size_t)n_tris);
    const int32_t n = admm_host_bend_hinges(n_verts, n_tris, tris.data(), verts, 0, nullptr, nullptr, nullptr);
    if (n < 0) throw std::runtime_error("create_bends_from_mesh: triangle index out of range");
    std::vector<int32_t> idx(4 * (size_t)n); std::vector<double> coef(4 * (size_t)n), area(n);
    if (n) admm_host_bend_hinges(n_verts, n_tris, tris.data(), verts, n, idx.data(), coef.data(), area.data());
    for (int32_t h = 0; h < n; ++h)
        energyterms.emplace_back(std::make_shared<BendEnergyTerm>(Vec4i(idx[4 * h] + vertex_offset, idx[4 * h + 1] + vertex_offset, idx[4 * h + 2] + vertex_offset, idx[4 * h + 3] + vertex_offset),
                                                                  &coef[4 * (size_t)h], area[h], k_bend));
    return n;
}
double StableNeoHookeanTet::energy(const VecX &F) {      // Smith et al. 2018, Eq. 14 with the Lame re-parametrisation, times the volume
    double U[9], S[3], V[9];
    host_signed_svd3(F.data(), U, S, V);
    const double mus = (4.0 / 3.0) * lame.mu, las = lame.lambda + (5.0 / 6.0) * lame.mu, al = 1.0 + 0.75 * mus / las;
    const double IC = S[0] * S[0] + S[1] * S[1] + S[2] * S[2], J = S[0] * S[1] * S[2];
    return (0.5 * mus * (IC - 3.0) + 0.5 * las * (J - al) * (J - al) - 0.5 * mus * std::log(IC + 1.0)) * volume;
}

// ---------------------------------------------------------------- LinearSolver ----------------------
// The reference reads a solver object's public tuning members on EVERY solve (src/NodalMultiColorGS.hpp:40-46,100; src/UzawaCG.hpp:44-45,92),
// so callers may cast Solver::m_linsolver and change them after initialize(): whatever the members hold now goes to the context
// before the next solve (the library does nothing when the values are the ones in effect).
void LinearSolver::push_params() {
    if (!ctx_) return;
    admm_hip_ctx *c = (admm_hip_ctx *)ctx_;
    if (auto *s1 = dynamic_cast<NodalMultiColorGS *>(this))
        check(admm_hip_set_solver_params(c, ADMM_LS_NCMCGS, s1->max_iters, s1->m_tol, s1->m_omega), "NodalMultiColorGS (tuning members)");
    else if (auto *s2 = dynamic_cast<UzawaCG *>(this)) {
        check(admm_hip_set_solver_params(c, ADMM_LS_UZAWACG, s2->max_iters, s2->m_tol, 0.0), "UzawaCG (tuning members)");
        check(admm_hip_set_solver_params(c, ADMM_LS_LDLT_AS_PCG, s2->pcg_max_iters, s2->pcg_tol, 0.0), "UzawaCG (tuning members)");
    } else if (auto *s0 = dynamic_cast<LDLTSolver *>(this))
        check(admm_hip_set_solver_params(c, ADMM_LS_LDLT_AS_PCG, s0->pcg_max_iters, s0->pcg_tol, 0.0), "LDLTSolver (tuning members)");
}

int LinearSolver::solve(VecX &x, const VecX &b) {
    if (!ctx_) throw std::runtime_error("LinearSolver::solve: not attached to an initialized Solver");
    push_params();
    int32_t it = 0;
    check(admm_hip_global_solve((admm_hip_ctx *)ctx_, b.data(), x.data(), &it), "LinearSolver::solve");
    return it;
}

// ---------------------------------------------------------------- Solver ----------------------------
Solver::Solver() : device(0), obstacle_grid_nodes(64), obstacle_grid_lo(0, 0, 0), obstacle_grid_hi(0, 0, 0), build_global_matrices(true), m_ctx(nullptr), initialized(false), m_constraints(std::make_shared<ConstraintSet>()) {}
Solver::~Solver() { release(); }
void Solver::release() { if (m_ctx) { admm_hip_destroy((admm_hip_ctx *)m_ctx); m_ctx = nullptr; } }

void Solver::set_pins(const std::vector<int> &inds, const std::vector<Vec3> &points) { // src/Solver.cpp:113-157
    const int n_pins = (int)inds.size();
    const bool pin_in_place = (int)points.size() != n_pins;
    if ((m_x.rows() == 0 && pin_in_place) || (pin_in_place && points.size() > 0))
        throw std::runtime_error("**Solver::set_pins Error: Bad input.");
    m_constraints->pins.clear();
    for (int i = 0; i < n_pins; ++i)
        m_constraints->pins[inds[i]] = pin_in_place ? Vec3(m_x.segment<3>(inds[i] * 3)) : points[i];
    if (!initialized) return;
    std::vector<int32_t> v; std::vector<double> p;
    if (m_settings.linsolver == 0 || m_settings.linsolver == 2) {
        for (auto &pe : m_pin_energies) pe.second->set_active(false);
        for (int i = 0; i < n_pins; ++i) {
            auto it = m_pin_energies.find(inds[i]);
            if (it == m_pin_energies.end()) {
                std::stringstream err;
                err << "**Solver::set_pins Error: Constraint for " << inds[i] << " not found.\n";
                throw std::runtime_error(err.str());
            }
            it->second->set_active(true);
            it->second->set_pin(m_constraints->pins[inds[i]]);
        }
    }
    push_pins();
}

void Solver::push_pins() {      // the context's pin list = ordinary pins + slide constraints (a pin it is not given again is deactivated)
    std::vector<int32_t> v; std::vector<double> p;
    for (auto &kv : m_constraints->pins) { v.push_back(kv.first); for (int j = 0; j < 3; ++j) p.push_back(kv.second[j]); }
    for (auto &kv : m_constraints->slides) {
        if (m_constraints->pins.count(kv.first)) continue;
        v.push_back(kv.first); for (int j = 0; j < 3; ++j) p.push_back(kv.second.first[j]);
    }
    check(admm_hip_set_pins((admm_hip_ctx *)m_ctx, (int32_t)v.size(), v.data(), p.data()), "Solver::set_pins");
}

void Solver::set_slide_pins(const std::vector<int> &inds, const std::vector<Vec3> &points, const std::vector<Vec3> &normals) {
    if (points.size() != inds.size() || normals.size() != inds.size()) throw std::runtime_error("**Solver::set_slide_pins Error: Bad input.");
    std::unordered_map<int, std::pair<Vec3, Vec3> > fresh;
    for (size_t i = 0; i < inds.size(); ++i) {
        const double l = normals[i].norm();
        if (!(l > 0.0)) throw std::runtime_error("**Solver::set_slide_pins Error: zero normal");
        fresh[inds[i]] = std::make_pair(points[i], normals[i] * (1.0 / l));
    }
    if (initialized && (m_settings.linsolver == 0 || m_settings.linsolver == 2))
        for (auto &kv : fresh)
            if (!m_slide_energies.count(kv.first)) {
                std::stringstream err;
                err << "**Solver::set_pins Error: Constraint for " << kv.first << " not found.\n";
                throw std::runtime_error(err.str());
            }
    std::vector<int32_t> left;      // a vertex that leaves the set must not keep its normal (it would slide again when pinned later)
    for (auto &kv : m_constraints->slides) if (!fresh.count(kv.first)) left.push_back(kv.first);
    m_constraints->slides = fresh;
    if (!initialized) return;
    if (!left.empty()) {
        std::vector<double> zero(3 * left.size(), 0.0);
        check(admm_hip_set_pin_normals((admm_hip_ctx *)m_ctx, (int32_t)left.size(), left.data(), zero.data()), "Solver::set_slide_pins");
    }
    for (auto &se : m_slide_energies) se.second->set_active(false);
    for (auto &kv : fresh) {
        auto it = m_slide_energies.find(kv.first);
        if (it == m_slide_energies.end()) continue;
        it->second->set_active(true); it->second->set_pin(kv.second.first); it->second->set_normal(kv.second.second);
    }
    push_pins();
    std::vector<int32_t> v; std::vector<double> n;
    for (auto &kv : fresh) { v.push_back(kv.first); for (int j = 0; j < 3; ++j) n.push_back(kv.second.second[j]); }
    check(admm_hip_set_pin_normals((admm_hip_ctx *)m_ctx, (int32_t)v.size(), v.data(), n.data()), "Solver::set_slide_pins");
}

void Solver::add_obstacle(std::shared_ptr<PassiveCollision> obj) { m_constraints->collider->add_passive_obj(obj); }
void Solver::add_dynamic_collider(std::shared_ptr<DynamicCollision> obj) { m_constraints->collider->add_dynamic_obj(obj); }

bool Solver::initialize(const Settings &settings_) { // src/Solver.cpp:167-261
    m_settings = settings_;
    const int dof = m_x.rows();
    if (m_settings.verbose > 0) std::cout << "Solver::initialize: " << std::endl;
    if (m_settings.timestep_s <= 0.0) {
        std::cerr << "\n**Solver Error: timestep set to " << m_settings.timestep_s << "s, changing to 1/24s." << std::endl;
        m_settings.timestep_s = 1.0 / 24.0;
    }
    if (!(m_masses.rows() == dof && dof >= 3)) {
        std::cerr << "\n**Solver Error: Problem with node data!" << std::endl;
        return false;
    }
    if (m_v.rows() != dof) m_v.resize(dof);
    m_v.setZero();
    release();

    // energy-based hard constraints (src/Solver.cpp:190-196)
    if (m_settings.linsolver == 0 || m_settings.linsolver == 2)
        for (auto &pin : m_constraints->pins) {
            m_pin_energies[pin.first] = std::make_shared<SpringPin>(pin.first, pin.second);
            energyterms.emplace_back(m_pin_energies[pin.first]);
        }
    if (m_settings.linsolver == 0 || m_settings.linsolver == 2)      // slide constraints: SlidePin terms behind the pins
        for (auto &sl : m_constraints->slides) {
            if (m_constraints->pins.count(sl.first)) continue;           // (a pinned node cannot slide as well)
            m_slide_energies[sl.first] = std::make_shared<SlidePin>(sl.first, sl.second.first, sl.second.second);
            energyterms.emplace_back(m_slide_energies[sl.first]);
        }
    // the reference sizes D here (and assigns every term its first row); we keep that side effect
    std::vector<Triplet> triplets; std::vector<double> weights;
    Flat flat;
    for (auto &term : energyterms) {
        term->get_reduction(triplets, weights);
        FlatTerm ft;
        std::memset(&ft, 0, sizeof(ft));
        if (!term->flatten(ft))
            throw std::runtime_error("Solver::initialize: an EnergyTerm subclass without a GPU kernel was added "
                                     "(the MI355X hot path has no CPU fallback)");
        flat.add(ft);
    }
    if (build_global_matrices) {      // src/Solver.cpp:204-226: D from the triplets, W, dt^2 D^T W^T W
        const int n_row = (int)weights.size();
        m_W_diag.resize(n_row);
        for (int i = 0; i < n_row; ++i) m_W_diag[i] = weights[i];
        m_D.resize(n_row, dof);
        m_D.setFromTriplets(triplets.begin(), triplets.end());
        m_Dt = m_D.transpose();
        VecX w2(n_row);
        for (int i = 0; i < n_row; ++i) w2[i] = weights[i] * weights[i];
        solver_Dt_Wt_W = m_Dt;
        la::scale_columns(solver_Dt_Wt_W, w2, m_settings.timestep_s * m_settings.timestep_s);
    }
    switch (m_settings.linsolver) {
        default: if (!std::dynamic_pointer_cast<LDLTSolver>(m_linsolver)) m_linsolver = std::make_shared<LDLTSolver>(); break;
        case 1: if (!std::dynamic_pointer_cast<NodalMultiColorGS>(m_linsolver)) m_linsolver = std::make_shared<NodalMultiColorGS>(m_constraints); break;
        case 2: if (!std::dynamic_pointer_cast<UzawaCG>(m_linsolver)) m_linsolver = std::make_shared<UzawaCG>(m_constraints); break;
    }
    if (m_settings.linsolver == 0 && (m_constraints->collider->passive_objs.size() > 0 || m_constraints->collider->dynamic_objs.size() > 0))
        throw std::runtime_error("**Solver::add_obstacle Error: No collisions with LDLT solver"); // :249-254
    std::vector<DynamicCollision::DynFlat> dyn_flat;
    for (auto &obj : m_constraints->collider->dynamic_objs) {
        DynamicCollision::DynFlat f;
        if (!obj->flatten(f)) throw std::runtime_error("Solver::initialize: only TetMeshCollision dynamic colliders have GPU kernels");
        dyn_flat.push_back(f);
    }

    admm_hip_desc d;
    std::memset(&d, 0, sizeof(d));
    d.struct_size = sizeof(d); d.device = device;
    d.n_verts = dof / 3; d.masses = m_masses.data(); d.dt = m_settings.timestep_s;
    d.vert_xyz = m_x.data();      // smooth coordinates for the coarse space of the on-chip PCG (the solve does not depend on them)
    flat.tabulate();              // user-defined xu::Spline objects -> device tables
    flat.fill(d);
    // pins that are not energy terms (linsolver 1) go through the in-sweep pin list
    std::vector<int32_t> gs_v; std::vector<double> gs_p, gs_n;
    if (m_settings.linsolver == 1) {
        for (auto &kv : m_constraints->pins) { gs_v.push_back(kv.first); for (int j = 0; j < 3; ++j) { gs_p.push_back(kv.second[j]); gs_n.push_back(0.0); } }
        for (auto &kv : m_constraints->slides) {
            if (m_constraints->pins.count(kv.first)) continue;
            gs_v.push_back(kv.first); for (int j = 0; j < 3; ++j) { gs_p.push_back(kv.second.first[j]); gs_n.push_back(kv.second.second[j]); }
        }
        d.n_pins = (int32_t)gs_v.size(); d.pin_vert = gs_v.data(); d.pin_xyz = gs_p.data(); d.pin_active = nullptr;
        d.pin_normal = m_constraints->slides.empty() ? nullptr : gs_n.data();
    }
    d.linsolver = m_settings.linsolver; d.constraint_w = m_settings.constraint_w;
    d.gs_tol = -1.0;
    if (auto s0 = std::dynamic_pointer_cast<LDLTSolver>(m_linsolver)) { d.pcg_max_iters = s0->pcg_max_iters; d.pcg_tol = s0->pcg_tol; }
    if (auto s1 = std::dynamic_pointer_cast<NodalMultiColorGS>(m_linsolver)) { d.gs_max_iters = s1->max_iters; d.gs_tol = s1->m_tol; d.gs_omega = s1->m_omega; }
    if (auto s2 = std::dynamic_pointer_cast<UzawaCG>(m_linsolver)) { d.uzawa_max_iters = s2->max_iters; d.uzawa_tol = s2->m_tol; d.pcg_max_iters = s2->pcg_max_iters; d.pcg_tol = s2->pcg_tol; }
    std::vector<int32_t> okind; std::vector<double> opar, gmeta, gdata;
    for (auto &obj : m_constraints->collider->passive_objs) {
        int k; double q[4];
        if (!obj->flatten(k, q)) {
            // a user-defined PassiveCollision (src/Collider.hpp:66-83): sampled from its own signed_distance, interpolated on the device
            Vec3 lo = obstacle_grid_lo, hi = obstacle_grid_hi;
            if (!(hi[0] > lo[0] && hi[1] > lo[1] && hi[2] > lo[2])) {
                lo = hi = Vec3(m_x.segment<3>(0));
                for (int v = 0; v < dof / 3; ++v) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], m_x[3 * v + a]); hi[a] = std::max(hi[a], m_x[3 * v + a]); }
                const double grow = 0.5 * (hi - lo).norm() + 1e-9;
                for (int a = 0; a < 3; ++a) { lo[a] -= grow; hi[a] += grow; }
            }
            const int32_t dims[3] = {obstacle_grid_nodes, obstacle_grid_nodes, obstacle_grid_nodes};
            const size_t nodes = (size_t)dims[0] * dims[1] * dims[2], first = gdata.size() / 4;
            double meta[10];
            gdata.resize(gdata.size() + 4 * nodes);
            PassiveCollision *raw = obj.get();
            auto eval = [](void *user, const double *x, double *out) {
                PassiveCollision::Payload p(0);
                static_cast<PassiveCollision *>(user)->signed_distance(Vec3(x[0], x[1], x[2]), p);
                out[0] = p.dx; for (int a = 0; a < 3; ++a) { out[1 + a] = p.point[a]; out[4 + a] = p.normal[a]; }
            };
            check(admm_host_sample_obstacle(eval, raw, lo.data(), hi.data(), dims, meta, gdata.data() + 4 * first), "Solver::initialize (sampling a user-defined obstacle)");
            meta[9] = (double)first;
            k = 3; q[0] = (double)(gmeta.size() / 10); q[1] = q[2] = q[3] = 0.0;
            gmeta.insert(gmeta.end(), meta, meta + 10);
        }
        okind.push_back(k); for (int i = 0; i < 4; ++i) opar.push_back(q[i]);
    }
    d.n_obstacles = (int32_t)okind.size(); d.obstacle_kind = okind.data(); d.obstacle_params = opar.data();
    d.n_obstacle_grids = (int32_t)(gmeta.size() / 10); d.obstacle_grid_meta = gmeta.data(); d.obstacle_grid_data = gdata.data();

    admm_hip_ctx *ctx = nullptr;
    check(admm_hip_create(&d, &ctx), "Solver::initialize");
    m_ctx = ctx;
    if (!surface_inds.empty())   // Collider::detect only looks at these vertices (src/Collider.hpp:157,163)
        check(admm_hip_set_surface_inds(ctx, (int32_t)surface_inds.size(), surface_inds.data()), "Solver::initialize");
    for (const DynamicCollision::DynFlat &f : dyn_flat)
        check(admm_hip_add_dynamic_tetmesh(ctx, f.vert_offset, (int32_t)(f.rest.size() / 3), f.rest.data(), (int32_t)(f.tets.size() / 4),
                                           f.tets.data(), (int32_t)(f.faces.size() / 3), f.faces.data()), "Solver::initialize");
    m_linsolver->attach(ctx);
    if (m_settings.soft_modes > 0 && m_settings.linsolver != 1)      // (not in any step: inverse subspace iteration with the context's own solver)
        check(admm_hip_compute_soft_modes(ctx, (int32_t)m_settings.soft_modes, 0), "Solver::initialize (soft modes)");
    // keep the assembled matrix host-side (save_matrix, LinearSolver::matrix)
    int32_t nnz = 0;
    check(admm_hip_get_matrix(ctx, nullptr, nullptr, nullptr, &nnz), "Solver::initialize");
    std::vector<int32_t> rp(d.n_verts + 1), ci(nnz); std::vector<double> va(nnz);
    check(admm_hip_get_matrix(ctx, rp.data(), ci.data(), va.data(), &nnz), "Solver::initialize");
    la::set_csr(solver_termA, d.n_verts, std::vector<int>(rp.begin(), rp.end()), std::vector<int>(ci.begin(), ci.end()), va);
    m_linsolver->update_system(solver_termA);
    if (m_settings.verbose >= 1) printf("%d nodes, %d energy terms\n", (int)m_x.size() / 3, (int)energyterms.size());
    initialized = true;
    return true;
}

void Solver::step() { // src/Solver.cpp:35-110
    if (!initialized) throw std::runtime_error("Solver::step: initialize() first");
    if (m_settings.verbose > 0) std::cout << "\nSimulating with dt: " << m_settings.timestep_s << "s..." << std::flush;
    const double dt = m_settings.timestep_s;
    for (auto &f : ext_forces) f->project(dt, m_x, m_v, m_masses); // :54 (host, pre-loop)
    admm_hip_ctx *ctx = (admm_hip_ctx *)m_ctx;
    m_linsolver->push_params();      // (tuning members changed since the last step, src/NodalMultiColorGS.hpp:40-46, src/UzawaCG.hpp:44-45)
    check(admm_hip_set_state(ctx, m_x.data(), m_v.data()), "Solver::step");
    admm_hip_stats st;
    check(admm_hip_step(ctx, m_settings.admm_iters, m_settings.gravity, &st), "Solver::step");
    check(admm_hip_get_state(ctx, m_x.data(), m_v.data()), "Solver::step");
    m_runtime = RuntimeData();
    m_runtime.global_ms = st.global_ms; m_runtime.local_ms = st.local_ms; m_runtime.collision_ms = st.collision_ms;
    m_runtime.inner_iters = st.inner_iters;
    if (m_settings.verbose > 0) m_runtime.print(m_settings);
}

void Solver::save_matrix(const std::string &filename) { // src/Solver.cpp:264-269 (Ahat; A = diag(m) + Ahat (x) I3)
    std::cout << "Saving matrix (" << solver_termA.rows() << "x" << solver_termA.cols() << ") to " << filename << std::endl;
    std::ofstream out(filename.c_str());
    std::vector<int> rp, ci; std::vector<double> va;
    la::get_csr(solver_termA, rp, ci, va);
    for (int i = 0; i < (int)solver_termA.rows(); ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            out << i << " " << ci[k] << " " << va[k] << "\n";
}

// The command-line switches of the reference's samples (src/Solver.cpp:273-307), as one table: switch, the field it sets, help text.
namespace {
struct Switch { const char *flag; double Solver::Settings::*dfield; int Solver::Settings::*ifield; const char *text; };
const Switch kSwitches[] = {
    {"-dt", &Solver::Settings::timestep_s, nullptr, "time step (s)"},
    {"-v", nullptr, &Solver::Settings::verbose, "verbosity (higher -> show more)"},
    {"-it", nullptr, &Solver::Settings::admm_iters, "# admm iters"},
    {"-g", &Solver::Settings::gravity, nullptr, "gravity (m/s^2)"},
    {"-ls", nullptr, &Solver::Settings::linsolver, "linear solver (0=LDLT as GPU PCG, 1=NCMCGS, 2=UzawaCG) "},
    {"-ck", &Solver::Settings::constraint_w, nullptr, "constraint weights (-1 = auto) "},
    {"-sm", nullptr, &Solver::Settings::soft_modes, "soft modes of the PCG's end projection (GPU build; 0 = off) "},
};
bool wants_help(const char *a) { const std::string s(a); return s == "-help" || s == "--help" || s == "-h"; }
} // namespace

bool Solver::Settings::parse_args(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        if (wants_help(argv[i])) { help(); return true; }
        if (i + 1 >= argc) break;                     // a switch needs a value behind it
        for (const Switch &sw : kSwitches) {
            if (std::string(argv[i]) != sw.flag) continue;
            std::stringstream val(argv[i + 1]);
            if (sw.dfield) val >> this->*sw.dfield; else val >> this->*sw.ifield;
        }
    }
    return false;
}

void Solver::Settings::help() {
    printf("\n==========================================\nArgs:\n");
    for (const Switch &sw : kSwitches) printf("\t%s: %s\n", sw.flag, sw.text);
    printf("==========================================\n");
}

void Solver::RuntimeData::print(const Solver::Settings &settings) { // src/Solver.cpp:309-319
    const double n = double(settings.admm_iters);
    std::cout << "\nTotal global step: " << global_ms << "ms\nTotal local step: " << local_ms << "ms\nTotal collision update: " << collision_ms
              << "ms\nAvg global step: " << global_ms / n << "ms\nAvg local step: " << local_ms / n << "ms\nAvg collision update: "
              << collision_ms / n << "ms\nADMM Iters: " << settings.admm_iters << "\nAvg Inner Iters: " << float(inner_iters) / float(settings.admm_iters)
              << std::endl;
}

} // namespace admm
