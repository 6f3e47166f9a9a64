/*
 * admm_oracle.c -- CPU restatement of the ADMM hot path of mattoverby/admm-elastic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (admm-elastic_amd/, include/)
 * may link, import or execute this file.  It is used by tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose
 * arithmetic it restates.  Nothing here is copied from the reference; the SVD is a
 * one-sided (Hestenes) Jacobi instead of Eigen's two-sided Jacobi (any convergent SVD is
 * equivalent: prox outputs depend on U,V only through U f(S) V^T).
 *
 * PINNING STATUS (see oracle/README.md):
 *   - linear tet + exact solve + step loop : pinned by the reference's own known answers
 *     (samples/tests/test_lineartet.cpp), restated in tests/test_known_answers.py
 *   - signed SVD, triangle term, pin term, EnergyTerm::update, ConstraintSet::make_matrix:
 *     pinned against the real reference sources compiled into oracle/_ref (oracle/ref_driver.cpp)
 *   - Neo-Hookean / StVK / spline prox: objective, gradient and stop rule are the reference's
 *     (src/TetEnergyTerm.cpp:173-265); the minimiser (mcl::optlib::LBFGS, mattoverby/mcloptlib,
 *     un-vendored submodule, no pinned SHA recoverable) is ABSENT => "parity unpinned" at the
 *     iterate level.  Restated here from the published L-BFGS algorithm (Nocedal 1980 two-loop
 *     recursion + backtracking Armijo line search).
 *   - NodalMultiColorGS: sweeps restated; the colouring library (mclscene) is ABSENT =>
 *     "parity unpinned" for the colour order; colours are an input here.
 *
 *   - TetMeshCollision (dynamic self-collision): the BVH traversals are mclscene's (ABSENT) => "parity
 *     unpinned"; restated by brute force with index tie rules (see orc_detect_dynamic).
 *
 * Layout conventions: 3x3 matrices are column-major (Eigen default), index c*3+r.
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#include <stdint.h>

#define M3(A, r, c) ((A)[(c) * 3 + (r)])

static double det3(const double *A) {
    return M3(A,0,0) * (M3(A,1,1) * M3(A,2,2) - M3(A,1,2) * M3(A,2,1))
         - M3(A,0,1) * (M3(A,1,0) * M3(A,2,2) - M3(A,1,2) * M3(A,2,0))
         + M3(A,0,2) * (M3(A,1,0) * M3(A,2,1) - M3(A,1,1) * M3(A,2,0));
}

static void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double norm3(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* any unit vector orthogonal to unit vector a */
static void any_orthogonal(const double *a, double *o) {
    double e[3] = {0, 0, 0};
    int k = 0;
    if (fabs(a[1]) < fabs(a[k])) k = 1;
    if (fabs(a[2]) < fabs(a[k])) k = 2;
    e[k] = 1.0;
    cross3(a, e, o);
    double n = norm3(o);
    o[0] /= n; o[1] /= n; o[2] /= n;
}

/*
 * One-sided Jacobi SVD of a 3 x n column-major matrix (n = 2 or 3).
 * Out: U 3x3 (full, orthonormal), S[n] sorted descending >= 0, V n x n (col-major, ld n).
 * Stands in for Eigen::JacobiSVD<..>(F, ComputeFullU|ComputeFullV) as used at
 * src/FastSVD.hpp:47, src/TetEnergyTerm.cpp:76, src/TriEnergyTerm.cpp:78.
 */
void orc_svd3xn(int n, const double *A, double *U, double *S, double *V) {
    double G[9], Vw[9];
    memcpy(G, A, sizeof(double) * 3 * n);
    for (int i = 0; i < n * n; ++i) Vw[i] = 0.0;
    for (int i = 0; i < n; ++i) Vw[i * n + i] = 1.0;

    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double *gp = G + 3 * p, *gq = G + 3 * q;
                double al = gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
                double be = gq[0] * gq[0] + gq[1] * gq[1] + gq[2] * gq[2];
                double ga = gp[0] * gq[0] + gp[1] * gq[1] + gp[2] * gq[2];
                if (fabs(ga) <= 1e-17 * sqrt(al * be) || ga == 0.0) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    double a = gp[r], b = gq[r];
                    gp[r] = c * a - s * b;
                    gq[r] = s * a + c * b;
                }
                for (int r = 0; r < n; ++r) {
                    double a = Vw[p * n + r], b = Vw[q * n + r];
                    Vw[p * n + r] = c * a - s * b;
                    Vw[q * n + r] = s * a + c * b;
                }
            }
        }
        if (!rotated) break;
    }
    /* singular values = column norms; sort descending (Eigen sorts too, JacobiSVD.h step 4) */
    double sig[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < n; ++j) sig[j] = norm3(G + 3 * j);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (sig[ord[j]] > sig[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double smax = sig[ord[0]];
    int rank = 0, open_rank = 1;
    for (int j = 0; j < n; ++j) {
        int o = ord[j];
        S[j] = sig[o];
        for (int r = 0; r < n; ++r) V[j * n + r] = Vw[o * n + r];
        if (open_rank && sig[o] > 1e-300 && sig[o] > 1e-15 * smax) {
            for (int r = 0; r < 3; ++r) U[3 * j + r] = G[3 * o + r] / sig[o];
            rank = j + 1;
        } else {
            open_rank = 0; /* remaining U columns are completed below */
        }
    }
    /* complete U to a full orthonormal basis */
    if (rank == 0) { memset(U, 0, sizeof(double) * 9); U[0] = U[4] = U[8] = 1.0; }
    else if (rank == 1) { any_orthogonal(U, U + 3); cross3(U, U + 3, U + 6); }
    else if (rank == 2) { cross3(U, U + 3, U + 6); }
}

/* src/FastSVD.hpp:43-68 -- SVD with U,V in SO(3); S[2] carries the sign of det F. */
void orc_signed_svd3(const double *F, double *S, double *U, double *V) {
    orc_svd3xn(3, F, U, S, V);
    if (det3(U) < 0.0) { /* FastSVD.hpp:55-58 */
        for (int r = 0; r < 3; ++r) M3(U, r, 2) = -M3(U, r, 2);
        S[2] = -S[2];
    }
    if (det3(V) < 0.0) { /* FastSVD.hpp:61-66 */
        for (int r = 0; r < 3; ++r) M3(V, r, 2) = -M3(V, r, 2);
        S[2] = -S[2];
    }
}

static void usvt(const double *U, const double *S, const double *V, double *out) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += M3(U, r, k) * S[k] * M3(V, c, k);
            M3(out, r, c) = a;
        }
}

/* src/TetEnergyTerm.cpp:73-92 -- linear ("ARAP-like") tet prox: z <- (P + z)/2 */
void orc_prox_tet_linear(double *z) {
    double U[9], S[3], V[9], P[9];
    orc_svd3xn(3, z, U, S, V);
    double one[3] = {1.0, 1.0, 1.0};
    if (det3(z) < 0.0) one[2] = -1.0; /* :80 */
    usvt(U, one, V, P);               /* :82 */
    for (int i = 0; i < 9; ++i) z[i] = 0.5 * (P[i] + z[i]); /* :86 */
}

/* src/TetEnergyTerm.cpp:94-100 -- linear tet energy = k/2 vol |sigma - 1|^2 */
double orc_energy_tet_linear(const double *F, double k, double vol) {
    double U[9], S[3], V[9];
    orc_svd3xn(3, F, U, S, V);
    double e = 0.0;
    for (int i = 0; i < 3; ++i) e += (S[i] - 1.0) * (S[i] - 1.0);
    return 0.5 * k * vol * e;
}

/* ---- principal-stretch objectives: kind 1 = NeoHookean, 2 = StVK, 3 / 4 / 5 = SplineTet with xu::NeoHookean (the default) /
 * xu::StVK / xu::CoRotated with their compression term kappa (src/XuSpline.hpp:43-45; 0 as the reference constructs them) ---- */
typedef struct { int kind; double mu, lambda, k, kappa; double x0[3]; } prox_problem;

#define ORC_FLT_MAX 3.40282346638528859812e+38 /* std::numeric_limits<float>::max() */

/* Eq. 16 compression term of the xu:: splines and its derivative, exactly as src/XuSpline.hpp:44-45 writes them (the
 * second derivative is derived: it is only used by the "tight" Newton polish) */
static double xu_compress(double kappa, double x) { double t = (1.0 - x) / 6.0; return (kappa / 12.0) * t * t * t; }
static double xu_d_compress(double kappa, double x) { double t = (1.0 - x) / 6.0; return (-kappa / 24.0) * t * t; }
static double xu_dd_compress(double kappa, double x) { return (kappa / 72.0) * ((1.0 - x) / 6.0); }

/* Xu spline NeoHookean (src/XuSpline.hpp:48-62); kappa = 0 as constructed at TetEnergyTerm.hpp:194 */
static double xu_nh_f(double mu, double x) { return 0.5 * mu * (x * x - 1.0); }
static double xu_nh_h(double mu, double la, double x) { double l = log(x); return -mu * l + 0.5 * la * l * l; }
static double xu_nh_df(double mu, double x) { return mu * x; }
static double xu_nh_dh(double mu, double la, double x) { return -mu / x + la * log(x) / x; }

/* Xu spline StVK (src/XuSpline.hpp:64-82) and CoRotated (:84-96), kappa = 0 (no compression term):
 * Psi = sum f(x_i) + sum g(x_i x_j) + h(x_0 x_1 x_2)  (TetEnergyTerm.cpp:243-247) */
static double xu_f(int kind, double mu, double la, double x) {
    if (kind == 3) return xu_nh_f(mu, x);
    if (kind == 4) { double x2 = x * x; return 0.125 * la * (x2 * x2 - 6.0 * x2 + 5.0) + 0.25 * mu * (x2 - 1.0) * (x2 - 1.0); }
    return 0.5 * la * (x * x - 6.0 * x + 5.0) + mu * (x - 1.0) * (x - 1.0);
}
static double xu_g(int kind, double la, double x) {
    if (kind == 3) return 0.0;
    if (kind == 4) return 0.25 * la * (x * x - 1.0);
    return la * (x - 1.0);
}
static double xu_h(int kind, double mu, double la, double kappa, double x) { return (kind == 3 ? xu_nh_h(mu, la, x) : 0.0) + xu_compress(kappa, x); }
static double xu_df(int kind, double mu, double la, double x) {
    if (kind == 3) return xu_nh_df(mu, x);
    if (kind == 4) { double x2 = x * x; return 0.125 * la * (4.0 * x2 * x - 12.0 * x) + mu * x * (x2 - 1.0); }
    return 0.5 * la * (2.0 * x - 6.0) + 2.0 * mu * (x - 1.0);
}
static double xu_dg(int kind, double la, double x) {
    if (kind == 3) return 0.0;
    if (kind == 4) return 0.5 * la * x;
    return la;
}
static double xu_dh(int kind, double mu, double la, double kappa, double x) { return (kind == 3 ? xu_nh_dh(mu, la, x) : 0.0) + xu_d_compress(kappa, x); }

/* STABLE NEO-HOOKEAN, kind 7 -- no reference code ("parity unpinned"): the reference's README lists it as a TODO (README.md:23-28); it
 * would be one more HyperElasticTet beside NeoHookeanTet (src/TetEnergyTerm.hpp:116-136).  Restated from the paper: Smith, de Goes, Kim,
 * "Stable Neo-Hookean Flesh Simulation", ACM TOG 37(2), 2018, Eq. 14 with the Lame re-parametrisation of section 3.4:
 *   Psi = mu_s/2 (I_C - 3) + la_s/2 (J - alpha)^2 - mu_s/2 log(I_C + 1),  mu_s = 4/3 mu, la_s = lambda + 5/6 mu, alpha = 1 + 3 mu_s / (4 la_s),
 * I_C = sum x_i^2, J = x0 x1 x2 in the principal stretches.  Defined for negative stretches: value() has no FLT_MAX barrier. */
static void snh_constants(const prox_problem *p, double *mus, double *las, double *alpha) {
    *mus = (4.0 / 3.0) * p->mu; *las = p->lambda + (5.0 / 6.0) * p->mu; *alpha = 1.0 + 0.75 * (*mus) / (*las);
}

static double energy_density(const prox_problem *p, const double *x) {
    if (p->kind == 7) {
        double mus, las, al; snh_constants(p, &mus, &las, &al);
        const double IC = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], J = x[0] * x[1] * x[2];
        return 0.5 * mus * (IC - 3.0) + 0.5 * las * (J - al) * (J - al) - 0.5 * mus * log(IC + 1.0);
    }
    if (p->kind == 1) { /* TetEnergyTerm.cpp:173-182 */
        double J = x[0] * x[1] * x[2];
        double I1 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
        double logI3 = log(J * J);
        return 0.5 * p->mu * (I1 - logI3 - 3.0) + 0.125 * p->lambda * logI3 * logI3;
    } else if (p->kind == 2) { /* TetEnergyTerm.cpp:220-226 */
        double st[3], tr = 0.0, dd = 0.0;
        for (int i = 0; i < 3; ++i) { st[i] = 0.5 * (x[i] * x[i] - 1.0); tr += st[i]; dd += st[i] * st[i]; }
        return p->mu * dd + p->lambda * 0.5 * tr * tr;
    } else { /* TetEnergyTerm.cpp:243-247: kind 3 xu::NeoHookean (g == 0), 4 xu::StVK, 5 xu::CoRotated */
        const int kd = p->kind;
        return xu_f(kd, p->mu, p->lambda, x[0]) + xu_f(kd, p->mu, p->lambda, x[1]) + xu_f(kd, p->mu, p->lambda, x[2])
             + xu_g(kd, p->lambda, x[0] * x[1]) + xu_g(kd, p->lambda, x[1] * x[2]) + xu_g(kd, p->lambda, x[2] * x[0])
             + xu_h(kd, p->mu, p->lambda, p->kappa, x[0] * x[1] * x[2]);
    }
}

/* ::value  -- TetEnergyTerm.cpp:184-192, :210-218, :249-257 */
static double prox_value(const prox_problem *p, const double *x) {
    if (p->kind != 7 && (x[0] < 0.0 || x[1] < 0.0 || x[2] < 0.0)) return ORC_FLT_MAX;
    double q = 0.0;
    for (int i = 0; i < 3; ++i) q += (x[i] - p->x0[i]) * (x[i] - p->x0[i]);
    return energy_density(p, x) + 0.5 * p->k * q;
}

/* ::gradient -- TetEnergyTerm.cpp:195-204, :228-237, :259-265.  Returns value.  The reference throws
 * for NH when J <= 0; callers here never evaluate the gradient at an infeasible point. */
static double prox_gradient(const prox_problem *p, const double *x, double *g) {
    if (p->kind == 7) {
        double mus, las, al; snh_constants(p, &mus, &las, &al);
        const double IC = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], J = x[0] * x[1] * x[2], q = 1.0 / (IC + 1.0);
        const double dJ[3] = {x[1] * x[2], x[2] * x[0], x[0] * x[1]};
        for (int i = 0; i < 3; ++i) g[i] = mus * x[i] * (1.0 - q) + las * (J - al) * dJ[i] + p->k * (x[i] - p->x0[i]);
    } else if (p->kind == 1) {
        double J = x[0] * x[1] * x[2], lJ = log(J);
        for (int i = 0; i < 3; ++i) {
            double xi = 1.0 / x[i];
            g[i] = (p->mu * (x[i] - xi) + p->lambda * lJ * xi) + p->k * (x[i] - p->x0[i]);
        }
    } else if (p->kind == 2) {
        double xx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
        for (int i = 0; i < 3; ++i)
            g[i] = p->mu * x[i] * (x[i] * x[i] - 1.0) + 0.5 * p->lambda * (xx - 3.0) * x[i]
                 + p->k * (x[i] - p->x0[i]);
    } else { /* TetEnergyTerm.cpp:259-265 */
        const int kd = p->kind; const double mu = p->mu, la = p->lambda;
        double hp = xu_dh(kd, mu, la, p->kappa, x[0] * x[1] * x[2]);
        g[0] = xu_df(kd, mu, la, x[0]) + xu_dg(kd, la, x[0] * x[1]) * x[1] + xu_dg(kd, la, x[2] * x[0]) * x[2] + hp * x[1] * x[2] + p->k * (x[0] - p->x0[0]);
        g[1] = xu_df(kd, mu, la, x[1]) + xu_dg(kd, la, x[1] * x[2]) * x[2] + xu_dg(kd, la, x[0] * x[1]) * x[0] + hp * x[2] * x[0] + p->k * (x[1] - p->x0[1]);
        g[2] = xu_df(kd, mu, la, x[2]) + xu_dg(kd, la, x[2] * x[0]) * x[0] + xu_dg(kd, la, x[1] * x[2]) * x[1] + hp * x[0] * x[1] + p->k * (x[2] - p->x0[2]);
    }
    return prox_value(p, x);
}

/* Hessian of value() -- derived, used only for the "tight" Newton polish */
static void prox_hessian(const prox_problem *p, const double *x, double *H) {
    memset(H, 0, 9 * sizeof(double));
    if (p->kind == 7) {
        double mus, las, al; snh_constants(p, &mus, &las, &al);
        const double IC = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], J = x[0] * x[1] * x[2], q = 1.0 / (IC + 1.0);
        const double dJ[3] = {x[1] * x[2], x[2] * x[0], x[0] * x[1]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                M3(H, i, j) = 2.0 * mus * q * q * x[i] * x[j] + las * (dJ[i] * dJ[j] + (i == j ? 0.0 : (J - al) * x[3 - i - j]));
        for (int i = 0; i < 3; ++i) M3(H, i, i) += mus * (1.0 - q) + p->k;
        return;
    }
    if (p->kind == 1 || p->kind == 3) {
        double lJ = log(x[0] * x[1] * x[2]);
        double xi[3] = {1.0 / x[0], 1.0 / x[1], 1.0 / x[2]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M3(H, i, j) = p->lambda * xi[i] * xi[j];
        for (int i = 0; i < 3; ++i)
            M3(H, i, i) += p->mu * (1.0 + xi[i] * xi[i]) - p->lambda * lJ * xi[i] * xi[i] + p->k;
    } else if (p->kind == 5) { /* co-rotated: Psi = mu sum (x_i - 1)^2 + lambda/2 (sum x_i - 3)^2, quadratic */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M3(H, i, j) = p->lambda;
        for (int i = 0; i < 3; ++i) M3(H, i, i) += 2.0 * p->mu + p->k;
    } else { /* StVK, also as xu::StVK (kind 4) */
        double xx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M3(H, i, j) = p->lambda * x[i] * x[j];
        for (int i = 0; i < 3; ++i)
            M3(H, i, i) += p->mu * (3.0 * x[i] * x[i] - 1.0) + 0.5 * p->lambda * (xx - 3.0) + p->k;
    }
    if (p->kind >= 3 && p->kappa != 0.0) {   /* c(J), J = x0 x1 x2:  c'' dJ dJ^T + c' d2J  (d2J_ij = x_k off the diagonal) */
        const double J = x[0] * x[1] * x[2];
        const double c1 = xu_d_compress(p->kappa, J), c2 = xu_dd_compress(p->kappa, J);
        const double dJ[3] = {x[1] * x[2], x[2] * x[0], x[0] * x[1]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M3(H, i, j) += c2 * dJ[i] * dJ[j] + (i == j ? 0.0 : c1 * x[3 - i - j]);
    }
}

static int feasible(const prox_problem *p, const double *x) {
    if (p->kind == 7) return 1;                    /* stable Neo-Hookean: defined everywhere */
    return x[0] > 0.0 && x[1] > 0.0 && x[2] > 0.0;
}

/*
 * L-BFGS minimiser standing in for mcl::optlib::LBFGS<double,3>::minimize (ABSENT dependency;
 * call site src/TetEnergyTerm.cpp:133, interface src/TetEnergyTerm.hpp:90-97).
 * Published algorithm: Nocedal (1980) two-loop recursion, history M, backtracking Armijo line
 * search (c1 = 1e-4, halve), first step scaled by 1/|g|.  Stop rule = the reference's
 * HyperElasticTet::Prox::converged (TetEnergyTerm.hpp:93-95): |g| < 1e-6 or |x0-x1| < 1e-6.
 * Returns iterations used.
 */
#define LB_M 6
static int lbfgs3(const prox_problem *p, double *x, int max_iters) {
    double sh[LB_M][3], yh[LB_M][3], rho[LB_M];
    int hist = 0, head = 0;
    double g[3];
    double f = prox_gradient(p, x, g);
    int it = 0;
    for (; it < max_iters; ++it) {
        double d[3] = {-g[0], -g[1], -g[2]}, alpha[LB_M];
        for (int h = 0; h < hist; ++h) {
            int i = (head - 1 - h + LB_M) % LB_M;
            alpha[i] = rho[i] * (sh[i][0] * d[0] + sh[i][1] * d[1] + sh[i][2] * d[2]);
            for (int c = 0; c < 3; ++c) d[c] -= alpha[i] * yh[i][c];
        }
        if (hist > 0) {
            int i = (head - 1 + LB_M) % LB_M;
            double sy = sh[i][0] * yh[i][0] + sh[i][1] * yh[i][1] + sh[i][2] * yh[i][2];
            double yy = yh[i][0] * yh[i][0] + yh[i][1] * yh[i][1] + yh[i][2] * yh[i][2];
            double sc = sy / yy;
            for (int c = 0; c < 3; ++c) d[c] *= sc;
        }
        for (int h = hist - 1; h >= 0; --h) {
            int i = (head - 1 - h + LB_M) % LB_M;
            double b = rho[i] * (yh[i][0] * d[0] + yh[i][1] * d[1] + yh[i][2] * d[2]);
            for (int c = 0; c < 3; ++c) d[c] += (alpha[i] - b) * sh[i][c];
        }
        double gd = g[0] * d[0] + g[1] * d[1] + g[2] * d[2];
        if (!(gd < 0.0)) { /* not a descent direction: reset to steepest descent */
            for (int c = 0; c < 3; ++c) d[c] = -g[c];
            gd = -(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            hist = 0;
        }
        double gnorm = sqrt(g[0]*g[0] + g[1]*g[1] + g[2]*g[2]);
        double t = (hist == 0 && gnorm > 1.0) ? 1.0 / gnorm : 1.0;
        double xn[3], fn = 0.0;
        int ok = 0;
        for (int ls = 0; ls < 60; ++ls) {
            for (int c = 0; c < 3; ++c) xn[c] = x[c] + t * d[c];
            if (feasible(p, xn)) {
                fn = prox_value(p, xn);
                if (fn <= f + 1e-4 * t * gd) { ok = 1; break; }
            }
            t *= 0.5;
        }
        if (!ok) break;
        double gn[3];
        fn = prox_gradient(p, xn, gn);
        double s[3], y[3], sy = 0.0, gg = 0.0, ss = 0.0;
        for (int c = 0; c < 3; ++c) {
            s[c] = xn[c] - x[c]; y[c] = gn[c] - g[c];
            sy += s[c] * y[c]; gg += gn[c] * gn[c]; ss += s[c] * s[c];
        }
        if (sy > 1e-300) {
            for (int c = 0; c < 3; ++c) { sh[head][c] = s[c]; yh[head][c] = y[c]; }
            rho[head] = 1.0 / sy;
            head = (head + 1) % LB_M;
            if (hist < LB_M) hist++;
        }
        for (int c = 0; c < 3; ++c) { x[c] = xn[c]; g[c] = gn[c]; }
        f = fn;
        if (sqrt(gg) < 1e-6 || sqrt(ss) < 1e-6) { ++it; break; } /* TetEnergyTerm.hpp:93-95 */
    }
    return it;
}

/* Damped (projected) Newton polish to the exact minimiser ("tight" oracle mode).
 * StVK's feasible set is x >= 0 (value() is FLT_MAX only for x < 0); for inverted elements the
 * minimiser lies on that boundary, so components at the bound with an outward-pointing gradient are
 * frozen and trial points are projected back onto x >= 0. */
static int newton3(const prox_problem *p, double *x, int max_iters) {
    int it = 0;
    const int proj = (p->kind == 2 || p->kind == 4 || p->kind == 5);   /* no log barrier: the feasible set is x >= 0 */
    for (; it < max_iters; ++it) {
        double g[3], H[9];
        double f = prox_gradient(p, x, g);
        prox_hessian(p, x, H);
        double gr[3] = {g[0], g[1], g[2]};
        for (int i = 0; i < 3; ++i)
            if (proj && x[i] <= 0.0 && g[i] > 0.0) { /* active bound: remove from the Newton system */
                for (int j = 0; j < 3; ++j) { M3(H, i, j) = 0.0; M3(H, j, i) = 0.0; }
                M3(H, i, i) = 1.0; gr[i] = 0.0;
            }
        /* solve H d = -g by Cramer; shift the diagonal until H is positive definite */
        double d[3] = {-gr[0], -gr[1], -gr[2]};
        double shift = 0.0, tr = fabs(M3(H,0,0)) + fabs(M3(H,1,1)) + fabs(M3(H,2,2));
        for (int tries = 0; tries < 60; ++tries) {
            double Hs[9];
            memcpy(Hs, H, sizeof(Hs));
            for (int c = 0; c < 3; ++c) M3(Hs, c, c) += shift;
            double D = det3(Hs);
            int pd = M3(Hs,0,0) > 0 && (M3(Hs,0,0)*M3(Hs,1,1) - M3(Hs,0,1)*M3(Hs,1,0)) > 0 && D > 0;
            if (pd) {
                double Hc[9];
                for (int c = 0; c < 3; ++c) {
                    memcpy(Hc, Hs, sizeof(Hc));
                    for (int r = 0; r < 3; ++r) M3(Hc, r, c) = -gr[r];
                    d[c] = det3(Hc) / D;
                }
                break;
            }
            shift = (shift == 0.0) ? 1e-3 * tr + 1e-300 : 10.0 * shift;
        }
        double t = 1.0, xn[3];
        int ok = 0;
        const double fscale = (fabs(p->mu) + fabs(p->lambda) + fabs(p->k)) * (1.0 + x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        for (int ls = 0; ls < 60; ++ls) {
            double gs = 0.0;
            for (int c = 0; c < 3; ++c) {
                xn[c] = x[c] + t * d[c];
                if (proj && xn[c] < 0.0) xn[c] = 0.0;
                gs += g[c] * (xn[c] - x[c]);
            }
            int feas = proj ? 1 : feasible(p, xn);
            /* round-off in f is absolute: O(mu + lambda + k) terms cancel to O(strain^2), so the allowance is measured against
             * the size of the terms, not against |f| -- with 1e-14 |f| the search refused the last Newton steps and the polish
             * stalled ~1e-8 away from the minimiser (the GPU path, checked with this file's own gradient, was closer) */
            if (feas && prox_value(p, xn) <= f + 1e-4 * gs + 1e-15 * (fabs(f) + fscale)) { ok = 1; break; }
            t *= 0.5;
        }
        if (!ok) break;
        double ss = 0.0;
        for (int c = 0; c < 3; ++c) { ss += (xn[c] - x[c]) * (xn[c] - x[c]); x[c] = xn[c]; }
        if (sqrt(ss) < 1e-14 * (1.0 + norm3(x))) { ++it; break; }
    }
    return it;
}

/*
 * src/TetEnergyTerm.cpp:114-136 -- HyperElasticTet::prox.
 * kind 1 NH, 2 StVK, 3 / 4 / 5 Spline(NH / StVK / CoRotated).  mode 0 = reference stop rule only (L-BFGS), mode 1 = "tight"
 * (L-BFGS then Newton polish to the exact minimiser).  Returns minimiser iterations.
 */
int orc_prox_tet_hyper_k(int kind, double mu, double lambda, double k, double kappa, double *z, int mode) {
    double U[9], S[3], V[9];
    orc_signed_svd3(z, S, U, V);
    prox_problem p;
    p.kind = kind; p.mu = mu; p.lambda = lambda; p.k = k; p.kappa = kind >= 3 ? kappa : 0.0;
    p.x0[0] = S[0]; p.x0[1] = S[1]; p.x0[2] = S[2];           /* :124 set_x0 BEFORE the fix-ups */
    const double eps = 1e-6;
    if (fabs(S[0]) < eps && fabs(S[1]) < eps && fabs(S[2]) < eps) { S[0] = S[1] = S[2] = eps; } /* :128-131 */
    if (S[2] < 0.0) S[2] = -S[2];                               /* :133 */
    if (S[2] == 0.0) S[2] = 1e-12; /* reference would take log(0); keep the start strictly feasible */
    int it = lbfgs3(&p, S, 200);                                /* :135 */
    if (mode == 1) it += newton3(&p, S, 100);
    usvt(U, S, V, z);                                           /* :136-137 */
    return it;
}

int orc_prox_tet_hyper(int kind, double mu, double lambda, double k, double *z, int mode) {
    return orc_prox_tet_hyper_k(kind, mu, lambda, k, 0.0, z, mode);
}

/* SplineProx::value / ::gradient with the spline's compression term kappa (kinds 3..5; ignored for 1, 2) */
double orc_prox_value_k(int kind, double mu, double lambda, double k, double kappa, const double *x0, const double *x) {
    prox_problem p; p.kind = kind; p.mu = mu; p.lambda = lambda; p.k = k; p.kappa = kind >= 3 ? kappa : 0.0;
    memcpy(p.x0, x0, sizeof(p.x0));
    return prox_value(&p, x);
}
void orc_prox_gradient_k(int kind, double mu, double lambda, double k, double kappa, const double *x0, const double *x, double *g) {
    prox_problem p; p.kind = kind; p.mu = mu; p.lambda = lambda; p.k = k; p.kappa = kind >= 3 ? kappa : 0.0;
    memcpy(p.x0, x0, sizeof(p.x0));
    prox_gradient(&p, x, g);
}
double orc_prox_value(int kind, double mu, double lambda, double k, const double *x0, const double *x) {
    return orc_prox_value_k(kind, mu, lambda, k, 0.0, x0, x);
}
void orc_prox_gradient(int kind, double mu, double lambda, double k, const double *x0, const double *x, double *g) {
    orc_prox_gradient_k(kind, mu, lambda, k, 0.0, x0, x, g);
}

/* src/TriEnergyTerm.cpp:73-101 -- triangle prox on the 3x2 F (col-major, 6 doubles) + strain limit */
void orc_prox_tri(double *z, double limit_min, double limit_max) {
    double U[9], S[2], V[4], P[6];
    orc_svd3xn(2, z, U, S, V);
    /* P = U [I2;0] V^T  (:79-81) */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 2; ++c)
            P[c * 3 + r] = U[0 * 3 + r] * V[0 * 2 + c] + U[1 * 3 + r] * V[1 * 2 + c];
    for (int i = 0; i < 6; ++i) z[i] = 0.5 * (P[i] + z[i]); /* :83 */
    int check = limit_min > 0.0 || limit_max < 99.0;          /* :91 */
    if (check) {
        double l0 = norm3(z), l1 = norm3(z + 3);
        if (l0 < limit_min) for (int i = 0; i < 3; ++i) z[i] *= (limit_min / l0);
        if (l1 < limit_min) for (int i = 0; i < 3; ++i) z[3 + i] *= (limit_min / l1);
        if (l0 > limit_max) for (int i = 0; i < 3; ++i) z[i] *= (limit_max / l0);
        if (l1 > limit_max) for (int i = 0; i < 3; ++i) z[3 + i] *= (limit_max / l1);
    }
}

/* ---- element set-up ------------------------------------------------------------------------ */

/* src/TetEnergyTerm.cpp:31-48 -- Binv (col-major), volume; returns -1 on an inverted rest tet */
int orc_tet_rest(const double *v0, const double *v1, const double *v2, const double *v3,
                 double *Binv, double *vol) {
    double B[9];
    for (int r = 0; r < 3; ++r) { M3(B, r, 0) = v1[r] - v0[r]; M3(B, r, 1) = v2[r] - v0[r]; M3(B, r, 2) = v3[r] - v0[r]; }
    double d = det3(B);
    *vol = d / 6.0;
    if (*vol < 0) return -1;
    double id = 1.0 / d;
    M3(Binv,0,0) =  (M3(B,1,1)*M3(B,2,2) - M3(B,1,2)*M3(B,2,1)) * id;
    M3(Binv,0,1) = -(M3(B,0,1)*M3(B,2,2) - M3(B,0,2)*M3(B,2,1)) * id;
    M3(Binv,0,2) =  (M3(B,0,1)*M3(B,1,2) - M3(B,0,2)*M3(B,1,1)) * id;
    M3(Binv,1,0) = -(M3(B,1,0)*M3(B,2,2) - M3(B,1,2)*M3(B,2,0)) * id;
    M3(Binv,1,1) =  (M3(B,0,0)*M3(B,2,2) - M3(B,0,2)*M3(B,2,0)) * id;
    M3(Binv,1,2) = -(M3(B,0,0)*M3(B,1,2) - M3(B,0,2)*M3(B,1,0)) * id;
    M3(Binv,2,0) =  (M3(B,1,0)*M3(B,2,1) - M3(B,1,1)*M3(B,2,0)) * id;
    M3(Binv,2,1) = -(M3(B,0,0)*M3(B,2,1) - M3(B,0,1)*M3(B,2,0)) * id;
    M3(Binv,2,2) =  (M3(B,0,0)*M3(B,1,1) - M3(B,0,1)*M3(B,1,0)) * id;
    return 0;
}

/* src/TriEnergyTerm.cpp:29-52 -- rest_pose (2x2 col-major), area; -1 on negative area */
int orc_tri_rest(const double *v0, const double *v1, const double *v2, double *rest, double *area) {
    double e12[3], e13[3], n1[3], n2[3];
    for (int r = 0; r < 3; ++r) { e12[r] = v1[r] - v0[r]; e13[r] = v2[r] - v0[r]; }
    double l = norm3(e12);
    for (int r = 0; r < 3; ++r) n1[r] = e12[r] / l;
    double dp = e13[0] * n1[0] + e13[1] * n1[1] + e13[2] * n1[2];
    for (int r = 0; r < 3; ++r) n2[r] = e13[r] - dp * n1[r];
    l = norm3(n2);
    for (int r = 0; r < 3; ++r) n2[r] /= l;
    /* M = basis^T * edges (2x2): M(i,j) = n_i . e_j */
    double m00 = n1[0]*e12[0] + n1[1]*e12[1] + n1[2]*e12[2];
    double m01 = n1[0]*e13[0] + n1[1]*e13[1] + n1[2]*e13[2];
    double m10 = n2[0]*e12[0] + n2[1]*e12[1] + n2[2]*e12[2];
    double m11 = n2[0]*e13[0] + n2[1]*e13[1] + n2[2]*e13[2];
    double d = m00 * m11 - m01 * m10;
    *area = d / 2.0;
    if (*area < 0) return -1;
    rest[0] = m11 / d;  /* (0,0) */
    rest[1] = -m10 / d; /* (1,0) */
    rest[2] = -m01 / d; /* (0,1) */
    rest[3] = m00 / d;  /* (1,1) */
    return 0;
}

/* ---- local step (src/EnergyTerm.hpp:130-140 applied to every term, Solver.cpp:84-87) ------- */

/* tets: F = [x1-x0,x2-x0,x3-x0] * Binv, equal to D_i x with the D-block of TetEnergyTerm.cpp:50-71.
 * z,u: AoS [nt][9] in the reference's row order (row 3r+j <-> F(j,r)).  x: [nv][3]. */
void orc_local_tets_k(int nt, const int32_t *idx, const double *Binv, const int32_t *kind,
                      const double *mu, const double *lambda, const double *k, const double *kappa,
                      const double *x, double *z, double *u, int mode) {
#pragma omp parallel for schedule(static)
    for (int t = 0; t < nt; ++t) {
        const int32_t *id = idx + 4 * t;
        const double *Bi = Binv + 9 * t;
        double Ds[9], Dix[9], zi[9];
        for (int m = 0; m < 3; ++m)
            for (int j = 0; j < 3; ++j) M3(Ds, j, m) = x[3 * id[m + 1] + j] - x[3 * id[0] + j];
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 3; ++r) {
                double a = 0.0;
                for (int m = 0; m < 3; ++m) a += M3(Ds, j, m) * M3(Bi, m, r);
                M3(Dix, j, r) = a;
            }
        double *ui = u + 9 * t;
        for (int i = 0; i < 9; ++i) zi[i] = Dix[i] + ui[i];            /* EnergyTerm.hpp:135 */
        if (kind[t] == 0) orc_prox_tet_linear(zi);
        else orc_prox_tet_hyper_k(kind[t], mu[t], lambda[t], k[t], kappa ? kappa[t] : 0.0, zi, mode);
        for (int i = 0; i < 9; ++i) { ui[i] += Dix[i] - zi[i]; z[9 * t + i] = zi[i]; } /* :137-139 */
    }
}
void orc_local_tets(int nt, const int32_t *idx, const double *Binv, const int32_t *kind,
                    const double *mu, const double *lambda, const double *k,
                    const double *x, double *z, double *u, int mode) {
    orc_local_tets_k(nt, idx, Binv, kind, mu, lambda, k, (const double *)0, x, z, u, mode);
}

/* tris: rows i and 3+i <-> columns of the 3x2 F = [x1-x0,x2-x0] * rest (TriEnergyTerm.cpp:54-69) */
void orc_local_tris(int n, const int32_t *idx, const double *rest, const double *lmin, const double *lmax,
                    const double *x, double *z, double *u) {
#pragma omp parallel for schedule(static)
    for (int t = 0; t < n; ++t) {
        const int32_t *id = idx + 3 * t;
        const double *R = rest + 4 * t; /* col-major 2x2 */
        double Dix[6], zi[6];
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < 3; ++j) {
                double e1 = x[3 * id[1] + j] - x[3 * id[0] + j];
                double e2 = x[3 * id[2] + j] - x[3 * id[0] + j];
                Dix[c * 3 + j] = e1 * R[c * 2 + 0] + e2 * R[c * 2 + 1];
            }
        double *ui = u + 6 * t;
        for (int i = 0; i < 6; ++i) zi[i] = Dix[i] + ui[i];
        orc_prox_tri(zi, lmin[t], lmax[t]);
        for (int i = 0; i < 6; ++i) { ui[i] += Dix[i] - zi[i]; z[6 * t + i] = zi[i]; }
    }
}

/* pins: SpringPin (src/SpringEnergyTerm.hpp:31-73): dim 6, rows 0-2 = the vertex, rows 3-5 are
 * never populated (restated as zero). prox: z = pin if active else identity. */
void orc_local_pins(int n, const int32_t *vidx, const double *pin, const int32_t *active,
                    const double *x, double *z, double *u) {
    for (int t = 0; t < n; ++t) {
        double *ui = u + 6 * t, *zo = z + 6 * t;
        for (int j = 0; j < 3; ++j) {
            double Dix = x[3 * vidx[t] + j];
            double zi = Dix + ui[j];
            if (active[t]) zi = pin[3 * t + j];
            ui[j] += Dix - zi;
            zo[j] = zi;
        }
        for (int j = 3; j < 6; ++j) { ui[j] = 0.0; zo[j] = 0.0; }
    }
}

/* ---- global step pieces ----------------------------------------------------------------------- */

/* y = A x for a CSR matrix */
void orc_csr_matvec(int n, const int32_t *rp, const int32_t *ci, const double *v, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) a += v[k] * x[ci[k]];
        y[i] = a;
    }
}

/*
 * src/NodalMultiColorGS.hpp:60-146,180-262 -- multi-colour nodal SOR on A = Ahat (x) I3.
 * Ahat in CSR (nv x nv).  colours: concatenated node lists, cptr[ncolors+1].
 * pin_flag[v] != 0 -> x_v = pin_xyz[v] (:111-117).  Passive objects: array of (kind, 4 params):
 * kind 0 = Floor(y0) (PassiveObject.hpp:32-45), kind 1 = Sphere(cx,cy,cz,r) (:48-64), kind 2 = a user-side plane (unit n, d); first object
 * with dx<0 wins (Collider.hpp:137-150).  Returns the sweep count like the reference (iter at break).
 */
static int passive_hit(int nobj, const int32_t *okind, const double *opar, const double *x, double *n, double *p) {
    double best = DBL_MAX; /* Payload ctor: dx = max (Collider.hpp:73) */
    for (int j = 0; j < nobj; ++j) {
        const double *q = opar + 4 * j;
        if (okind[j] == 0) {
            double dx = x[1] - q[0];
            if (!(dx > best)) { best = dx; p[0] = x[0]; p[1] = q[0]; p[2] = x[2]; n[0] = 0; n[1] = 1; n[2] = 0; }
        } else if (okind[j] == 2) { /* a user-side PassiveCollision (Collider.hpp:66-83): the half space n.x < d, q = unit n, d */
            double dx = q[0] * x[0] + q[1] * x[1] + q[2] * x[2] - q[3];
            if (!(dx > best)) { best = dx; for (int c = 0; c < 3; ++c) { n[c] = q[c]; p[c] = x[c] - dx * q[c]; } }
        } else {
            double dir[3] = {x[0] - q[0], x[1] - q[1], x[2] - q[2]};
            double l = norm3(dir), dx = l - q[3];
            if (!(dx > best)) {
                best = dx;
                for (int c = 0; c < 3; ++c) { dir[c] /= l; p[c] = q[c] + dir[c] * q[3]; n[c] = dir[c]; }
            }
        }
        if (best < 0) return 1; /* detect_passive returns at the first object with dx<0 */
    }
    return 0;
}

/* SLIDE pins inside the sweeps (README.md:23-28 TODO of the reference, no reference code: "parity unpinned"): pin_flag[v] == 2 -> the node
 * takes the plane-constrained Jacobi value of :218-262 on the plane through pin_xyz[v] with the unit normal g_pin_nrm[v] (set by
 * orc_gs_set_pin_normals before the solve): x_v = jac - n (n . (jac - p)), jac = D^-1 (b - LUx).  */
static const double *g_pin_nrm = 0;
void orc_gs_set_pin_normals(const double *nrm) { g_pin_nrm = nrm; }
static void slide_update(int v, const int32_t *rp, const int32_t *ci, const double *val, const double *b, const double *pin_xyz, double *x) {
    double LUx[3] = {0, 0, 0}, aii = 0.0, jac[3], d = 0.0;
    for (int k = rp[v]; k < rp[v + 1]; ++k) {
        if (fabs(val[k]) <= 0.0) continue;
        int col = ci[k];
        if (col == v) { aii = val[k]; continue; }
        for (int s = 0; s < 3; ++s) LUx[s] += val[k] * x[3 * col + s];
    }
    for (int s = 0; s < 3; ++s) { jac[s] = (b[3 * v + s] - LUx[s]) / aii; d += g_pin_nrm[3 * v + s] * (jac[s] - pin_xyz[3 * v + s]); }
    for (int s = 0; s < 3; ++s) x[3 * v + s] = jac[s] - d * g_pin_nrm[3 * v + s];
}

int orc_gs_solve(int nv, const int32_t *rp, const int32_t *ci, const double *val,
                 const double *b, double *x,
                 int ncolors, const int32_t *cptr, const int32_t *cnodes,
                 const int32_t *pin_flag, const double *pin_xyz,
                 int nobj, const int32_t *okind, const double *opar,
                 double omega, int max_iters, double tol) {
    double bnorm = 1.0, tol2 = tol * tol;
    if (tol > 0) { bnorm = 0.0; for (int i = 0; i < 3 * nv; ++i) bnorm += b[i] * b[i]; }
    int iter = 0;
    for (; iter < max_iters; ++iter) {
        for (int c = 0; c < ncolors; ++c) {
            int beg = cptr[c], end = cptr[c + 1];
#pragma omp parallel for schedule(static) if (end - beg > 31)
            for (int ii = beg; ii < end; ++ii) {
                int v = cnodes[ii];
                if (pin_flag && pin_flag[v] == 2 && g_pin_nrm) { slide_update(v, rp, ci, val, b, pin_xyz, x); continue; }
                if (pin_flag && pin_flag[v]) { for (int s = 0; s < 3; ++s) x[3 * v + s] = pin_xyz[3 * v + s]; continue; }
                double LUx[3] = {0, 0, 0}, aii = 0.0;
                for (int k = rp[v]; k < rp[v + 1]; ++k) {
                    if (fabs(val[k]) <= 0.0) continue;              /* :194 */
                    int col = ci[k];
                    if (col == v) { aii = val[k]; continue; }
                    for (int s = 0; s < 3; ++s) LUx[s] += val[k] * x[3 * col + s];
                }
                double cx[3], nx[3];
                for (int s = 0; s < 3; ++s) {
                    cx[s] = x[3 * v + s];
                    double xn = (b[3 * v + s] - LUx[s]) / aii;
                    nx[s] = (1.0 - omega) * cx[s] + omega * xn;     /* :210 */
                }
                double n[3], p[3];
                if (nobj > 0 && passive_hit(nobj, okind, opar, nx, n, p)) {
                    /* constrained_segment_update :218-262, orthoG :171-177 */
                    double dx[3], nn[3] = {0, 0, 0}, uu[3], vv[3];
                    for (int s = 0; s < 3; ++s) dx[s] = (b[3 * v + s] - LUx[s]) / aii - p[s];
                    if (n[0] > 0.999) nn[2] = 1.0; else nn[0] = 1.0;
                    cross3(nn, n, uu); double l = norm3(uu); for (int s = 0; s < 3; ++s) uu[s] /= l;
                    cross3(n, uu, vv); l = norm3(vv); for (int s = 0; s < 3; ++s) vv[s] /= l;
                    double t0 = uu[0] * dx[0] + uu[1] * dx[1] + uu[2] * dx[2];
                    double t1 = vv[0] * dx[0] + vv[1] * dx[1] + vv[2] * dx[2];
                    for (int s = 0; s < 3; ++s) nx[s] = uu[s] * t0 + vv[s] * t1 + p[s];
                }
                for (int s = 0; s < 3; ++s) x[3 * v + s] = nx[s];
            }
        }
        if (tol > 0) { /* :136-140 */
            double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+:err)
            for (int v = 0; v < nv; ++v) {
                double a[3] = {0, 0, 0};
                for (int k = rp[v]; k < rp[v + 1]; ++k)
                    for (int s = 0; s < 3; ++s) a[s] += val[k] * x[3 * ci[k] + s];
                for (int s = 0; s < 3; ++s) { double r = b[3 * v + s] - a[s]; err += r * r; }
            }
            if (err / bnorm < tol2) break;
        }
    }
    return iter;
}

/* ---- dynamic (self-)collision: TetMeshCollision ------------------------------------------------------------
 * src/DynamicObject.hpp:31-121 (signed_distance), src/Collider.hpp:152-212 (detect: dynamic objects in order,
 * "only resolve one dynamic collision at a time" = the first object with dx<0 keeps the payload).
 * The AABB trees and the point-in-tet / nearest-triangle traversals are mclscene's (mcl::bvh, ABSENT
 * un-vendored submodule) => PARITY UNPINNED at that boundary.  What the reference's own code fixes and this
 * restates: the query skips primitives that contain the query vertex (:78,:100); inside a tet -> barycentric
 * coordinates of the point in the CURRENT tet (:85-91) -> the same combination of the REST vertices (:92-97)
 * -> nearest REST surface triangle (:99-104) -> dx = -|proj - restx|, face, barycentrics of proj in the rest
 * triangle, normal of the REST triangle (:105-116).  Where the absent traversal order would decide, this
 * restatement decides by index: lowest tet index among the containing tets, lowest face index among equally
 * near triangles.  Arithmetic is FP64 throughout (the reference keeps the rest mesh in float).
 * Closest point on a triangle: Ericson, Real-Time Collision Detection, 5.1.5 (Voronoi regions).          */
static int tet_barycentric(const double *x, const double *p0, const double *p1, const double *p2, const double *p3, double *b) {
    double E[9], r[3], cf[3];
    for (int c = 0; c < 3; ++c) { E[c] = p1[c] - p0[c]; E[3 + c] = p2[c] - p0[c]; E[6 + c] = p3[c] - p0[c]; r[c] = x[c] - p0[c]; }
    double det = det3(E);
    if (det == 0.0 || !(det == det)) return 0;
    cross3(E + 3, E + 6, cf); b[1] = (r[0] * cf[0] + r[1] * cf[1] + r[2] * cf[2]) / det;
    cross3(E + 6, E, cf);     b[2] = (r[0] * cf[0] + r[1] * cf[1] + r[2] * cf[2]) / det;
    cross3(E, E + 3, cf);     b[3] = (r[0] * cf[0] + r[1] * cf[1] + r[2] * cf[2]) / det;
    b[0] = 1.0 - b[1] - b[2] - b[3];
    return 1;
}

static double dot3v(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* returns |p - proj|^2; bc = barycentrics of proj w.r.t. (a,b,c) */
static double closest_on_triangle(const double *p, const double *a, const double *b, const double *c, double *proj, double *bc) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    double d1 = dot3v(ab, ap), d2 = dot3v(ac, ap), d3 = dot3v(ab, bp), d4 = dot3v(ac, bp), d5 = dot3v(ab, cp), d6 = dot3v(ac, cp);
    double u, v, w;
    double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0.0 && d2 <= 0.0) { u = 1; v = 0; w = 0; }                                  /* vertex a */
    else if (d3 >= 0.0 && d4 <= d3) { u = 0; v = 1; w = 0; }                              /* vertex b */
    else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { v = d1 / (d1 - d3); u = 1 - v; w = 0; } /* edge ab */
    else if (d6 >= 0.0 && d5 <= d6) { u = 0; v = 0; w = 1; }                              /* vertex c */
    else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { w = d2 / (d2 - d6); u = 1 - w; v = 0; } /* edge ac */
    else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) { w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); v = 1 - w; u = 0; } /* edge bc */
    else { double den = 1.0 / (va + vb + vc); v = vb * den; w = vc * den; u = 1.0 - v - w; } /* face */
    double d = 0.0;
    for (int i = 0; i < 3; ++i) { proj[i] = u * a[i] + v * b[i] + w * c[i]; d += (p[i] - proj[i]) * (p[i] - proj[i]); }
    bc[0] = u; bc[1] = v; bc[2] = w;
    return d;
}

/* One TetMeshCollision against nq query vertices (query == NULL: vertices 0..nq-1) at positions x (global node
 * vector).  tets / faces index the mesh's own vertices (0..n_mesh_verts-1), rest = its rest vertices; global id =
 * local + vert_offset.  hit[q] != 0 on entry = an earlier object already holds the payload (skipped, :73).
 * Outputs per query (written only for new hits): face (global ids), barys, normal, dx (<0).               */
void orc_detect_dynamic(int nq, const int32_t *query, const double *x, int vert_offset, const double *rest,
                        int ntets, const int32_t *tets, int nfaces, const int32_t *faces,
                        int32_t *hit, int32_t *face_out, double *bary_out, double *normal_out, double *dx_out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int q = 0; q < nq; ++q) {
        if (hit[q]) continue;
        const int vg = query ? query[q] : q, vl = vg - vert_offset;
        const double *px = x + 3 * (size_t)vg;
        int found = -1; double fb[4] = {0, 0, 0, 0};
        for (int t = 0; t < ntets && found < 0; ++t) {
            const int32_t *tt = tets + 4 * (size_t)t;
            if (tt[0] == vl || tt[1] == vl || tt[2] == vl || tt[3] == vl) continue;          /* skip_vert_idx :78 */
            const double *p0 = x + 3 * (size_t)(tt[0] + vert_offset), *p1 = x + 3 * (size_t)(tt[1] + vert_offset);
            const double *p2 = x + 3 * (size_t)(tt[2] + vert_offset), *p3 = x + 3 * (size_t)(tt[3] + vert_offset);
            /* cheap reject: bounding box */
            int out = 0;
            for (int c = 0; c < 3 && !out; ++c) {
                double lo = fmin(fmin(p0[c], p1[c]), fmin(p2[c], p3[c])), hi = fmax(fmax(p0[c], p1[c]), fmax(p2[c], p3[c]));
                out = (px[c] < lo) || (px[c] > hi);
            }
            if (out) continue;
            double b[4];
            if (!tet_barycentric(px, p0, p1, p2, p3, b)) continue;
            if (b[0] >= 0.0 && b[1] >= 0.0 && b[2] >= 0.0 && b[3] >= 0.0) { found = t; memcpy(fb, b, sizeof fb); }
        }
        if (found < 0) continue;
        const int32_t *tt = tets + 4 * (size_t)found;
        double restx[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) for (int c = 0; c < 3; ++c) restx[c] += fb[k] * rest[3 * (size_t)tt[k] + c];   /* :92-97 */
        int best = -1; double bd = DBL_MAX, bproj[3] = {0, 0, 0}, bbc[3] = {0, 0, 0};
        for (int f = 0; f < nfaces; ++f) {
            const int32_t *ff = faces + 3 * (size_t)f;
            if (ff[0] == vl || ff[1] == vl || ff[2] == vl) continue;                           /* :100 */
            double proj[3], bc[3];
            double d = closest_on_triangle(restx, rest + 3 * (size_t)ff[0], rest + 3 * (size_t)ff[1], rest + 3 * (size_t)ff[2], proj, bc);
            if (d < bd) { bd = d; best = f; memcpy(bproj, proj, sizeof proj); memcpy(bbc, bc, sizeof bc); }
        }
        if (best < 0) continue;   /* the reference throws (:102-104); a mesh whose every face touches the vertex */
        const int32_t *ff = faces + 3 * (size_t)best;
        const double *a = rest + 3 * (size_t)ff[0], *b = rest + 3 * (size_t)ff[1], *c = rest + 3 * (size_t)ff[2];
        double e1[3], e2[3], nrm[3];
        for (int i = 0; i < 3; ++i) { e1[i] = b[i] - a[i]; e2[i] = c[i] - a[i]; }
        cross3(e1, e2, nrm);
        double l = norm3(nrm);
        double dx = -sqrt(bd);                                                                   /* :112 */
        if (!(dx < 0.0)) continue;                                                               /* Collider.hpp:203 */
        hit[q] = 1; dx_out[q] = dx;
        for (int i = 0; i < 3; ++i) {
            face_out[3 * (size_t)q + i] = ff[i] + vert_offset;                                   /* :113 */
            bary_out[3 * (size_t)q + i] = bbc[i];                                                /* :114 */
            normal_out[3 * (size_t)q + i] = nrm[i] / l;                                          /* :110-111,:115 */
        }
        (void)bproj;
    }
}

/* NodalMultiColorGS::solve on a GENERAL dof x dof matrix (src/NodalMultiColorGS.hpp:60-146, segment_update :180-215,
 * constrained_segment_update :218-262): the sweeps after "A <- A + C^T C" for dynamic hits (:80-86), where the rows
 * of a node are no longer three copies of one scalar row.  M in CSR (3 nv rows); everything else as orc_gs_solve.
 * The re-colouring of M (:85, mclscene, ABSENT) is an input. */
int orc_gs_solve_full(int nv, const int32_t *rp, const int32_t *ci, const double *val,
                      const double *b, double *x,
                      int ncolors, const int32_t *cptr, const int32_t *cnodes,
                      const int32_t *pin_flag, const double *pin_xyz,
                      int nobj, const int32_t *okind, const double *opar,
                      double omega, int max_iters, double tol) {
    double bnorm = 1.0, tol2 = tol * tol;
    if (tol > 0) { bnorm = 0.0; for (int i = 0; i < 3 * nv; ++i) bnorm += b[i] * b[i]; }
    int iter = 0;
    for (; iter < max_iters; ++iter) {
        for (int c = 0; c < ncolors; ++c) {
            int beg = cptr[c], end = cptr[c + 1];
#pragma omp parallel for schedule(static) if (end - beg > 31)
            for (int ii = beg; ii < end; ++ii) {
                int v = cnodes[ii];
                if (pin_flag && pin_flag[v] == 2 && g_pin_nrm) { slide_update(v, rp, ci, val, b, pin_xyz, x); continue; }
                if (pin_flag && pin_flag[v]) { for (int s = 0; s < 3; ++s) x[3 * v + s] = pin_xyz[3 * v + s]; continue; }
                double LUx[3] = {0, 0, 0}, aii[3] = {0, 0, 0}, nx[3], jac[3];
                for (int s = 0; s < 3; ++s) {
                    int row = 3 * v + s;
                    for (int k = rp[row]; k < rp[row + 1]; ++k) {
                        if (fabs(val[k]) <= 0.0) continue;              /* :194 */
                        if (ci[k] == row) { aii[s] = val[k]; continue; }
                        LUx[s] += val[k] * x[ci[k]];
                    }
                }
                for (int s = 0; s < 3; ++s) {
                    jac[s] = (b[3 * v + s] - LUx[s]) / aii[s];
                    nx[s] = (1.0 - omega) * x[3 * v + s] + omega * jac[s]; /* :210 */
                }
                double n[3], p[3];
                if (nobj > 0 && passive_hit(nobj, okind, opar, nx, n, p)) {
                    double dx[3], nn[3] = {0, 0, 0}, uu[3], vv[3];
                    for (int s = 0; s < 3; ++s) dx[s] = jac[s] - p[s];
                    if (n[0] > 0.999) nn[2] = 1.0; else nn[0] = 1.0;
                    cross3(nn, n, uu); double l = norm3(uu); for (int s = 0; s < 3; ++s) uu[s] /= l;
                    cross3(n, uu, vv); l = norm3(vv); for (int s = 0; s < 3; ++s) vv[s] /= l;
                    double t0 = uu[0] * dx[0] + uu[1] * dx[1] + uu[2] * dx[2];
                    double t1 = vv[0] * dx[0] + vv[1] * dx[1] + vv[2] * dx[2];
                    for (int s = 0; s < 3; ++s) nx[s] = uu[s] * t0 + vv[s] * t1 + p[s];
                }
                for (int s = 0; s < 3; ++s) x[3 * v + s] = nx[s];
            }
        }
        if (tol > 0) { /* :136-140 */
            double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+:err)
            for (int row = 0; row < 3 * nv; ++row) {
                double a = 0.0;
                for (int k = rp[row]; k < rp[row + 1]; ++k) a += val[k] * x[ci[k]];
                double r = b[row] - a; err += r * r;
            }
            if (err / bnorm < tol2) break;
        }
    }
    return iter;
}
