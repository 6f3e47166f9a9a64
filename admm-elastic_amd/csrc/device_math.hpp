// device_math.hpp -- per-element FP64 math of the ADMM local step for gfx950 (one lane = one element).
//
// Reference behaviour restated (file:line relative to the reference repo):
//   signed SVD                     src/FastSVD.hpp:43-68
//   linear tet prox                src/TetEnergyTerm.cpp:73-92
//   hyperelastic prox (NH, StVK)   src/TetEnergyTerm.cpp:114-136, :173-237
//   triangle prox + strain limit   src/TriEnergyTerm.cpp:73-101
//
// Design notes (MI355X): everything lives in VGPRs (arrays are fully unrolled -> SROA); the SVD is a
// cyclic Jacobi eigen-solve of the symmetric 3x3 F^T F (cheap rotations: 1 rcp + 1 sqrt + 1 rsqrt each)
// followed by a Gram-Schmidt reconstruction of U from F V, which (a) returns U,V in SO(3) by
// construction, (b) recovers the *signed* smallest stretch as u2 . F v2 directly from F, so inverted
// and flat elements need no special-casing.  The principal-stretch minimisation is a safeguarded
// Newton with the analytic diag + rank-one Hessian (Sherman-Morrison solve, no 3x3 factorisation),
// iterated to the exact minimiser -- the reference's L-BFGS (absent mcloptlib) stops at
// |g|<1e-6 or |dx|<1e-6 of the same objective (src/TetEnergyTerm.hpp:93-95).
#pragma once
#include <hip/hip_runtime.h>

namespace admm_dev {

#define ADMM_M3(A, r, c) ((A)[(c) * 3 + (r)])

__device__ __forceinline__ double dot3(const double *a, const double *b) {
    return fma(a[0], b[0], fma(a[1], b[1], a[2] * b[2]));
}
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// One Jacobi rotation annihilating a_pq of a symmetric 3x3; r is the third index.
// vp, vq: the two affected eigenvector columns.
__device__ __forceinline__ bool jacobi_rotate(double &app, double &aqq, double &apq, double &arp, double &arq,
                                              double *vp, double *vq) {
    const double scale = fabs(app) + fabs(aqq);
    if (!(fabs(apq) > 1e-17 * scale)) { return false; }
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
    const double c = rsqrt(fma(t, t, 1.0));
    const double s = t * c;
    app = fma(-t, apq, app);
    aqq = fma(t, apq, aqq);
    apq = 0.0;
    const double rp = arp, rq = arq;
    arp = c * rp - s * rq;
    arq = s * rp + c * rq;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = vp[i], b = vq[i];
        vp[i] = c * a - s * b;
        vq[i] = s * a + c * b;
    }
    return true;
}

// Signed SVD  F = U diag(S) V^T,  U,V in SO(3),  S[0] >= S[1] >= |S[2]|, sign(S[2]) = sign(det F).
// F, U, V column-major.
__device__ __forceinline__ void signed_svd3(const double *F, double *U, double *S, double *V) {
    // C = F^T F
    double c00 = dot3(F + 0, F + 0), c01 = dot3(F + 0, F + 3), c02 = dot3(F + 0, F + 6);
    double c11 = dot3(F + 3, F + 3), c12 = dot3(F + 3, F + 6), c22 = dot3(F + 6, F + 6);
    double v0[3] = {1, 0, 0}, v1[3] = {0, 1, 0}, v2[3] = {0, 0, 1};
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool any = false;
        any |= jacobi_rotate(c00, c11, c01, c02, c12, v0, v1);
        any |= jacobi_rotate(c00, c22, c02, c01, c12, v0, v2);
        any |= jacobi_rotate(c11, c22, c12, c01, c02, v1, v2);
        if (!any) break;
    }
    // B = F V ; stretches are the column norms (more accurate than sqrt of the eigenvalues)
    double b0[3], b1[3], b2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        b0[r] = fma(F[r], v0[0], fma(F[3 + r], v0[1], F[6 + r] * v0[2]));
        b1[r] = fma(F[r], v1[0], fma(F[3 + r], v1[1], F[6 + r] * v1[2]));
        b2[r] = fma(F[r], v2[0], fma(F[3 + r], v2[1], F[6 + r] * v2[2]));
    }
    double n0 = dot3(b0, b0), n1 = dot3(b1, b1), n2 = dot3(b2, b2);
    // sort descending (swap b and v columns together)
#define ADMM_SWAP3(x, y) { _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) { double t_ = x[i_]; x[i_] = y[i_]; y[i_] = t_; } }
    if (n0 < n1) { ADMM_SWAP3(b0, b1); ADMM_SWAP3(v0, v1); double t = n0; n0 = n1; n1 = t; }
    if (n0 < n2) { ADMM_SWAP3(b0, b2); ADMM_SWAP3(v0, v2); double t = n0; n0 = n2; n2 = t; }
    if (n1 < n2) { ADMM_SWAP3(b1, b2); ADMM_SWAP3(v1, v2); double t = n1; n1 = n2; n2 = t; }
#undef ADMM_SWAP3
    // make V a rotation: v2 = v0 x v1 (equals +-v2); flip b2 with it
    double vx[3];
    cross3(v0, v1, vx);
    if (dot3(vx, v2) < 0.0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { v2[i] = -v2[i]; b2[i] = -b2[i]; }
    }
    // U by Gram-Schmidt on b0, b1; u2 = u0 x u1
    double u0[3], u1[3], u2[3];
    const double s0 = sqrt(n0);
    if (s0 > 1e-300) {
        const double inv = 1.0 / s0;
#pragma unroll
        for (int i = 0; i < 3; ++i) u0[i] = b0[i] * inv;
    } else { u0[0] = 1.0; u0[1] = 0.0; u0[2] = 0.0; }
    const double pr = dot3(u0, b1);
    double w[3] = {fma(-pr, u0[0], b1[0]), fma(-pr, u0[1], b1[1]), fma(-pr, u0[2], b1[2])};
    double wn = dot3(w, w);
    if (!(wn > 1e-30 * n0) || !(wn > 1e-300)) {
        // rank <= 1: any unit vector orthogonal to u0
        double e[3] = {0.0, 0.0, 0.0};
        const double a0 = fabs(u0[0]), a1 = fabs(u0[1]), a2 = fabs(u0[2]);
        if (a0 <= a1 && a0 <= a2) e[0] = 1.0; else if (a1 <= a2) e[1] = 1.0; else e[2] = 1.0;
        cross3(u0, e, w);
        wn = dot3(w, w);
    }
    {
        const double inv = rsqrt(wn);
#pragma unroll
        for (int i = 0; i < 3; ++i) u1[i] = w[i] * inv;
    }
    cross3(u0, u1, u2);
    S[0] = s0;
    S[1] = dot3(u1, b1);
    S[2] = dot3(u2, b2);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        U[i] = u0[i]; U[3 + i] = u1[i]; U[6 + i] = u2[i];
        V[i] = v0[i]; V[3 + i] = v1[i]; V[6 + i] = v2[i];
    }
}

// out = U diag(s) V^T
__device__ __forceinline__ void usvt(const double *U, const double *s, const double *V, double *out) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            ADMM_M3(out, r, c) = fma(U[r] * s[0], V[c], fma(U[3 + r] * s[1], V[3 + c], U[6 + r] * s[2] * V[6 + c]));
}

// ---- principal-stretch objectives --------------------------------------------------------------
// KIND 1: Neo-Hookean  Psi = mu/2 (I1 - ln I3 - 3) + lambda/8 ln^2 I3     (TetEnergyTerm.cpp:173-182)
// KIND 2: StVK         Psi = mu |E|^2 + lambda/2 tr(E)^2, E_i=(s_i^2-1)/2  (TetEnergyTerm.cpp:220-226)
// objective = Psi(s) + k/2 |s - x0|^2                                      (:184-192, :210-218)
template <int KIND>
struct StretchModel {
    double mu, la, k;
    double x0[3];

    // value, gradient and the Hessian split H = diag(D) + la * w w^T
    __device__ __forceinline__ double eval(const double *s, double *g, double *D, double *w) const {
        if (KIND == 1) {
            const double J = s[0] * s[1] * s[2];
            const double lJ = log(J);
            double f = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double si = 1.0 / s[i];
                const double d = s[i] - x0[i];
                w[i] = si;
                g[i] = fma(mu, s[i] - si, fma(la * lJ, si, k * d));
                D[i] = fma(mu, fma(si, si, 1.0), fma(-la * lJ, si * si, k));
                f = fma(s[i], s[i], f);
                q = fma(d, d, q);
            }
            return 0.5 * mu * (f - 3.0) - mu * lJ + 0.5 * la * lJ * lJ + 0.5 * k * q;
        } else {
            const double ss = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
            const double trE = 0.5 * (ss - 3.0);
            double ee = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double E = 0.5 * (s[i] * s[i] - 1.0);
                const double d = s[i] - x0[i];
                w[i] = s[i];
                g[i] = fma(mu * s[i], s[i] * s[i] - 1.0, fma(la * trE, s[i], k * d));
                D[i] = fma(mu, 3.0 * s[i] * s[i] - 1.0, fma(la, trE, k));
                ee = fma(E, E, ee);
                q = fma(d, d, q);
            }
            return mu * ee + 0.5 * la * trE * trE + 0.5 * k * q;
        }
    }
    __device__ __forceinline__ double value(const double *s) const {
        if (KIND == 1) {
            const double lJ = log(s[0] * s[1] * s[2]);
            double f = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double d = s[i] - x0[i]; f = fma(s[i], s[i], f); q = fma(d, d, q); }
            return 0.5 * mu * (f - 3.0) - mu * lJ + 0.5 * la * lJ * lJ + 0.5 * k * q;
        } else {
            const double ss = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
            const double trE = 0.5 * (ss - 3.0);
            double ee = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double E = 0.5 * (s[i] * s[i] - 1.0);
                const double d = s[i] - x0[i];
                ee = fma(E, E, ee);
                q = fma(d, d, q);
            }
            return mu * ee + 0.5 * la * trE * trE + 0.5 * k * q;
        }
    }
    // NH needs s > 0 (log barrier); StVK accepts s >= 0 (value() returns FLT_MAX only for s < 0)
    __device__ __forceinline__ bool feasible(const double *s) const {
        if (KIND == 1) return s[0] > 0.0 && s[1] > 0.0 && s[2] > 0.0;
        return s[0] >= 0.0 && s[1] >= 0.0 && s[2] >= 0.0;
    }
};

// argmin_s Psi(s) + k/2 |s - x0|^2 by safeguarded Newton, started from s (in/out). Returns iterations.
// NH: the log barrier keeps the iterates strictly positive.  StVK: the feasible set is s >= 0 (value()
// is FLT_MAX only for s < 0) and for inverted elements the minimiser sits ON that boundary, so the
// iteration is a projected Newton: components at the bound whose gradient points outward are frozen,
// the rest take the reduced Newton step, and trial points are projected back onto s >= 0.
template <int KIND>
__device__ __forceinline__ int minimize_stretch(const StretchModel<KIND> &m, double *s) {
    double g[3], D[3], w[3];
    double f = m.eval(s, g, D, w);
    int it = 0;
    for (; it < 60; ++it) {
        // modified Hessian: keep the diagonal part safely positive
        const double floorD = 1e-8 * (fabs(m.k) + fabs(m.mu)) + 1e-300;
        double a[3], y[3];
        double wDg = 0.0, wDw = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double Di = fmax(fabs(D[i]), floorD);
            const bool active = (KIND == 2) && (s[i] <= 0.0) && (g[i] > 0.0);
            a[i] = active ? 0.0 : 1.0 / Di;
            y[i] = g[i] * a[i];
            wDg = fma(w[i], y[i], wDg);
            wDw = fma(w[i] * w[i], a[i], wDw);
        }
        const double den = fma(m.la, wDw, 1.0);
        const double coef = (den > 1e-12) ? m.la * wDg / den : 0.0;
        double d[3], gd = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            d[i] = -(y[i] - coef * w[i] * a[i]);
            gd = fma(g[i], d[i], gd);
        }
        if (!(gd < 0.0)) { // not a descent direction (indefinite rank-one part): scaled steepest descent
            gd = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { d[i] = -y[i]; gd = fma(g[i], d[i], gd); }
            if (!(gd < 0.0)) break; // zero (reduced) gradient
        }
        double t = 1.0, sn[3], fn = f;
        bool ok = false;
        for (int ls = 0; ls < 50; ++ls) {
            double gs = 0.0; // g . (sn - s) along the projected path
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sn[i] = fma(t, d[i], s[i]);
                if (KIND == 2) sn[i] = fmax(sn[i], 0.0);
                gs = fma(g[i], sn[i] - s[i], gs);
            }
            if (m.feasible(sn)) {
                fn = m.value(sn);
                if (fn <= f + 1e-4 * gs + 1e-15 * fabs(f)) { ok = true; break; }
            }
            t *= 0.5;
        }
        if (!ok) break;
        double step = 0.0, mag = 1.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            step = fmax(step, fabs(sn[i] - s[i]));
            mag = fmax(mag, fabs(sn[i]));
            s[i] = sn[i];
        }
        f = m.eval(s, g, D, w);
        if (step <= 1e-10 * mag) { ++it; break; }
    }
    return it;
}

// src/TetEnergyTerm.cpp:73-92.  q (col-major 3x3) -> z
__device__ __forceinline__ void prox_tet_linear(const double *q, double *z) {
    double U[9], S[3], V[9];
    signed_svd3(q, U, S, V);
    // P = U diag(1,1,sign det F) V^T with Eigen's unsigned factors == U V^T with the signed ones
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double p = fma(U[r], V[c], fma(U[3 + r], V[3 + c], U[6 + r] * V[6 + c]));
            ADMM_M3(z, r, c) = 0.5 * (p + ADMM_M3(q, r, c));
        }
}

// src/TetEnergyTerm.cpp:114-136
template <int KIND>
__device__ __forceinline__ void prox_tet_hyper(double mu, double la, double k, const double *q, double *z) {
    double U[9], S[3], V[9];
    signed_svd3(q, U, S, V);
    StretchModel<KIND> m;
    m.mu = mu; m.la = la; m.k = k;
    m.x0[0] = S[0]; m.x0[1] = S[1]; m.x0[2] = S[2];       // :124 (before the fix-ups)
    const double eps = 1e-6;
    if (fabs(S[0]) < eps && fabs(S[1]) < eps && fabs(S[2]) < eps) { S[0] = eps; S[1] = eps; S[2] = eps; } // :128-131
    if (S[2] < 0.0) S[2] = -S[2];                           // :133
    if (KIND == 1) { // keep the start strictly inside the log barrier
        S[0] = fmax(S[0], 1e-12); S[1] = fmax(S[1], 1e-12); S[2] = fmax(S[2], 1e-12);
    }
    minimize_stretch<KIND>(m, S);
    usvt(U, S, V, z);
}

// src/TriEnergyTerm.cpp:73-101.  q = 3x2 column-major (6 doubles) -> z
__device__ __forceinline__ void prox_tri(const double *q, double lmin, double lmax, double *z) {
    // C = F^T F (2x2); one exact Jacobi rotation diagonalises it
    double c00 = dot3(q, q), c01 = dot3(q, q + 3), c11 = dot3(q + 3, q + 3);
    double cs = 1.0, sn = 0.0;
    if (fabs(c01) > 1e-17 * (c00 + c11)) {
        const double theta = (c11 - c00) / (2.0 * c01);
        const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
        cs = rsqrt(fma(t, t, 1.0));
        sn = t * cs;
    }
    // v0 = (cs, -sn), v1 = (sn, cs);  b_j = F v_j
    double b0[3], b1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        b0[r] = cs * q[r] - sn * q[3 + r];
        b1[r] = sn * q[r] + cs * q[3 + r];
    }
    double n0 = dot3(b0, b0), n1 = dot3(b1, b1);
    double u0[3], u1[3];
    // order so that (b0,n0) is the larger one
    bool swapped = false;
    if (n0 < n1) {
        swapped = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) { double t = b0[r]; b0[r] = b1[r]; b1[r] = t; }
        double t = n0; n0 = n1; n1 = t;
    }
    if (n0 > 1e-300) {
        const double inv = rsqrt(n0);
#pragma unroll
        for (int r = 0; r < 3; ++r) u0[r] = b0[r] * inv;
    } else { u0[0] = 1.0; u0[1] = 0.0; u0[2] = 0.0; }
    const double pr = dot3(u0, b1);
    double w[3] = {fma(-pr, u0[0], b1[0]), fma(-pr, u0[1], b1[1]), fma(-pr, u0[2], b1[2])};
    double wn = dot3(w, w);
    if (!(wn > 1e-30 * n0) || !(wn > 1e-300)) {
        double e[3] = {0.0, 0.0, 0.0};
        const double a0 = fabs(u0[0]), a1 = fabs(u0[1]), a2 = fabs(u0[2]);
        if (a0 <= a1 && a0 <= a2) e[0] = 1.0; else if (a1 <= a2) e[1] = 1.0; else e[2] = 1.0;
        cross3(u0, e, w);
        wn = dot3(w, w);
    }
    {
        const double inv = rsqrt(wn);
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] = w[r] * inv;
    }
    // P = u0 v0^T + u1 v1^T  (v's in the possibly swapped order)
    double va[2], vb[2];
    if (!swapped) { va[0] = cs; va[1] = -sn; vb[0] = sn; vb[1] = cs; }
    else          { va[0] = sn; va[1] = cs;  vb[0] = cs; vb[1] = -sn; }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double p = fma(u0[r], va[c], u1[r] * vb[c]);
            z[c * 3 + r] = 0.5 * (p + q[c * 3 + r]);
        }
    if (lmin > 0.0 || lmax < 99.0) { // :91-99
        const double l0 = sqrt(dot3(z, z)), l1 = sqrt(dot3(z + 3, z + 3));
        double f0 = 1.0, f1 = 1.0;
        if (l0 < lmin) f0 = lmin / l0;
        if (l1 < lmin) f1 = lmin / l1;
        if (l0 > lmax) f0 = lmax / l0;
        if (l1 > lmax) f1 = lmax / l1;
#pragma unroll
        for (int r = 0; r < 3; ++r) { z[r] *= f0; z[3 + r] *= f1; }
    }
}

} // namespace admm_dev
