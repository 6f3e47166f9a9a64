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
#ifndef ADMM_COUNT
#define ADMM_COUNT(slot)      // experiments/hostmath counts sweeps / rotations through this; nothing on the device
#endif
#ifndef ADMM_RECORD
#define ADMM_RECORD(slot, value)
#endif

__device__ __forceinline__ double dot3(const double *a, const double *b) {
    return fma(a[0], b[0], fma(a[1], b[1], a[2] * b[2]));
}
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// ---- fast FP64 reciprocal / rsqrt: hardware seed + two Newton steps (no IEEE div/sqrt sequences) ----
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0); r = fma(r, e, r);
    e = fma(-x, r, 1.0); r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    return y;
}

// FP32 rotation used only to SEED the eigenvectors (quarter-rate hardware rcp/sqrt/rsq).  Unconditional: with b = 0 it is
// the identity (t = 0, c = 1), so there is no branch whose two sides the compiler would have to merge with register copies.
__device__ __forceinline__ void jacobi_rotate_f32(float &app, float &aqq, float &apq, float &arp, float &arq,
                                                  float *vp, float *vq) {
    const float a = aqq - app, b = apq + apq;
    const float h = __builtin_amdgcn_sqrtf(fmaxf(fmaf(a, a, b * b), 1e-36f));
    const float t = b * __builtin_amdgcn_rcpf(a + copysignf(h, a));      // sgn(a) b / (|a| + h)
    const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.0f));
    const float s = t * c;
    app = fmaf(-t, apq, app);
    aqq = fmaf(t, apq, aqq);
    apq = 0.0f;
    const float rp = arp, rq = arq;
    arp = c * rp - s * rq;
    arq = fmaf(s, rp, c * rq);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float x = vp[i], y = vq[i];
        vp[i] = c * x - s * y;
        vq[i] = fmaf(s, x, c * y);
    }
}

// One-sided (Hestenes) Jacobi rotation in FP64: rotates columns p, q of B = F V (and of V) so that b_p . b_q = 0.
// al = |b_p|^2, be = |b_q|^2, ga = b_p . b_q.  Same angle as the two-sided rotation of F^T F: t = sgn(a) b / (|a| + sqrt(a^2 + b^2)).
// t only steers the convergence (one Newton step on the 2^-24 hardware seeds); c decides the orthogonality of V (two steps).
__device__ __forceinline__ void hestenes_rotate(double *bp, double *bq, double *vp, double *vq, double al, double be, double ga) {
    ADMM_COUNT(2);
    const double a = be - al, b = ga + ga;
    const double h2 = fmax(fma(a, a, b * b), 1e-300);
    double y = __builtin_amdgcn_rsq(h2);
    y = fma(y, fma(-0.5 * h2 * y, y, 0.5), y);
    const double den = fma(h2, y, fabs(a));
    double r = __builtin_amdgcn_rcp(den);
    r = fma(r, fma(-den, r, 1.0), r);
    const double t = copysign(b, a * b) * r;        // sgn(a) b / den   (a = 0: the sign of b, i.e. a 45 degree rotation)
    const double c = fast_rsqrt(fma(t, t, 1.0));
    const double s = t * c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double x = bp[i], z = bq[i];
        bp[i] = c * x - s * z;
        bq[i] = fma(s, x, c * z);
        const double vx = vp[i], vz = vq[i];
        vp[i] = c * vx - s * vz;
        vq[i] = fma(s, vx, c * vz);
    }
}

// Signed SVD  F = U diag(S) V^T with U, V in SO(3).  S is NOT sorted (every energy of the library is a symmetric function of
// the stretches; the reference sorts, src/FastSVD.hpp:43-68); at most one entry is negative, and then it is the one of
// smallest magnitude (the reference's convention: the sign of det F goes to the smallest stretch).
// F, U, V column-major.  Mixed precision, FP64 result:
//  (1) FP32 cyclic Jacobi on the DEVIATORIC part of F^T F, scaled to unit norm: the eigenvectors are those of F^T F, and an
//      element at strain 1e-6 is resolved as well as one at strain 1 (on the trace-normalised matrix the FP32 phase stalled at
//      1e-7 / strain and the FP64 phase paid for it with extra sweeps).  Three sweeps, a fourth if some lane of the wave is not
//      at the FP32 floor yet (wave-uniform decisions: straight-line code, no merges).
//  (2) V is re-orthonormalised in FP64, B = F V.
//  (3) ONE one-sided (Hestenes) FP64 Jacobi sweep on the columns of B -- Jacobi converges quadratically, so the 1e-7 of the seed
//      becomes round-off; further sweeps only while some pair of columns of some lane is not orthogonal to 3e-14 (cosine).
//  (4) U by Gram-Schmidt on B: U in SO(3) by construction, the last stretch u2 . b2 carries the sign of det F; if it is negative
//      and not the smallest, the sign is moved by flipping two columns of U (no sorting, no special case for flat elements).
__device__ __forceinline__ void signed_svd3(const double *F, double *U, double *S, double *V) {
    double v0[3], v1[3], v2[3];
    const double c00 = dot3(F + 0, F + 0), c01 = dot3(F + 0, F + 3), c02 = dot3(F + 0, F + 6);
    const double c11 = dot3(F + 3, F + 3), c12 = dot3(F + 3, F + 6), c22 = dot3(F + 6, F + 6);
    const double m = (c00 + c11 + c22) * (1.0 / 3.0);
    const double d0 = c00 - m, d1 = c11 - m, d2 = c22 - m;
    const double o2 = fma(c01, c01, fma(c02, c02, c12 * c12));
    const double nd2 = fma(d0, d0, fma(d1, d1, fma(d2, d2, o2 + o2)));
    const bool seeded = nd2 > 1e-30 * m * m && nd2 > 1e-280;     // else: a multiple of the identity to round-off, V = I
    {
        const double inv = seeded ? __builtin_amdgcn_rsq(nd2) : 0.0;
        float a00 = (float)(d0 * inv), a01 = (float)(c01 * inv), a02 = (float)(c02 * inv);
        float a11 = (float)(d1 * inv), a12 = (float)(c12 * inv), a22 = (float)(d2 * inv);
        float w0[3] = {1.f, 0.f, 0.f}, w1[3] = {0.f, 1.f, 0.f}, w2[3] = {0.f, 0.f, 1.f};
#ifndef ADMM_F32_OFF2
#define ADMM_F32_OFF2 2e-13f
#endif
#define ADMM_F32_SWEEP() { ADMM_COUNT(0); jacobi_rotate_f32(a00, a11, a01, a02, a12, w0, w1); jacobi_rotate_f32(a00, a22, a02, a01, a12, w0, w2); \
                           jacobi_rotate_f32(a11, a22, a12, a01, a02, w1, w2); }
        ADMM_F32_SWEEP(); ADMM_F32_SWEEP(); ADMM_F32_SWEEP();
        if (__any(fmaf(a01, a01, fmaf(a02, a02, a12 * a12)) > ADMM_F32_OFF2)) ADMM_F32_SWEEP();
#undef ADMM_F32_SWEEP
        // (2) orthonormalise in FP64 (Gram-Schmidt; the columns are unit vectors to 1e-7: 1/sqrt(1 + e) by its series)
#pragma unroll
        for (int i = 0; i < 3; ++i) { v0[i] = (double)w0[i]; v1[i] = (double)w1[i]; }
        double e = dot3(v0, v0) - 1.0;
        double n = fma(e, fma(e, 0.375, -0.5), 1.0);
#pragma unroll
        for (int i = 0; i < 3; ++i) v0[i] *= n;
        const double pr01 = dot3(v0, v1);
#pragma unroll
        for (int i = 0; i < 3; ++i) v1[i] = fma(-pr01, v0[i], v1[i]);
        e = dot3(v1, v1) - 1.0;
        n = fma(e, fma(e, 0.375, -0.5), 1.0);
#pragma unroll
        for (int i = 0; i < 3; ++i) v1[i] *= n;
        cross3(v0, v1, v2);
    }
    // B = F V
    double b0[3], b1[3], b2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        b0[r] = fma(F[r], v0[0], fma(F[3 + r], v0[1], F[6 + r] * v0[2]));
        b1[r] = fma(F[r], v1[0], fma(F[3 + r], v1[1], F[6 + r] * v1[2]));
        b2[r] = fma(F[r], v2[0], fma(F[3 + r], v2[1], F[6 + r] * v2[2]));
    }
    // (3) FP64 one-sided sweeps until every pair of columns is orthogonal: cos^2 <= ADMM_SVD_TOL2.  Then F = U diag(S) V^T to
    // ~3e-14 |F| whatever the order of the columns in the Gram-Schmidt below.
#ifndef ADMM_SVD_TOL2
#define ADMM_SVD_TOL2 1e-27
#endif
    double n0, n1, n2;
#pragma unroll 1
    for (int sweep = 0; sweep < 12; ++sweep) {
        hestenes_rotate(b0, b1, v0, v1, dot3(b0, b0), dot3(b1, b1), dot3(b0, b1));
        hestenes_rotate(b0, b2, v0, v2, dot3(b0, b0), dot3(b2, b2), dot3(b0, b2));
        hestenes_rotate(b1, b2, v1, v2, dot3(b1, b1), dot3(b2, b2), dot3(b1, b2));
        ADMM_COUNT(1);
        n0 = dot3(b0, b0); n1 = dot3(b1, b1); n2 = dot3(b2, b2);
        const double g01 = dot3(b0, b1), g02 = dot3(b0, b2), g12 = dot3(b1, b2);
        ADMM_RECORD(8 + sweep, fmax(g01 * g01 / fmax(n0 * n1, 1e-300), fmax(g02 * g02 / fmax(n0 * n2, 1e-300), g12 * g12 / fmax(n1 * n2, 1e-300))));
        const bool open = g01 * g01 > ADMM_SVD_TOL2 * n0 * n1 || g02 * g02 > ADMM_SVD_TOL2 * n0 * n2 || g12 * g12 > ADMM_SVD_TOL2 * n1 * n2;
        if (!__any(open)) break;
    }
    // (4) U by Gram-Schmidt on b0, b1; u2 = u0 x u1.  A column that is round-off (flat or collapsed element) must not be one
    // of the two the basis is built from: only then (rare, whole waves skip it) sort the columns by norm.  Every swap is a
    // rotation (the moved column changes sign), so V stays in SO(3) and F = B V^T.
    if (fmin(n0, fmin(n1, n2)) <= 1e-28 * fmax(n0, fmax(n1, n2))) {
#define ADMM_SWAPNEG3(x, y) { _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) { const double t_ = x[i_]; x[i_] = y[i_]; y[i_] = -t_; } }
        if (n0 < n1) { ADMM_SWAPNEG3(b0, b1); ADMM_SWAPNEG3(v0, v1); const double t = n0; n0 = n1; n1 = t; }
        if (n0 < n2) { ADMM_SWAPNEG3(b0, b2); ADMM_SWAPNEG3(v0, v2); const double t = n0; n0 = n2; n2 = t; }
        if (n1 < n2) { ADMM_SWAPNEG3(b1, b2); ADMM_SWAPNEG3(v1, v2); const double t = n1; n1 = n2; n2 = t; }
#undef ADMM_SWAPNEG3
    }
    double u0[3], u1[3], u2[3];
    double s0 = 0.0;
    if (n0 > 1e-300) {
        const double inv = fast_rsqrt(n0);
        s0 = n0 * inv;
#pragma unroll
        for (int i = 0; i < 3; ++i) u0[i] = b0[i] * inv;
    } else { u0[0] = 1.0; u0[1] = 0.0; u0[2] = 0.0; }
    const double pr = dot3(u0, b1);
    double w[3] = {fma(-pr, u0[0], b1[0]), fma(-pr, u0[1], b1[1]), fma(-pr, u0[2], b1[2])};
    double wn = dot3(w, w);
    if (!(wn > 1e-30 * n0) || !(wn > 1e-300)) {
        // rank <= 1: any unit vector orthogonal to u0
        double ee[3] = {0.0, 0.0, 0.0};
        const double a0 = fabs(u0[0]), a1 = fabs(u0[1]), a2 = fabs(u0[2]);
        if (a0 <= a1 && a0 <= a2) ee[0] = 1.0; else if (a1 <= a2) ee[1] = 1.0; else ee[2] = 1.0;
        cross3(u0, ee, w);
        wn = dot3(w, w);
    }
    {
        const double inv = fast_rsqrt(wn);
#pragma unroll
        for (int i = 0; i < 3; ++i) u1[i] = w[i] * inv;
    }
    cross3(u0, u1, u2);
    double s1 = dot3(u1, b1), s2 = dot3(u2, b2);
    // the sign of det F sits on s2; the convention wants it on the smallest stretch: flip u2 and that column of U
    const bool f0 = s2 < 0.0 && s0 < -s2 && s0 <= s1, f1 = s2 < 0.0 && s1 < -s2 && !f0;
    const double g0 = f0 ? -1.0 : 1.0, g1 = f1 ? -1.0 : 1.0, g2 = (f0 || f1) ? -1.0 : 1.0;
    S[0] = s0 * g0; S[1] = s1 * g1; S[2] = s2 * g2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        U[i] = u0[i] * g0; U[3 + i] = u1[i] * g1; U[6 + i] = u2[i] * g2;
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
// KIND 3: co-rotated   Psi = mu sum (s_i-1)^2 + lambda/2 (sum s_i - 3)^2   (SplineTet with xu::CoRotated, kappa = 0)
// objective = Psi(s) + k/2 |s - x0|^2                                      (:184-192, :210-218)
// The model is templated on the scalar type: the same Newton runs first in FP32 (cheap iterations that
// get within ~1e-6 of the minimiser) and then in FP64 (one or two polishing iterations).  Newton is
// self-correcting, so the FP64 result does not depend on the FP32 phase beyond its starting point.
__device__ __forceinline__ float t_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double t_rcp(double x) { return fast_rcp(x); }
__device__ __forceinline__ float t_log(float x) { return __logf(x); }
// ln(x) for finite normal x > 0 (the callers pass J = s0 s1 s2 with every s >= 1e-12).  x = m 2^e with
// m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh(z), z = (m - 1)/(m + 1), |z| <= 0.1716: ten odd-series terms give
// < 1 ulp of truncation error.  ~30 instructions against ~120 for the library log (the NH polish is VALU-bound).
__device__ __forceinline__ double fast_log(double x) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    int e = ((hi >> 20) & 0x7ff) - 1023;
    double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);   // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double z = (m - 1.0) * fast_rcp(m + 1.0);
    const double z2 = z * z;
    double p = 1.0 / 21.0;
    p = fma(p, z2, 1.0 / 19.0); p = fma(p, z2, 1.0 / 17.0); p = fma(p, z2, 1.0 / 15.0); p = fma(p, z2, 1.0 / 13.0);
    p = fma(p, z2, 1.0 / 11.0); p = fma(p, z2, 1.0 / 9.0); p = fma(p, z2, 1.0 / 7.0); p = fma(p, z2, 1.0 / 5.0);
    p = fma(p, z2, 1.0 / 3.0);
    const double lm = fma(z * z2, 2.0 * p, 2.0 * z);                  // 2 z (1 + z^2 p)
    return fma((double)e, 0.6931471805599453, lm);
}
#ifdef ADMM_LIBM_LOG
__device__ __forceinline__ double t_log(double x) { return log(x); }
#else
__device__ __forceinline__ double t_log(double x) { return fast_log(x); }
#endif
__device__ __forceinline__ float t_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float t_abs(float a) { return fabsf(a); }
__device__ __forceinline__ double t_abs(double a) { return fabs(a); }
__device__ __forceinline__ float t_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float t_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return fmin(a, b); }

template <int KIND, typename T>
struct StretchModel {
    T mu, la, k;
    T x0[3];

    // value, gradient and the Hessian split H = diag(D) + la * w w^T
    __device__ __forceinline__ T eval(const T *s, T *g, T *D, T *w) const {
        if (KIND == 1) {
            const T J = s[0] * s[1] * s[2];
            const T lJ = t_log(J);
            T f = T(0), q = T(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const T si = t_rcp(s[i]);
                const T d = s[i] - x0[i];
                w[i] = si;
                g[i] = t_fma(mu, s[i] - si, t_fma(la * lJ, si, k * d));
                D[i] = t_fma(mu, t_fma(si, si, T(1)), t_fma(-la * lJ, si * si, k));
                f = t_fma(s[i], s[i], f);
                q = t_fma(d, d, q);
            }
            return T(0.5) * mu * (f - T(3)) - mu * lJ + T(0.5) * la * lJ * lJ + T(0.5) * k * q;
        } else if (KIND == 3) {
            // co-rotated linear: Psi = mu sum (s_i - 1)^2 + la/2 (sum s_i - 3)^2  (xu::CoRotated, XuSpline.hpp:84-96,
            // with kappa = 0: sum f(s_i) + sum g(s_i s_j) collapses to this); quadratic, H = (2 mu + k) I + la 1 1^T
            const T tr = s[0] + s[1] + s[2] - T(3);
            T ee = T(0), q = T(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const T e = s[i] - T(1);
                const T d = s[i] - x0[i];
                w[i] = T(1);
                g[i] = t_fma(T(2) * mu, e, t_fma(la, tr, k * d));
                D[i] = t_fma(T(2), mu, k);
                ee = t_fma(e, e, ee);
                q = t_fma(d, d, q);
            }
            return mu * ee + T(0.5) * la * tr * tr + T(0.5) * k * q;
        } else {
            const T ss = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
            const T trE = T(0.5) * (ss - T(3));
            T ee = T(0), q = T(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const T E = T(0.5) * (s[i] * s[i] - T(1));
                const T d = s[i] - x0[i];
                w[i] = s[i];
                g[i] = t_fma(mu * s[i], s[i] * s[i] - T(1), t_fma(la * trE, s[i], k * d));
                D[i] = t_fma(mu, T(3) * s[i] * s[i] - T(1), t_fma(la, trE, k));
                ee = t_fma(E, E, ee);
                q = t_fma(d, d, q);
            }
            return mu * ee + T(0.5) * la * trE * trE + T(0.5) * k * q;
        }
    }
    // NH needs s > 0 (log barrier); StVK / co-rotated accept s >= 0 (value() returns FLT_MAX only for s < 0)
    __device__ __forceinline__ bool feasible(const T *s) const {
        if (KIND == 1) return s[0] > T(0) && s[1] > T(0) && s[2] > T(0);
        return s[0] >= T(0) && s[1] >= T(0) && s[2] >= T(0);
    }
};

// Newton direction at (s, g, D, w): Sherman-Morrison solve with |D| floored, frozen components on the s = 0 boundary (StVK),
// scaled steepest descent if the rank-one part makes it no descent direction.  Returns false at a zero (reduced) gradient.
// pure = the exact Newton step (no floored / flipped curvature, no frozen component).
template <int KIND, typename T>
__device__ __forceinline__ bool newton_direction(const StretchModel<KIND, T> &m, const T *s, const T *g, const T *D, const T *w,
                                                 T *d, T &gd, bool &pure) {
    const T floorD = T(1e-8) * (t_abs(m.k) + t_abs(m.mu)) + T(1e-30);
    // H = diag(D) + la w w^T.  Every D_i > 0: positive definite.  Exactly ONE D_i < 0: still positive definite iff
    // 1 + la sum w_i^2 / D_i < 0 (det H = prod D_i (1 + la sum w_i^2 / D_i); Cauchy interlacing settles the other minors) --
    // the case of a strongly compressed stretch next to strongly extended ones (NH: log J > 0 turns a diagonal entry negative
    // while the rank-one barrier term keeps the Hessian convex).  In both cases the Sherman-Morrison solve with the SIGNED
    // diagonal is the exact Newton step.  Otherwise |D| is floored: a convexified model, linear convergence (it used to
    // serve the second case too, and elements at stretches like (4.9, 2.9, 2.4) then stopped at the iteration cap 5e-5 short
    // of the minimiser).
    T a[3], y[3];
    int nneg = 0;
    bool okD = true, act[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        act[i] = (KIND >= 2) && (s[i] <= T(0)) && (g[i] > T(0));
        okD = okD && !act[i] && (t_abs(D[i]) >= floorD);
        nneg += (D[i] < T(0)) ? 1 : 0;
    }
    T wDw = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = t_rcp(okD ? D[i] : T(1));
        wDw = t_fma(w[i] * w[i], a[i], wDw);
    }
    T den = t_fma(m.la, wDw, T(1));
    pure = okD && (nneg == 0 ? den > T(1e-6) : (nneg == 1 && den < T(-1e-6)));
    if (!pure) {
        wDw = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] = act[i] ? T(0) : t_rcp(t_max(t_abs(D[i]), floorD));
            wDw = t_fma(w[i] * w[i], a[i], wDw);
        }
        den = t_fma(m.la, wDw, T(1));
    }
    T wDg = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        y[i] = g[i] * a[i];
        wDg = t_fma(w[i], y[i], wDg);
    }
    const T coef = (pure || den > T(1e-6)) ? m.la * wDg * t_rcp(den) : T(0);
    gd = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = -(y[i] - coef * w[i] * a[i]);
        gd = t_fma(g[i], d[i], gd);
    }
    if (!(gd < T(0))) { // not a descent direction (indefinite rank-one part): scaled steepest descent
        gd = T(0); pure = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) { d[i] = -y[i]; gd = t_fma(g[i], d[i], gd); }
        if (!(gd < T(0))) return false;
    }
    return true;
}
// The lean version for the common case: the EXACT Newton step, or false if the Hessian is not positive definite by the test
// above / a bound is active / the step is no descent direction -- the caller then leaves the element to the general loop.
template <int KIND, typename T>
__device__ __forceinline__ bool newton_direction_exact(const StretchModel<KIND, T> &m, const T *s, const T *g, const T *D, const T *w, T *d) {
    const T floorD = T(1e-8) * (t_abs(m.k) + t_abs(m.mu)) + T(1e-30);
    T a[3];
    int nneg = 0;
    bool okD = true;
    T wDw = T(0), wDg = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        okD = okD && !((KIND >= 2) && (s[i] <= T(0)) && (g[i] > T(0))) && (t_abs(D[i]) >= floorD);
        nneg += (D[i] < T(0)) ? 1 : 0;
        a[i] = t_rcp(D[i]);
        wDw = t_fma(w[i] * w[i], a[i], wDw);
        wDg = t_fma(w[i], g[i] * a[i], wDg);
    }
    const T den = t_fma(m.la, wDw, T(1));
    const bool pd = okD && (nneg == 0 ? den > T(1e-6) : (nneg == 1 && den < T(-1e-6)));
    const T coef = m.la * wDg * t_rcp(den);
    T gd = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = -(g[i] - coef * w[i]) * a[i];
        gd = t_fma(g[i], d[i], gd);
    }
    return pd && gd < T(0);
}
// Is the step d small enough to be applied without re-evaluation (quadratic convergence: the remaining error is ~ step^2)?
// That error is ~ step^2 times (third / second derivative) ~ step^2 / s_min near the barrier / the s = 0 boundary: the
// threshold scales with min(1, s_min)^2, never below the 1e-9 the FP64 callers tolerate; and the estimate only holds for
// an exact Newton step.
template <typename T>
__device__ __forceinline__ bool newton_step_is_final(const T *s, const T *d, bool pure, T tol_final) {
    T dmax = T(0), mag = T(1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { dmax = t_max(dmax, t_abs(d[i])); mag = t_max(mag, t_abs(s[i])); }
    const T smin = t_max(T(0), t_min(T(1), t_min(s[0], t_min(s[1], s[2]))));
    const T tol_floor = sizeof(T) == 4 ? tol_final * T(0.03) : T(1e-9);
    return dmax <= (pure ? t_max(tol_final * smin * smin, tol_floor) : tol_floor) * mag;
}

// (everything by value: an array whose address is passed to a real call lives in scratch memory for its whole life)
template <typename T> struct NewtonIO { T s[3], g[3], D[3], w[3], f; int it; };
template <int KIND, typename T>
__device__ __attribute__((noinline)) NewtonIO<T> newton_stretch_general(StretchModel<KIND, T> m, NewtonIO<T> io, int max_it, T tol_final, T noise, int max_ls);
// Safeguarded Newton for argmin_s Psi(s) + k/2 |s - x0|^2, started from s (in/out); returns iterations.
//  - Hessian = diag(D) + la w w^T  -> Sherman-Morrison solve, |D| floored to stay positive definite
//  - NH: the log barrier keeps iterates strictly positive.  StVK: the feasible set is s >= 0 and for
//    inverted elements the minimiser sits ON that boundary -> projected Newton (components at the bound
//    with an outward gradient are frozen, trial points are projected back)
//  - Armijo backtracking on the objective, evaluated once per trial point (value+gradient+Hessian)
//  - a predicted step below tol_final is applied without re-evaluation and ends the iteration
//  - the common case -- the FIRST step is already that final step in every lane of the wave (the callers start close) --
//    is straight-line code under a wave-uniform branch; the general loop follows only for waves that need it
template <int KIND, typename T>
__device__ __forceinline__ int newton_stretch(const StretchModel<KIND, T> &m, T *s, int max_it, T tol_final, T noise, int max_ls) {
    T g[3], D[3], w[3];
    T f = m.eval(s, g, D, w);
    {
        T d[3], sn[3];
        const bool exact = newton_direction_exact<KIND, T>(m, s, g, D, w, d);
        bool fin = exact && newton_step_is_final<T>(s, d, true, tol_final);
#pragma unroll
        for (int i = 0; i < 3; ++i) { sn[i] = s[i] + d[i]; if (KIND >= 2) sn[i] = t_max(sn[i], T(0)); }
        fin = fin && m.feasible(sn);
        if (__all(fin)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) s[i] = sn[i];
            return 1;
        }
    }
    NewtonIO<T> io;
#pragma unroll
    for (int i = 0; i < 3; ++i) { io.s[i] = s[i]; io.g[i] = g[i]; io.D[i] = D[i]; io.w[i] = w[i]; }
    io.f = f; io.it = 0;
    io = newton_stretch_general<KIND, T>(m, io, max_it, tol_final, noise, max_ls);
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = io.s[i];
    return io.it;
}
// the general loop, OUTLINED (noinline): it is the rare path (waves whose first step is not final in every lane) but it is the
// part with the most values alive -- inlined, it dictated the register allocation of the whole local-step kernel
template <int KIND, typename T>
__device__ __attribute__((noinline)) NewtonIO<T> newton_stretch_general(StretchModel<KIND, T> m, NewtonIO<T> io, int max_it, T tol_final, T noise, int max_ls) {
    T s[3], g[3], D[3], w[3], f = io.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { s[i] = io.s[i]; g[i] = io.g[i]; D[i] = io.D[i]; w[i] = io.w[i]; }
    const T fscale = T(4) * (t_abs(m.mu) + t_abs(m.la) + t_abs(m.k));
    int it = 0;
#pragma unroll 1
    for (; it < max_it; ++it) {
        T d[3], gd;
        bool pure;
        if (!newton_direction<KIND, T>(m, s, g, D, w, d, gd, pure)) break;
        if (newton_step_is_final<T>(s, d, pure, tol_final)) { // final correction: apply and stop
            T sn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { sn[i] = s[i] + d[i]; if (KIND >= 2) sn[i] = t_max(sn[i], T(0)); }
            if (m.feasible(sn)) {
#pragma unroll
                for (int i = 0; i < 3; ++i) s[i] = sn[i];
            }
            ++it;
            break;
        }
        T t = T(1), sn[3], gn[3], Dn[3], wn[3], fn = f;
        bool ok = false;
#pragma unroll 1
        for (int ls = 0; ls < max_ls; ++ls) {
            T gs = T(0); // g . (sn - s) along the projected path
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sn[i] = t_fma(t, d[i], s[i]);
                if (KIND >= 2) sn[i] = t_max(sn[i], T(0));
                gs = t_fma(g[i], sn[i] - s[i], gs);
            }
            if (m.feasible(sn)) {
                fn = m.eval(sn, gn, Dn, wn);
                // round-off in f is absolute (O(1) terms cancel to O(strain^2)): scale the allowance by the
                // magnitude of the terms, not by |f|
                if (fn <= f + T(1e-4) * gs + noise * (t_abs(f) + fscale)) { ok = true; break; }
            }
            t *= T(0.5);
        }
        if (!ok) break;
        f = fn;
#pragma unroll
        for (int i = 0; i < 3; ++i) { s[i] = sn[i]; g[i] = gn[i]; D[i] = Dn[i]; w[i] = wn[i]; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) io.s[i] = s[i];
    io.it = it;
    return io;
}

// mixed-precision driver: FP32 iterations, then FP64 polish.  Parameters are normalised by k so the
// FP32 phase works with O(1) coefficients.
//  - Start: for moderate strains (|x0 - 1| < 1/4) the exact Newton step from the REST state, whose gradient
//    and Hessian are known in closed form for both models (g = k (1 - x0), H = (2 mu + k) I + la 1 1^T), so
//    the first objective evaluation is saved and the start is O(strain^2) from the minimiser instead of
//    O(strain).  Larger strains start from the caller's s as before.
//  - FP32 Newton to a step of 3e-5, then FP64 Newton: its first step is ~1e-7 (FP32 rounding), below the
//    2e-6 threshold under which a step is applied without re-evaluation (remaining error ~ step^2).
template <int KIND>
__device__ __forceinline__ int minimize_stretch(double mu, double la, double k, const double *x0, double *s) {
    const double ik = fast_rcp(k);
    StretchModel<KIND, float> mf;
    mf.mu = (float)(mu * ik); mf.la = (float)(la * ik); mf.k = 1.0f;
    float sf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { mf.x0[i] = (float)x0[i]; sf[i] = (float)s[i]; }
    {
        const float v0 = mf.x0[0] - 1.0f, v1 = mf.x0[1] - 1.0f, v2 = mf.x0[2] - 1.0f;
        if (fmaxf(fabsf(v0), fmaxf(fabsf(v1), fabsf(v2))) < 0.25f) {
            const float D = fmaf(2.0f, mf.mu, 1.0f);
            const float iD = __frcp_rn(D);
            const float c = mf.la * (v0 + v1 + v2) * iD * __frcp_rn(fmaf(3.0f, mf.la, D));
            sf[0] = 1.0f + fmaf(v0, iD, -c); sf[1] = 1.0f + fmaf(v1, iD, -c); sf[2] = 1.0f + fmaf(v2, iD, -c);
        }
    }
    if (KIND == 1) { sf[0] = fmaxf(sf[0], 1e-12f); sf[1] = fmaxf(sf[1], 1e-12f); sf[2] = fmaxf(sf[2], 1e-12f); }
    int it = newton_stretch<KIND, float>(mf, sf, 10, 3e-5f, 1e-6f, 12);
    StretchModel<KIND, double> md;
    md.mu = mu * ik; md.la = la * ik; md.k = 1.0;
    bool fin = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) { md.x0[i] = x0[i]; fin = fin && (sf[i] == sf[i]) && (fabsf(sf[i]) < 1e30f); }
    if (fin && mf.feasible(sf)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s[i] = (double)sf[i];
    }
    it += newton_stretch<KIND, double>(md, s, 60, 2e-6, 4e-16, 50);
    return it;
}

// Prox in principal stretches.  S (in) = signed stretches of q = D_i x + u_i; S (out) = stretches of z.
// KIND 0, linear tet (src/TetEnergyTerm.cpp:73-92): z = (P + q)/2 with P = U V^T (signed factors)
//         == U diag((1 + S)/2) V^T.
// KIND 1/2/3, hyperelastic (src/TetEnergyTerm.cpp:114-136): minimise over the stretches.
// ---- xu:: splines WITH their compression term (SplineTet with a spline constructed with kappa != 0) --------------------
// Psi(s) = Psi_spline(s) + c(J),  J = s0 s1 s2,  c(J) = kappa/12 ((1 - J)/6)^3      (src/XuSpline.hpp:43-45, Eq. 16 of Xu et al.)
// c' = -kappa/24 ((1 - J)/6)^2 as the reference writes it; c'' = kappa/72 (1 - J)/6.  The Hessian of c(J) is
// c'' dJ dJ^T + c' d2J (d2J_ij = s_k off the diagonal): not diagonal-plus-rank-one next to the StVK / co-rotated terms, so
// this (rare) model runs a Newton with the dense 3 x 3 Hessian, FP64 only.  type: 0 xu::NeoHookean, 1 xu::StVK, 2 xu::CoRotated.
struct SplineKappaModel {
    int type;
    double mu, la, k, kappa, x0[3];
    __device__ __forceinline__ double lower() const { return 0.0; }
    __device__ __forceinline__ bool feasible(const double *s) const {
        if (type == 0) return s[0] > 0.0 && s[1] > 0.0 && s[2] > 0.0;
        return s[0] >= 0.0 && s[1] >= 0.0 && s[2] >= 0.0;
    }
    // value, gradient, Hessian H = {h00, h01, h02, h11, h12, h22}
    __device__ __forceinline__ double eval(const double *s, double *g, double *H) const {
        double D[3], w[3], f;
        if (type == 0) { StretchModel<1, double> m; m.mu = mu; m.la = la; m.k = k; m.x0[0] = x0[0]; m.x0[1] = x0[1]; m.x0[2] = x0[2]; f = m.eval(s, g, D, w); }
        else if (type == 1) { StretchModel<2, double> m; m.mu = mu; m.la = la; m.k = k; m.x0[0] = x0[0]; m.x0[1] = x0[1]; m.x0[2] = x0[2]; f = m.eval(s, g, D, w); }
        else { StretchModel<3, double> m; m.mu = mu; m.la = la; m.k = k; m.x0[0] = x0[0]; m.x0[1] = x0[1]; m.x0[2] = x0[2]; f = m.eval(s, g, D, w); }
        H[0] = fma(la * w[0], w[0], D[0]); H[1] = la * w[0] * w[1]; H[2] = la * w[0] * w[2];
        H[3] = fma(la * w[1], w[1], D[1]); H[4] = la * w[1] * w[2]; H[5] = fma(la * w[2], w[2], D[2]);
        const double J = s[0] * s[1] * s[2], t = (1.0 - J) * (1.0 / 6.0);
        const double c0 = kappa * (1.0 / 12.0) * t * t * t, c1 = -kappa * (1.0 / 24.0) * t * t, c2 = kappa * (1.0 / 72.0) * t;
        const double dJ[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = fma(c1, dJ[i], g[i]);
        H[0] = fma(c2 * dJ[0], dJ[0], H[0]); H[3] = fma(c2 * dJ[1], dJ[1], H[3]); H[5] = fma(c2 * dJ[2], dJ[2], H[5]);
        H[1] += fma(c2 * dJ[0], dJ[1], c1 * s[2]); H[2] += fma(c2 * dJ[0], dJ[2], c1 * s[1]); H[4] += fma(c2 * dJ[1], dJ[2], c1 * s[0]);
        return f + c0;
    }
};
// argmin_s Psi(s) + c(J) + k/2 |s - x0|^2: projected, safeguarded Newton with the dense Hessian (Cholesky of the free block,
// scaled steepest descent when it is not positive definite), Armijo backtracking; iterated until the step is below 1e-9.
template <class MODEL>
__device__ __forceinline__ int newton_stretch_dense(const MODEL &m, double *s, int max_it) {
    double g[3], H[6];
    double f = m.eval(s, g, H);
    const double fscale = 4.0 * (fabs(m.mu) + fabs(m.la) + fabs(m.k));
    int it = 0;
#pragma unroll 1
    for (; it < max_it; ++it) {
        bool fr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) fr[i] = !(m.type != 0 && s[i] <= m.lower() && g[i] > 0.0);   // components held at the bound
        // frozen components: unit row / column, zero right-hand side
        const double a00 = fr[0] ? H[0] : 1.0, a11 = fr[1] ? H[3] : 1.0, a22 = fr[2] ? H[5] : 1.0;
        const double a01 = (fr[0] && fr[1]) ? H[1] : 0.0, a02 = (fr[0] && fr[2]) ? H[2] : 0.0, a12 = (fr[1] && fr[2]) ? H[4] : 0.0;
        const double r0 = fr[0] ? -g[0] : 0.0, r1 = fr[1] ? -g[1] : 0.0, r2 = fr[2] ? -g[2] : 0.0;
        double d[3];
        bool newton = false;
        const double floorD = 1e-8 * (fabs(m.k) + fabs(m.mu)) + 1e-30;
        if (a00 > floorD) {   // LDL^T of the 3 x 3
            const double l10 = a01 / a00, l20 = a02 / a00;
            const double d1 = a11 - l10 * a01;
            if (d1 > floorD) {
                const double l21 = (a12 - l20 * a01) / d1;
                const double d2 = a22 - l20 * a02 - l21 * l21 * d1;
                if (d2 > floorD) {
                    const double y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
                    d[2] = y2 / d2;
                    d[1] = y1 / d1 - l21 * d[2];
                    d[0] = y0 / a00 - l10 * d[1] - l20 * d[2];
                    newton = true;
                }
            }
        }
        double gd = 0.0;
        if (newton) { gd = g[0] * d[0] + g[1] * d[1] + g[2] * d[2]; newton = gd < 0.0; }
        if (!newton) {   // scaled steepest descent on the free components
            d[0] = r0 / fmax(fabs(a00), floorD); d[1] = r1 / fmax(fabs(a11), floorD); d[2] = r2 / fmax(fabs(a22), floorD);
            gd = g[0] * d[0] + g[1] * d[1] + g[2] * d[2];
            if (!(gd < 0.0)) break;   // zero (reduced) gradient
        }
        const double dmax = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
        const double mag = fmax(1.0, fmax(fabs(s[0]), fmax(fabs(s[1]), fabs(s[2]))));
        if (dmax <= 1e-9 * mag) {   // final correction: apply and stop
            double sn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { sn[i] = s[i] + d[i]; if (m.type != 0) sn[i] = fmax(sn[i], m.lower()); }
            if (m.feasible(sn)) { s[0] = sn[0]; s[1] = sn[1]; s[2] = sn[2]; }
            ++it;
            break;
        }
        double t = 1.0, sn[3], gn[3], Hn[6], fn = f;
        bool ok = false;
#pragma unroll 1
        for (int ls = 0; ls < 50; ++ls) {
            double gs = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sn[i] = fma(t, d[i], s[i]);
                if (m.type != 0) sn[i] = fmax(sn[i], m.lower());
                gs = fma(g[i], sn[i] - s[i], gs);
            }
            if (m.feasible(sn)) {
                fn = m.eval(sn, gn, Hn);
                if (fn <= f + 1e-4 * gs + 4e-16 * (fabs(f) + fscale)) { ok = true; break; }
            }
            t *= 0.5;
        }
        if (!ok) break;
        f = fn;
#pragma unroll
        for (int i = 0; i < 3; ++i) { s[i] = sn[i]; g[i] = gn[i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) H[i] = Hn[i];
    }
    return it;
}
// ---- USER-DEFINED xu::Spline (src/XuSpline.hpp:34-46: any object with f, g, h, df, dg, dh; src/TetEnergyTerm.hpp:197-204) ----
// The host samples the six functions once (admm_host_tabulate_spline) and the device evaluates the TABLES.  One table per
// function F in {f, g, h}: kSplineNodes nodes, uniform in t = ln x over [x_lo, x_hi] (stretches are positive and the relative
// resolution is what matters: f lives on stretches, g on products of two, h on J), per node (F, dF/dt, d2F/dt2) with dF/dt =
// x F'(x) from the spline's own derivative and the second derivative from central differences of it.  Between nodes: the
// QUINTIC Hermite interpolant -- C2, so value, gradient and Hessian of the interpolated energy are derivatives of ONE function
// and the safeguarded Newton iteration above converges on it exactly as on a closed form.  Interpolation error of the gradient:
// O(dt^5) ~ 1e-10 relative at 1024 nodes.  Outside the table the end node's Taylor quadratic in x continues the function (finite
// down to x = 0: the polynomial splines are reproduced there to O(x_lo^3); a spline that is singular at 0 is not, and does not
// need to be -- its minimiser stays away from 0).
constexpr int kSplineNodes = 1024;
constexpr int kSplineFnDoubles = 4 + 3 * kSplineNodes;          // {t0, dt, 1 / dt, n} + nodes
constexpr int kSplineTableDoubles = 3 * kSplineFnDoubles;       // f, g, h
// F(x), F'(x), F''(x) from one function's table
__device__ __forceinline__ void spline_table_eval(const double *tab, double x, double &F, double &F1, double &F2) {
    const double t0 = tab[0], dt = tab[1], idt = tab[2];
    const int n = (int)tab[3];
    const double t = t_log(fmax(x, 1e-300));
    double r = (t - t0) * idt;
    int i = (int)floor(r);
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    double u = r - (double)i;
    const double *a = tab + 4 + 3 * i;
    double p, pt, ptt;
    if (u < 0.0 || u > 1.0) {      // outside the table: the Taylor quadratic IN x of the nearest end node (finite down to x = 0)
        const double *e = u < 0.0 ? a : a + 3;
        const double xe = exp(u < 0.0 ? t0 : fma(dt, (double)(n - 1), t0)), ixe = 1.0 / xe;
        const double d1 = e[1] * ixe, d2 = (e[2] - e[1]) * ixe * ixe, dx = x - xe;
        F = fma(dx, fma(0.5 * dx, d2, d1), e[0]); F1 = fma(dx, d2, d1); F2 = d2;
        return;
    } else {
        const double F0 = a[0], G0 = a[1] * dt, H0 = a[2] * dt * dt, F1n = a[3], G1 = a[4] * dt, H1 = a[5] * dt * dt;
        // quintic Hermite: p(u) = c0 + c1 u + c2 u^2 + c3 u^3 + c4 u^4 + c5 u^5
        const double c0 = F0, c1 = G0, c2 = 0.5 * H0;
        const double dF = F1n - F0;
        const double c3 = 10.0 * dF - 6.0 * G0 - 4.0 * G1 - 1.5 * H0 + 0.5 * H1;
        const double c4 = -15.0 * dF + 8.0 * G0 + 7.0 * G1 + 1.5 * H0 - H1;
        const double c5 = 6.0 * dF - 3.0 * (G0 + G1) - 0.5 * H0 + 0.5 * H1;
        p = fma(u, fma(u, fma(u, fma(u, fma(u, c5, c4), c3), c2), c1), c0);
        pt = fma(u, fma(u, fma(u, fma(u, 5.0 * c5, 4.0 * c4), 3.0 * c3), 2.0 * c2), c1) * idt;
        ptt = fma(u, fma(u, fma(u, 20.0 * c5, 12.0 * c4), 6.0 * c3), 2.0 * c2) * idt * idt;
    }
    const double ix = fast_rcp(fmax(x, 1e-300));
    F = p; F1 = pt * ix; F2 = (ptt - pt) * ix * ix;
}
struct SplineTableModel {
    int type;                 // 1: stretches are kept at or above the table's lower end (projected steps)
    const double *tab;        // [3][kSplineFnDoubles]
    double mu, la, k, x0[3], lo;
    __device__ __forceinline__ double lower() const { return lo; }
    __device__ __forceinline__ bool feasible(const double *s) const { return s[0] >= lo && s[1] >= lo && s[2] >= lo; }
    // Psi = sum f(s_i) + sum_{i<j} g(s_i s_j) + h(s_0 s_1 s_2) + k/2 |s - x0|^2 (src/TetEnergyTerm.cpp:243-265), dense Hessian
    __device__ __forceinline__ double eval(const double *s, double *g, double *H) const {
        double val = 0.0;
        const double *tf = tab, *tg = tab + kSplineFnDoubles, *th = tab + 2 * kSplineFnDoubles;
        double f0, f1, f2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            spline_table_eval(tf, s[i], f0, f1, f2);
            const double dx = s[i] - x0[i];
            val += f0 + 0.5 * k * dx * dx;
            g[i] = fma(k, dx, f1);
            H[i == 0 ? 0 : i == 1 ? 3 : 5] = f2 + k;
        }
        H[1] = 0.0; H[2] = 0.0; H[4] = 0.0;
        // pairs (0,1), (0,2), (1,2)
        const int pi[3] = {0, 0, 1}, pj[3] = {1, 2, 2}, ph[3] = {1, 2, 4};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i = pi[q], j = pj[q];
            spline_table_eval(tg, s[i] * s[j], f0, f1, f2);
            val += f0;
            g[i] = fma(f1, s[j], g[i]); g[j] = fma(f1, s[i], g[j]);
            H[i == 0 ? 0 : 3] = fma(f2 * s[j], s[j], H[i == 0 ? 0 : 3]);
            H[j == 1 ? 3 : 5] = fma(f2 * s[i], s[i], H[j == 1 ? 3 : 5]);
            H[ph[q]] = fma(f2 * s[i], s[j], H[ph[q]]) + f1;
        }
        const double J = s[0] * s[1] * s[2];
        spline_table_eval(th, J, f0, f1, f2);
        val += f0;
        const double dJ[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = fma(f1, dJ[i], g[i]);
        H[0] = fma(f2 * dJ[0], dJ[0], H[0]); H[3] = fma(f2 * dJ[1], dJ[1], H[3]); H[5] = fma(f2 * dJ[2], dJ[2], H[5]);
        H[1] += fma(f2 * dJ[0], dJ[1], f1 * s[2]); H[2] += fma(f2 * dJ[0], dJ[2], f1 * s[1]); H[4] += fma(f2 * dJ[1], dJ[2], f1 * s[0]);
        return val;
    }
};
// HyperElasticTet::prox on the stretches (src/TetEnergyTerm.cpp:124-135) for a tabulated (user-defined) spline.  k_scale = a
// stiffness of the order of the spline's own (the tet's k) for the solver's round-off thresholds.
__device__ __forceinline__ void prox_stretches_table(const double *tab, double k, double *S) {
    SplineTableModel m;
    m.type = 1; m.tab = tab; m.k = k; m.mu = k; m.la = k;
    m.lo = 0.0;                                           // stretches stay >= 0 (the reference's objective is infinite below, src/TetEnergyTerm.cpp:211-214)
    m.x0[0] = S[0]; m.x0[1] = S[1]; m.x0[2] = S[2];       // :124 set_x0 (before the fix-ups)
    const double eps = 1e-6;
    if (fabs(S[0]) < eps && fabs(S[1]) < eps && fabs(S[2]) < eps) { S[0] = eps; S[1] = eps; S[2] = eps; } // :128-131
    S[0] = fabs(S[0]); S[1] = fabs(S[1]); S[2] = fabs(S[2]);   // :133
    newton_stretch_dense(m, S, 200);
}
// HyperElasticTet::prox on the stretches (src/TetEnergyTerm.cpp:124-135) for a spline with kappa != 0
__device__ __forceinline__ void prox_stretches_kappa(int type, double mu, double la, double k, double kappa, double *S) {
    SplineKappaModel m;
    const double ik = fast_rcp(k);
    m.type = type; m.mu = mu * ik; m.la = la * ik; m.k = 1.0; m.kappa = kappa * ik;
    m.x0[0] = S[0]; m.x0[1] = S[1]; m.x0[2] = S[2];       // :124 set_x0 (before the fix-ups)
    const double eps = 1e-6;
    if (fabs(S[0]) < eps && fabs(S[1]) < eps && fabs(S[2]) < eps) { S[0] = eps; S[1] = eps; S[2] = eps; } // :128-131
    S[0] = fabs(S[0]); S[1] = fabs(S[1]); S[2] = fabs(S[2]);   // :133
    if (type == 0) { S[0] = fmax(S[0], 1e-12); S[1] = fmax(S[1], 1e-12); S[2] = fmax(S[2], 1e-12); }
    newton_stretch_dense(m, S, 200);
}

// ---- STABLE NEO-HOOKEAN (the reference's README lists it as a TODO, README.md:23-28; no reference code: "parity unpinned") -------------
// Smith, de Goes, Kim, "Stable Neo-Hookean Flesh Simulation", ACM TOG 37(2), 2018, Eq. 14 with the re-parametrisation of their section
// 3.4 so that the model meets linear elasticity at the rest state with the TET's Lame constants:
//     Psi(F) = mu_s / 2 (I_C - 3) + la_s / 2 (J - alpha)^2 - mu_s / 2 log(I_C + 1),
//     mu_s = 4/3 mu,  la_s = lambda + 5/6 mu,  alpha = 1 + mu_s / la_s - mu_s / (4 la_s) = 1 + 3 mu_s / (4 la_s),
// I_C = |F|_F^2 = sum s_i^2, J = det F = s0 s1 s2: rotation invariant, so the prox of HyperElasticTet::prox (src/TetEnergyTerm.cpp:114-136)
// is a minimisation over the signed stretches like for the other models -- finite and smooth for inverted and degenerate elements (no
// log J barrier), which is the point of the model.  Gradient and Hessian in the stretches (dJ_i = prod_{j != i} s_j, q = 1 / (I_C + 1)):
//     g_i  = mu_s s_i (1 - q) + la_s (J - alpha) dJ_i + k (s_i - x0_i)
//     H_ii = mu_s (1 - q) + 2 mu_s q^2 s_i^2 + la_s dJ_i^2 + k
//     H_ij = 2 mu_s q^2 s_i s_j + la_s (dJ_i dJ_j + (J - alpha) s_k)
// minimised by the dense-Hessian safeguarded Newton above (no bound on the stretches: type 0, always feasible).
struct StableNHModel {
    int type;                 // 0: no projected steps (newton_stretch_dense)
    double mu, la, k, alpha, x0[3];      // mu, la = mu_s, la_s (already divided by k when the caller scales the problem)
    __device__ __forceinline__ double lower() const { return -1.7976931348623157e308; }
    __device__ __forceinline__ bool feasible(const double *) const { return true; }
    __device__ __forceinline__ double eval(const double *s, double *g, double *H) const {
        const double IC = fma(s[0], s[0], fma(s[1], s[1], s[2] * s[2])), J = s[0] * s[1] * s[2], q = 1.0 / (IC + 1.0);
        const double dJ[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
        const double Ja = J - alpha, a1 = mu * (1.0 - q), a2 = 2.0 * mu * q * q;
        double f = 0.5 * mu * (IC - 3.0) + 0.5 * la * Ja * Ja - 0.5 * mu * log(IC + 1.0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double dx = s[i] - x0[i];
            f = fma(0.5 * k * dx, dx, f);
            g[i] = fma(a1, s[i], fma(la * Ja, dJ[i], k * dx));
        }
        H[0] = a1 + a2 * s[0] * s[0] + la * dJ[0] * dJ[0] + k;
        H[3] = a1 + a2 * s[1] * s[1] + la * dJ[1] * dJ[1] + k;
        H[5] = a1 + a2 * s[2] * s[2] + la * dJ[2] * dJ[2] + k;
        H[1] = a2 * s[0] * s[1] + la * (dJ[0] * dJ[1] + Ja * s[2]);
        H[2] = a2 * s[0] * s[2] + la * (dJ[0] * dJ[2] + Ja * s[1]);
        H[4] = a2 * s[1] * s[2] + la * (dJ[1] * dJ[2] + Ja * s[0]);
        return f;
    }
};
// HyperElasticTet::prox on the stretches (src/TetEnergyTerm.cpp:124-135) for the stable Neo-Hookean model; mu, la = the tet's Lame constants
__device__ __forceinline__ void prox_stretches_stable_nh(double mu, double la, double k, double *S) {
    StableNHModel m;
    const double ik = fast_rcp(k), mus = (4.0 / 3.0) * mu, las = la + (5.0 / 6.0) * mu;
    m.type = 0; m.mu = mus * ik; m.la = las * ik; m.k = 1.0; m.alpha = 1.0 + 0.75 * mus / las;
    m.x0[0] = S[0]; m.x0[1] = S[1]; m.x0[2] = S[2];       // :124 set_x0 (before the fix-ups)
    const double eps = 1e-6;
    if (fabs(S[0]) < eps && fabs(S[1]) < eps && fabs(S[2]) < eps) { S[0] = eps; S[1] = eps; S[2] = eps; } // :128-131
    S[0] = fabs(S[0]); S[1] = fabs(S[1]); S[2] = fabs(S[2]);   // :133 (the start only: the minimiser itself may cross zero)
    newton_stretch_dense(m, S, 200);
}

template <int KIND>
__device__ __forceinline__ void prox_stretches(double mu, double la, double k, double *S) {
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) S[i] = 0.5 * (1.0 + S[i]);
        return;
    }
    double x0[3] = {S[0], S[1], S[2]};                      // :124 set_x0 (before the fix-ups)
    const double eps = 1e-6;
    if (fabs(S[0]) < eps && fabs(S[1]) < eps && fabs(S[2]) < eps) { S[0] = eps; S[1] = eps; S[2] = eps; } // :128-131
    S[0] = fabs(S[0]); S[1] = fabs(S[1]); S[2] = fabs(S[2]);   // :133 (flips the signed smallest stretch, wherever signed_svd3 left it)
    if (KIND == 1) { // keep the start strictly inside the log barrier
        S[0] = fmax(S[0], 1e-12); S[1] = fmax(S[1], 1e-12); S[2] = fmax(S[2], 1e-12);
    }
    minimize_stretch<(KIND == 0 ? 1 : KIND)>(mu, la, k, x0, S);
}

// src/TriEnergyTerm.cpp:73-101.  q = 3x2 column-major (6 doubles) -> z
__device__ __forceinline__ void prox_tri(const double *q, double lmin, double lmax, double *z) {
    // C = F^T F (2x2); one exact Jacobi rotation diagonalises it
    double c00 = dot3(q, q), c01 = dot3(q, q + 3), c11 = dot3(q + 3, q + 3);
    double cs = 1.0, sn = 0.0;
    if (fabs(c01) > 1e-17 * (c00 + c11)) {
        const double a = c11 - c00, b = 2.0 * c01;
        const double h2 = fma(a, a, b * b);
        const double t = copysign(1.0, a) * b * fast_rcp(fabs(a) + h2 * fast_rsqrt(h2));
        cs = fast_rsqrt(fma(t, t, 1.0));
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
        const double inv = fast_rsqrt(n0);
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
        const double inv = fast_rsqrt(wn);
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
