"""CPU oracle driver: the reference's Solver::initialize / Solver::step restated in numpy/scipy on top
of the C restatement (admm_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (admm-elastic_amd/) never imports this module.

Reference lines restated (relative to the reference repo root):
  initialize : src/Solver.cpp:167-261 (D, W, A = M + dt^2 D^T W^2 D, pins -> SpringPin terms)
  step       : src/Solver.cpp:35-110
  reductions : src/TetEnergyTerm.cpp:50-71, src/TriEnergyTerm.cpp:54-69, src/SpringEnergyTerm.hpp:54-59
  LDLT       : src/LinearSolver.hpp:79-90 (scipy SuperLU direct solve stands in for Eigen SimplicialLDLT;
               tests/test_oracle_vs_ref.py checks it against the real SimplicialLDLT in oracle/_ref)
  UzawaCG    : src/UzawaCG.hpp:57-125
  GS         : src/NodalMultiColorGS.hpp:60-146 (C), colours supplied by the caller
  collisions : src/Collider.hpp:152-212, src/ConstraintSet.hpp:59-116, src/PassiveObject.hpp:32-64
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "_build", "liboracle.so")
_REF = os.path.join(HERE, "_ref", "libadmm_ref.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


def build(force=False):
    """gcc the C restatement (and, when /root/reference exists, the real-reference driver)."""
    src = os.path.join(HERE, "admm_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src) or \
            (os.path.exists("/root/reference/src/TriEnergyTerm.cpp") and not os.path.exists(_REF)):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return _LIB


_lib = None


def _p(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_svd3xn.argtypes = [C.c_int, dp, dp, dp, dp]
        L.orc_signed_svd3.argtypes = [dp, dp, dp, dp]
        L.orc_prox_tet_linear.argtypes = [dp]
        L.orc_energy_tet_linear.argtypes = [dp, C.c_double, C.c_double]
        L.orc_energy_tet_linear.restype = C.c_double
        L.orc_prox_tet_hyper.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp, C.c_int]
        L.orc_prox_tet_hyper.restype = C.c_int
        L.orc_prox_value.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp, dp]
        L.orc_prox_value.restype = C.c_double
        L.orc_prox_gradient.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp, dp, dp]
        L.orc_prox_tri.argtypes = [dp, C.c_double, C.c_double]
        L.orc_tet_rest.argtypes = [dp, dp, dp, dp, dp, dp]
        L.orc_tet_rest.restype = C.c_int
        L.orc_tri_rest.argtypes = [dp, dp, dp, dp, dp]
        L.orc_tri_rest.restype = C.c_int
        L.orc_local_tets.argtypes = [C.c_int, ip, dp, ip, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_local_tets_k.argtypes = [C.c_int, ip, dp, ip, dp, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_prox_tet_hyper_k.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, C.c_int]
        L.orc_prox_value_k.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
        L.orc_prox_value_k.restype = C.c_double
        L.orc_prox_gradient_k.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp]
        L.orc_local_tris.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp, dp]
        L.orc_local_pins.argtypes = [C.c_int, ip, dp, ip, dp, dp, dp]
        L.orc_csr_matvec.argtypes = [C.c_int, ip, ip, dp, dp, dp]
        L.orc_gs_solve.argtypes = [C.c_int, ip, ip, dp, dp, dp, C.c_int, ip, ip, ip, dp, C.c_int, ip, dp,
                                   C.c_double, C.c_int, C.c_double]
        L.orc_gs_solve.restype = C.c_int
        L.orc_gs_solve_full.argtypes = [C.c_int, ip, ip, dp, dp, dp, C.c_int, ip, ip, ip, dp, C.c_int, ip, dp,
                                        C.c_double, C.c_int, C.c_double]
        L.orc_gs_solve_full.restype = C.c_int
        L.orc_gs_set_pin_normals.argtypes = [dp]
        L.orc_detect_dynamic.argtypes = [C.c_int, ip, dp, C.c_int, dp, C.c_int, ip, C.c_int, ip, ip, ip, dp, dp, dp]
        _lib = L
    return _lib


def ref_lib():
    """The real-reference pieces (oracle/_ref/libadmm_ref.so) or None when not built.  Only used where the reference tree itself
    exists (the build container): on a GPU box the golden vectors made from it (tests/golden/ref_vectors.npz) stand in, and a
    prebuilt library that travelled with a snapshot is never mapped (ADMM_ORACLE_FORCE_REF=1 overrides, for a calibration run)."""
    if not os.path.isdir("/root/reference") and os.environ.get("ADMM_ORACLE_FORCE_REF") != "1":
        return None
    if not os.path.exists(_REF):
        try:
            build()
        except Exception:
            pass
    if not os.path.exists(_REF):
        return None
    L = C.CDLL(_REF)
    L.ref_signed_svd.argtypes = [dp, dp, dp, dp]
    L.ref_jacobi_svd3.argtypes = [dp, dp, dp, dp]
    L.ref_lame.argtypes = [C.c_double, C.c_double, dp, dp, dp]
    L.ref_tri_local_step.argtypes = [C.c_int, ip, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_double,
                                     dp, dp, dp, dp, ip, ip, dp]
    L.ref_tri_local_step.restype = C.c_int
    L.ref_pin_local_step.argtypes = [C.c_int, ip, dp, ip, C.c_int, dp, dp, dp]
    L.ref_pin_local_step.restype = C.c_double
    L.ref_floor_constraints.argtypes = [C.c_int, dp, C.c_double, C.c_double, C.c_int, ip, dp, dp]
    L.ref_floor_constraints.restype = C.c_int
    L.ref_ldlt_solve.argtypes = [C.c_int, ip, ip, dp, C.c_int, dp, dp]
    L.ref_ldlt_solve.restype = C.c_int
    L.ref_xu_spline.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp]
    if hasattr(L, "ref_time_tri_local_step"):      # (a prebuilt library of an earlier round has no timing entry points)
        L.ref_time_tri_local_step.argtypes = [C.c_int, ip, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, C.c_int]
        L.ref_time_tri_local_step.restype = C.c_double
        L.ref_time_ldlt.argtypes = [C.c_int, ip, ip, dp, dp, dp, C.c_int, dp, dp]
        L.ref_time_ldlt.restype = C.c_int
    return L


def xu_spline(which, mu, la, kappa, x):
    """The three splines the reference ships, src/XuSpline.hpp:48-96 (which = 0 NeoHookean, 1 StVK, 2 CoRotated), with
    the compression term of :44-45: returns (f, g, h, df, dg, dh) at x.  Pinned on the real header through
    tests/golden/ref_vectors.npz (tests/test_oracle_vs_ref.py)."""
    t = (1.0 - x) / 6.0
    comp, dcomp = (kappa / 12.0) * t ** 3, (-kappa / 24.0) * t ** 2
    if which == 0:
        lx = np.log(x)
        return (0.5 * mu * (x * x - 1.0), 0.0, -mu * lx + 0.5 * la * lx * lx + comp, mu * x, 0.0, -mu / x + la * lx / x + dcomp)
    x2 = x * x
    if which == 1:
        return (0.125 * la * (x2 * x2 - 6.0 * x2 + 5.0) + 0.25 * mu * (x2 - 1.0) ** 2, 0.25 * la * (x2 - 1.0), comp,
                0.125 * la * (4.0 * x2 * x - 12.0 * x) + mu * x * (x2 - 1.0), 0.5 * la * x, dcomp)
    return (0.5 * la * (x2 - 6.0 * x + 5.0) + mu * (x - 1.0) ** 2, la * (x - 1.0), comp,
            0.5 * la * (2.0 * x - 6.0) + 2.0 * mu * (x - 1.0), la, dcomp)


# ---- small wrappers -------------------------------------------------------------------------------
def lame(youngs, poisson):
    """src/EnergyTerm.hpp:48-52 -> (mu, lambda, bulk)."""
    mu = youngs / (2.0 * (1.0 + poisson))
    la = youngs * poisson / ((1.0 + poisson) * (1.0 - 2.0 * poisson))
    return mu, la, la + (2.0 / 3.0) * mu


def svd3(F):
    F = np.asfortranarray(np.asarray(F, dtype=np.float64).reshape(3, 3))
    a = np.ascontiguousarray(F.T).copy()  # column-major flat
    U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
    lib().orc_svd3xn(3, _p(a), _p(U), _p(S), _p(V))
    return U.reshape(3, 3).T, S, V.reshape(3, 3).T


def signed_svd3(F):
    a = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(3, 3).T).copy()
    U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
    lib().orc_signed_svd3(_p(a), _p(S), _p(U), _p(V))
    return U.reshape(3, 3).T, S, V.reshape(3, 3).T


def tet_rest(verts, tets):
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
    n = len(tets)
    Binv = np.zeros((n, 9)); vol = np.zeros(n)
    v = C.c_double()
    for t in range(n):
        q = [np.ascontiguousarray(verts[tets[t][i]]) for i in range(4)]
        b = np.zeros(9)
        rc = lib().orc_tet_rest(_p(q[0]), _p(q[1]), _p(q[2]), _p(q[3]), _p(b), C.byref(v))
        if rc:
            raise RuntimeError("**TetEnergyTerm Error: Inverted initial tet")
        Binv[t] = b; vol[t] = v.value
    return Binv, vol


def tet_rest_fast(verts, tets):
    """Vectorised numpy version of tet_rest (same arithmetic) for large meshes."""
    verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    v0 = verts[tets[:, 0]]
    B = np.stack([verts[tets[:, 1]] - v0, verts[tets[:, 2]] - v0, verts[tets[:, 3]] - v0], axis=2)  # [n,3,3], B[:,:,c]
    det = np.linalg.det(B)
    if np.any(det < 0):
        raise RuntimeError("**TetEnergyTerm Error: Inverted initial tet")
    Bi = np.linalg.inv(B)
    return np.ascontiguousarray(Bi.transpose(0, 2, 1).reshape(-1, 9)), det / 6.0  # column-major flat


def tri_rest(verts, tris):
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
    n = len(tris)
    rest = np.zeros((n, 4)); area = np.zeros(n)
    a = C.c_double()
    for t in range(n):
        q = [np.ascontiguousarray(verts[tris[t][i]]) for i in range(3)]
        r = np.zeros(4)
        if lib().orc_tri_rest(_p(q[0]), _p(q[1]), _p(q[2]), _p(r), C.byref(a)):
            raise RuntimeError("**TriEnergyTerm Error: Inverted initial pose")
        rest[t] = r; area[t] = a.value
    return rest, area


def bend_hinges(verts, tris):
    """Hinges of a triangle mesh for the bending term (no reference code; README.md:23-28 TODO): every interior edge (v0 < v1) shared by
    exactly two triangles, v2 / v3 the opposite vertices of the first / second triangle in index order, hinges sorted by (v0, v1); the
    stencil of Bergou et al. 2006 ("A Quadratic Bending Model for Inextensible Surfaces", the vector K of their Eq. 4):
    (c03 + c04, c01 + c02, -(c01 + c03), -(c02 + c04)) with the cotangents of the rest angles at v0 (c01: first triangle, c02: second) and at
    v1 (c03, c04); area = rest area of the two triangles.  Returns (idx [n,4], coef [n,4], area [n])."""
    X = np.asarray(verts, dtype=np.float64).reshape(-1, 3); T = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    edges = {}
    for t in range(len(T)):
        for e in range(3):
            p, q, o = int(T[t, e]), int(T[t, (e + 1) % 3]), int(T[t, (e + 2) % 3])
            edges.setdefault((min(p, q), max(p, q)), []).append((t, o))

    def cot(p, q, r):
        u, v = X[q] - X[p], X[r] - X[p]
        return u.dot(v) / np.linalg.norm(np.cross(u, v))

    def area(p, q, r):
        return 0.5 * np.linalg.norm(np.cross(X[q] - X[p], X[r] - X[p]))
    idx, coef, ar = [], [], []
    for (a, b) in sorted(edges):
        use = sorted(edges[(a, b)])
        if len(use) != 2 or use[0][1] == use[1][1]:
            continue
        v2, v3 = use[0][1], use[1][1]
        c01, c02, c03, c04 = cot(a, b, v2), cot(a, b, v3), cot(b, a, v2), cot(b, a, v3)
        idx.append([a, b, v2, v3]); coef.append([c03 + c04, c01 + c02, -(c01 + c03), -(c02 + c04)]); ar.append(area(a, b, v2) + area(a, b, v3))
    return (np.array(idx, dtype=np.int32).reshape(-1, 4), np.array(coef, dtype=np.float64).reshape(-1, 4), np.array(ar, dtype=np.float64))


PIN_WEIGHT = np.sqrt(lame(10000000.0, 0.499)[2] * 2.0)  # src/SpringEnergyTerm.hpp:47-52


def recolor_touched(base_colors, Ah, dhits):
    """Colouring of A + C^T C (NodalMultiColorGS.hpp:85; the reference's colouring library is ABSENT, so this rule is
    the build's own, shared with the GPU): nodes that no hit touches keep their colour (their rows are rows of A);
    the touched nodes (hit vertex + the three face vertices) are taken out and coloured again, in increasing node
    order, first-fit over NEW colours placed after the old ones, two touched nodes conflicting when they share a hit
    or a non-zero of Ahat.  Sweep order: old colours (without the touched nodes), then the new ones."""
    colors = np.array(base_colors, dtype=np.int32).copy()
    K = int(colors.max()) + 1
    touched = sorted({int(v) for h in dhits for v in [h[0], *h[2]]})
    tset = {v: i for i, v in enumerate(touched)}
    adj = [set() for _ in touched]
    for h in dhits:
        nodes = [int(h[0])] + [int(f) for f in h[2]]
        for a in nodes:
            for bb in nodes:
                if a != bb:
                    adj[tset[a]].add(tset[bb])
    for v in touched:
        for k in range(Ah.indptr[v], Ah.indptr[v + 1]):
            j = int(Ah.indices[k])
            if j != v and j in tset and Ah.data[k] != 0.0:
                adj[tset[v]].add(tset[j])
    extra = [-1] * len(touched)
    for i in range(len(touched)):
        used = {extra[j] for j in adj[i] if extra[j] >= 0}
        e = 0
        while e in used:
            e += 1
        extra[i] = e
    for i, v in enumerate(touched):
        colors[v] = K + extra[i]
    return colors


class BigExactSolve:
    """A x = b TO ROUND-OFF for A = K (x) I3 at sizes where a sparse direct factorisation is out of reach for a test: at the 1 M-tet body
    of bench.py (K: 183 844^2, 2.6 M non-zeros) scipy's SuperLU needs 12 minutes and 345 M non-zeros of fill with MMD(A^T+A), 20 minutes /
    714 M with COLAMD (measured in the build container, round 5).  Stands for the reference's prefactored LDLT (src/LinearSolver.hpp:79-90)
    like the SuperLU solve does at small sizes: preconditioned CG in numpy / scipy on the three axes at once, iterated until the TRUE
    relative residual |b - K x| / |b| of every axis is <= rtol (default 1e-13), re-formed from scratch before it is trusted.
    Nothing of the product is used: the preconditioner is geometric (coordinate boxes carrying {1, x, y, z}, dense coarse inverse,
    three Chebyshev-Jacobi steps) and cannot change the solution, only the iteration count."""

    def __init__(self, K, xyz, rtol=1e-13, max_iters=4000):
        self.K = K.tocsr(); self.rtol, self.max_iters = rtol, max_iters
        nv = K.shape[0]
        self.dinv = 1.0 / K.diagonal()
        X = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
        lo, hi = X.min(axis=0), X.max(axis=0)
        g = max(2, int(round((nv / 500.0) ** (1.0 / 3.0))))
        ijk = np.minimum((np.floor((X - lo) / np.maximum(hi - lo, 1e-300) * g)).astype(np.int64), g - 1)
        _, box = np.unique((ijk[:, 0] * g + ijk[:, 1]) * g + ijk[:, 2], return_inverse=True)
        order = np.argsort(box, kind="stable"); cnt = np.bincount(box); start = np.concatenate([[0], np.cumsum(cnt)])
        rows, cols, vals = [], [], []; nc = 0
        for b in range(len(cnt)):
            idx = order[start[b]:start[b + 1]]
            Y = X[idx] - X[idx].mean(axis=0)
            sc = max(np.abs(Y).max(), 1e-300)
            F = np.column_stack([np.ones(len(idx)), Y / sc]) if len(idx) >= 16 else np.ones((len(idx), 1))
            for j in range(F.shape[1]):
                rows.append(idx); cols.append(np.full(len(idx), nc)); vals.append(F[:, j]); nc += 1
        self.P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nv, nc))
        Kc = (self.P.T @ self.K @ self.P).toarray()
        self.Kci = np.linalg.inv(Kc + 1e-12 * np.trace(Kc) / nc * np.eye(nc))
        v = np.random.default_rng(1).standard_normal(nv)
        lam = 2.0
        for _ in range(30):
            w = self.dinv * (self.K @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
        hi_l = 1.1 * lam; lo_l = hi_l / 30.0
        self.th, self.de = 0.5 * (hi_l + lo_l), 0.5 * (hi_l - lo_l)
        self.iterations = 0; self.solves = 0; self.worst_residual = 0.0

    def _prec(self, R):
        th, de, di = self.th, self.de, self.dinv[:, None]
        y = di * R; al = 1.0 / th; z = al * y; res = R - al * (self.K @ y); p = y
        for k in (1, 2):
            be = (0.5 if k == 1 else 0.25) * (de * al) ** 2
            al = 1.0 / (th - be / al); p = di * res + be * p
            z = z + al * p
            if k == 1:
                res = res - al * (self.K @ p)
        return z + self.P @ (self.Kci @ (self.P.T @ R))

    def solve(self, b, x0=None):
        B = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 3)
        X = np.zeros_like(B) if x0 is None else np.array(x0, dtype=np.float64).reshape(-1, 3).copy()
        bn = np.maximum(np.linalg.norm(B, axis=0), 1e-300 * max(1.0, np.abs(B).max()))
        it = 0
        while True:
            R = B - self.K @ X                                   # true residual: start of a pass
            rel = np.linalg.norm(R, axis=0) / bn
            if rel.max() <= self.rtol or it >= self.max_iters:
                break
            U = self._prec(R); Pd = U.copy(); g = np.einsum("ij,ij->j", R, U)
            for _ in range(300):                                  # one pass; the recursive residual is re-formed above
                W = self.K @ Pd
                a = g / np.maximum(np.einsum("ij,ij->j", Pd, W), 1e-300)
                X += a * Pd; R -= a * W; it += 1
                if (np.linalg.norm(R, axis=0) / bn).max() <= 0.5 * self.rtol or it >= self.max_iters:
                    break
                U = self._prec(R); g2 = np.einsum("ij,ij->j", R, U)
                Pd = U + (g2 / np.maximum(g, 1e-300)) * Pd; g = g2
        self.iterations += it; self.solves += 1; self.worst_residual = max(self.worst_residual, float(rel.max()))
        if rel.max() > 10.0 * self.rtol:
            raise RuntimeError("BigExactSolve: relative residual %.2e after %d iterations" % (rel.max(), it))
        return X.reshape(-1)


class OracleSolver:
    """Restatement of admm::Solver for tets / tris / pins / Floor / Sphere."""

    def __init__(self, x, masses, dt=1.0 / 24.0, gravity=-9.8, admm_iters=10, linsolver=0, constraint_w=-1.0,
                 tets=None, tris=None, pins=None, obstacles=(), mode=1, gs_colors=None,
                 gs_max_iters=30, gs_tol=1e-10, gs_omega=1.9, uzawa_max_iters=20, uzawa_tol=1e-10, big=False,
                 dynamic=(), surface_inds=None, exact="lu", bends=None, slides=None):
        """tets = dict(idx[n,4], verts(rest), kind[n], mu[n], la[n][, k[n]]); tris = dict(idx[n,3], verts, mu, la,
        limit_min, limit_max); pins = {vertex: xyz}; obstacles = [(kind, [4 params])];
        dynamic = [dict(offset, rest[n,3], tets[nt,4] local, faces[nf,3] local)] (TetMeshCollision, in
        add_dynamic_collider order); surface_inds = Solver::surface_inds (None / empty = every vertex);
        masses [3*nv]; mode 0 = reference stop rule, 1 = tight minimiser.
        NOT IN THE REFERENCE (its README lists them as TODOs, README.md:23-28; "parity unpinned: no reference code"), restated from the
        papers / the definitions in include/admm_hip.h: tets of kind 7 = stable Neo-Hookean (Smith et al. 2018; admm_oracle.c);
        bends = dict(idx[n,4], coef[n,4], weight[n], stiffness[n]): bending hinges, D_i x = sum_k c_k x_k (3 rows), E(z) = stiffness/2
        |z|^2 (Bergou et al. 2006), rows after the triangles; slides = {vertex: (point, unit normal)}: slide pins -- SpringPin terms
        (linsolver 0 / 2) whose prox projects onto the plane, or the plane-constrained update inside the GS sweeps (linsolver 1); their
        terms follow the ordinary pins."""
        self.x = np.ascontiguousarray(x, dtype=np.float64).ravel().copy()
        self.dof = self.x.size
        self.nv = self.dof // 3
        self.v = np.zeros(self.dof)
        self.m = np.ascontiguousarray(masses, dtype=np.float64).ravel().copy()
        self.dt, self.gravity, self.admm_iters, self.linsolver, self.mode = dt, gravity, admm_iters, linsolver, mode
        self.gs_max_iters, self.gs_tol, self.gs_omega = gs_max_iters, gs_tol, gs_omega
        self.uz_max_iters, self.uz_tol = uzawa_max_iters, uzawa_tol
        self.obstacles = list(obstacles)
        self.pins = dict(pins or {})
        self.dynamic = [dict(offset=int(d["offset"]), rest=np.ascontiguousarray(d["rest"], dtype=np.float64).reshape(-1, 3),
                             tets=np.ascontiguousarray(d["tets"], dtype=np.int32).reshape(-1, 4),
                             faces=np.ascontiguousarray(d["faces"], dtype=np.int32).reshape(-1, 3)) for d in dynamic]
        self.surface_inds = None if surface_inds is None or len(surface_inds) == 0 else \
            np.ascontiguousarray(surface_inds, dtype=np.int32)
        self._dhits = []
        if self.obstacles and linsolver == 0:
            raise RuntimeError("**Solver::add_obstacle Error: No collisions with LDLT solver")
        self.inner_iters = 0
        L = lib()

        trip_r, trip_c, trip_v, weights = [], [], [], []
        row = 0
        self.nt = 0
        if tets is not None and len(tets["idx"]):
            idx = np.ascontiguousarray(tets["idx"], dtype=np.int32).reshape(-1, 4)
            self.nt = idx.shape[0]
            Binv, vol = (tet_rest_fast if big else tet_rest)(tets["verts"], idx)
            self.t_idx, self.t_Binv, self.t_vol = idx, np.ascontiguousarray(Binv), vol
            self.t_kind = np.ascontiguousarray(np.broadcast_to(tets["kind"], (self.nt,)), dtype=np.int32)
            self.t_mu = np.ascontiguousarray(np.broadcast_to(tets["mu"], (self.nt,)), dtype=np.float64)
            self.t_la = np.ascontiguousarray(np.broadcast_to(tets["la"], (self.nt,)), dtype=np.float64)
            # k = the TET's bulk modulus (EnergyTerm.hpp:41); mu / la may be a SplineTet's own spline constants
            self.t_k = (np.ascontiguousarray(np.broadcast_to(tets["k"], (self.nt,)), dtype=np.float64) if "k" in tets
                        else self.t_la + (2.0 / 3.0) * self.t_mu)
            self.t_kappa = np.ascontiguousarray(np.broadcast_to(tets.get("kappa", 0.0), (self.nt,)), dtype=np.float64)  # xu:: splines
            self.t_w = np.sqrt(self.t_k * vol)                    # TetEnergyTerm.cpp:46-47
            if np.any(self.t_w <= 0):
                raise RuntimeError("**EnergyTerm::get_reduction Error: Some weight leq 0")
            # D-block (TetEnergyTerm.cpp:50-71): Dt(r,c) = (S Binv)^T ; rows {0,3,6}+j, cols 3 tet[c] + j
            Bm = Binv.reshape(-1, 3, 3).transpose(0, 2, 1)        # Bm[t, m, r] = Binv(m, r)
            Dm = np.concatenate([-Bm.sum(axis=1, keepdims=True), Bm], axis=1)  # [t, 4, 3] = S * Binv
            for r in range(3):
                for c in range(4):
                    for j in range(3):
                        trip_r.append(row + 9 * np.arange(self.nt) + 3 * r + j)
                        trip_c.append(3 * idx[:, c] + j)
                        trip_v.append(Dm[:, c, r])
            weights.append(np.repeat(self.t_w, 9))
            row += 9 * self.nt
        self.ntri = 0
        if tris is not None and len(tris["idx"]):
            idx = np.ascontiguousarray(tris["idx"], dtype=np.int32).reshape(-1, 3)
            self.ntri = idx.shape[0]
            rest, area = tri_rest(tris["verts"], idx)
            self.r_idx, self.r_rest = idx, np.ascontiguousarray(rest)
            mu = np.broadcast_to(tris["mu"], (self.ntri,)).astype(np.float64)
            la = np.broadcast_to(tris["la"], (self.ntri,)).astype(np.float64)
            self.r_lmin = np.ascontiguousarray(np.broadcast_to(tris.get("limit_min", -100.0), (self.ntri,)), dtype=np.float64)
            self.r_lmax = np.ascontiguousarray(np.broadcast_to(tris.get("limit_max", 100.0), (self.ntri,)), dtype=np.float64)
            self.r_w = np.sqrt((la + (2.0 / 3.0) * mu) * area)    # TriEnergyTerm.cpp:50-51
            Rm = rest.reshape(-1, 2, 2).transpose(0, 2, 1)        # Rm[t, m, c] = rest(m, c)
            Dm = np.concatenate([-Rm.sum(axis=1, keepdims=True), Rm], axis=1)  # [t, 3, 2] = S * rest
            for i in range(3):
                for j in range(3):
                    trip_r.append(row + 6 * np.arange(self.ntri) + i); trip_c.append(3 * idx[:, j] + i); trip_v.append(Dm[:, j, 0])
                    trip_r.append(row + 6 * np.arange(self.ntri) + 3 + i); trip_c.append(3 * idx[:, j] + i); trip_v.append(Dm[:, j, 1])
            weights.append(np.repeat(self.r_w, 6))
            row += 6 * self.ntri
        self.nbend = 0
        if bends is not None and len(bends["idx"]):
            self.h_idx = np.ascontiguousarray(bends["idx"], dtype=np.int64).reshape(-1, 4)
            self.h_coef = np.ascontiguousarray(bends["coef"], dtype=np.float64).reshape(-1, 4)
            self.h_w = np.ascontiguousarray(bends["weight"], dtype=np.float64)
            self.h_k = np.ascontiguousarray(bends["stiffness"], dtype=np.float64)
            self.nbend = self.h_idx.shape[0]
            for k in range(4):
                for j in range(3):
                    trip_r.append(row + 3 * np.arange(self.nbend) + j); trip_c.append(3 * self.h_idx[:, k] + j); trip_v.append(self.h_coef[:, k])
            weights.append(np.repeat(self.h_w, 3))
            self.h_row = row
            row += 3 * self.nbend
        # pins become SpringPin terms for LDLT / Uzawa (Solver.cpp:190-196); slide pins (not in the reference) follow them
        self.npin = 0
        self.slides = {int(k): (np.asarray(v[0], dtype=np.float64), np.asarray(v[1], dtype=np.float64) / np.linalg.norm(v[1])) for k, v in (slides or {}).items()}
        self.nslide = 0
        if linsolver in (0, 2) and (self.pins or self.slides):
            allv = list(self.pins.keys()) + list(self.slides.keys())
            self.p_vert = np.array(allv, dtype=np.int32)
            self.p_xyz = np.ascontiguousarray(np.array(list(self.pins.values()) + [v[0] for v in self.slides.values()], dtype=np.float64).reshape(-1, 3))
            self.p_nrm = np.ascontiguousarray(np.array([np.zeros(3)] * len(self.pins) + [v[1] for v in self.slides.values()], dtype=np.float64).reshape(-1, 3))
            self.p_active = np.ones(len(self.p_vert), dtype=np.int32)
            self.npin = len(self.p_vert); self.nslide = len(self.slides)
            for j in range(3):
                trip_r.append(row + 6 * np.arange(self.npin) + j); trip_c.append(3 * self.p_vert + j); trip_v.append(np.ones(self.npin))
            weights.append(np.full(6 * self.npin, PIN_WEIGHT))
            row += 6 * self.npin
        self.R = row
        self.W = np.concatenate(weights) if weights else np.zeros(0)
        rr = np.concatenate(trip_r) if trip_r else np.zeros(0, int)
        cc = np.concatenate(trip_c) if trip_c else np.zeros(0, int)
        vv = np.concatenate(trip_v) if trip_v else np.zeros(0)
        self.D = sp.csr_matrix((vv, (rr, cc)), shape=(self.R, self.dof))
        dt2 = dt * dt
        self.DtWtW = (dt2 * self.D.T @ sp.diags(self.W * self.W)).tocsr()   # Solver.cpp:225
        self.A = (sp.diags(self.m) + self.DtWtW @ self.D).tocsr()            # Solver.cpp:226
        self.A.sum_duplicates()
        # constraint weight (Solver.cpp:235, 239, 245)
        wmax = self.W.max() if self.W.size else 0.0
        self.constraint_w = 3.0 * wmax if linsolver == 1 else 1.0
        if constraint_w > 0:
            self.constraint_w = constraint_w
        self._lu = None; self._big = None
        if linsolver in (0, 2) and exact == "pcg":
            # (exact = "pcg": the prefactored solve at sizes SuperLU cannot factor in test time -- BigExactSolve, to round-off)
            K = self.A[0::3, :][:, 0::3].tocsr()
            if not (np.array_equal(self.m[0::3], self.m[1::3]) and np.array_equal(self.m[0::3], self.m[2::3])):
                raise RuntimeError("exact='pcg' needs per-vertex masses (A = K (x) I3)")
            self._big = BigExactSolve(K, self.x)
        elif linsolver in (0, 2):
            self._lu = spla.splu(self.A.tocsc())
        if linsolver == 1:
            Ah = self.A[0::3, :][:, 0::3].tocsr()
            Ah.sort_indices()
            self.Ah = Ah
            self.gs_colors = np.asarray(gs_colors, dtype=np.int32) if gs_colors is not None else None
        self.y = np.zeros(0)  # Uzawa multipliers (warm start, UzawaCG.hpp:74)
        # ADMM state
        self.z = np.zeros(self.R); self.u = np.zeros(self.R)

    # -- local step (Solver.cpp:84-87 + EnergyTerm.hpp:130-140) on the AoS z/u in reference row order
    def local_step(self, curr_x, z, u):
        L = lib()
        curr_x = np.ascontiguousarray(curr_x)
        o = 0
        if self.nt:
            zz = np.ascontiguousarray(z[o:o + 9 * self.nt]); uu = np.ascontiguousarray(u[o:o + 9 * self.nt])
            L.orc_local_tets_k(self.nt, _i(self.t_idx), _p(self.t_Binv), _i(self.t_kind), _p(self.t_mu), _p(self.t_la),
                               _p(self.t_k), _p(self.t_kappa), _p(curr_x), _p(zz), _p(uu), self.mode)
            z[o:o + 9 * self.nt] = zz; u[o:o + 9 * self.nt] = uu
            o += 9 * self.nt
        if self.ntri:
            zz = np.ascontiguousarray(z[o:o + 6 * self.ntri]); uu = np.ascontiguousarray(u[o:o + 6 * self.ntri])
            L.orc_local_tris(self.ntri, _i(self.r_idx), _p(self.r_rest), _p(self.r_lmin), _p(self.r_lmax), _p(curr_x), _p(zz), _p(uu))
            z[o:o + 6 * self.ntri] = zz; u[o:o + 6 * self.ntri] = uu
            o += 6 * self.ntri
        if self.nbend:      # EnergyTerm::update (EnergyTerm.hpp:130-140) of a bending hinge: prox of E(z) = stiffness / 2 |z|^2
            X = curr_x.reshape(-1, 3)
            Dx = np.einsum("hk,hkj->hj", self.h_coef, X[self.h_idx])
            uu = u[o:o + 3 * self.nbend].reshape(-1, 3)
            q = Dx + uu
            w2 = self.h_w * self.h_w
            zz = q * (w2 / (self.h_k + w2))[:, None]
            u[o:o + 3 * self.nbend] = (uu + Dx - zz).ravel(); z[o:o + 3 * self.nbend] = zz.ravel()
            o += 3 * self.nbend
        if self.npin:
            zz = np.ascontiguousarray(z[o:o + 6 * self.npin]); uu = np.ascontiguousarray(u[o:o + 6 * self.npin])
            uu_in = uu.copy()
            L.orc_local_pins(self.npin, _i(self.p_vert), _p(self.p_xyz), _i(self.p_active), _p(curr_x), _p(zz), _p(uu))
            if self.nslide:     # slide pins: z = q - n (n . (q - p)), q = x_v + u (rows 0-2 of the term; 3-5 stay zero like a SpringPin's)
                X = curr_x.reshape(-1, 3)
                for i in range(self.npin - self.nslide, self.npin):
                    if not self.p_active[i]:
                        continue
                    xv = X[self.p_vert[i]]; uo = uu_in[6 * i:6 * i + 3]; n = self.p_nrm[i]
                    q = xv + uo
                    zi = q - n * n.dot(q - self.p_xyz[i])
                    zz[6 * i:6 * i + 3] = zi; uu[6 * i:6 * i + 3] = uo + xv - zi
            z[o:o + 6 * self.npin] = zz; u[o:o + 6 * self.npin] = uu

    # -- Collider::detect with passive objects (Collider.hpp:152-212); hits in vertex order
    def detect_passive(self, x):
        if not self.obstacles:
            return []
        X = x.reshape(-1, 3)
        best = np.full(self.nv, np.finfo(np.float64).max)
        point = np.zeros((self.nv, 3)); normal = np.zeros((self.nv, 3))
        for kind, par in self.obstacles:
            if kind == 0:   # Floor, PassiveObject.hpp:37-43
                dx = X[:, 1] - par[0]
                upd = ~(dx > best)
                p = X.copy(); p[:, 1] = par[0]
                n = np.zeros_like(X); n[:, 1] = 1.0
            elif kind == 2:  # a user-side PassiveCollision: the half space n.x < d (unit n)
                nn = np.asarray(par[:3], dtype=np.float64)
                dx = X @ nn - par[3]
                upd = ~(dx > best)
                n = np.tile(nn, (len(X), 1))
                p = X - dx[:, None] * nn
            else:           # Sphere, PassiveObject.hpp:55-62
                d = X - np.asarray(par[:3])
                l = np.linalg.norm(d, axis=1)
                dx = l - par[3]
                upd = ~(dx > best)
                n = d / l[:, None]
                p = np.asarray(par[:3]) + n * par[3]
            best = np.where(upd, dx, best)
            point[upd] = p[upd]; normal[upd] = n[upd]
        hit = np.nonzero(best < 0)[0]
        if self.surface_inds is not None:                        # Collider.hpp:157,163: only the listed vertices
            hit = np.array([i for i in self.surface_inds if best[i] < 0], dtype=int)
        return [(int(i), best[i], point[i].copy(), normal[i].copy()) for i in hit]

    # -- Collider::detect with dynamic objects (Collider.hpp:192-201, DynamicObject.hpp:72-119); hits in query order
    def detect_dynamic(self, x):
        if not self.dynamic:
            return []
        L = lib()
        q = self.surface_inds if self.surface_inds is not None else np.arange(self.nv, dtype=np.int32)
        nq = len(q)
        hit = np.zeros(nq, dtype=np.int32); face = np.full((nq, 3), -1, dtype=np.int32)
        bary = np.zeros((nq, 3)); nrm = np.zeros((nq, 3)); dx = np.zeros(nq)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        for d in self.dynamic:
            L.orc_detect_dynamic(nq, _i(q), _p(xx), d["offset"], _p(d["rest"]), len(d["tets"]), _i(d["tets"]),
                                 len(d["faces"]), _i(d["faces"]), _i(hit), _i(face), _p(bary), _p(nrm), _p(dx))
        return [(int(q[i]), dx[i], face[i].copy(), bary[i].copy(), nrm[i].copy()) for i in np.nonzero(hit)[0]]

    # -- ConstraintSet::make_matrix (ConstraintSet.hpp:59-116): passive rows, then dynamic rows; a vertex that already
    #    holds a row keeps it and the later hit's row stays empty (`constrained`, :79-82,:96-99)
    def make_matrix(self, hits, dhits=()):
        ck = np.sqrt(max(0.0, self.constraint_w))
        rows, cols, vals = [], [], []
        c = np.zeros(len(hits) + len(dhits))
        constrained = {}
        for i, (vi, dx, p, n) in enumerate(hits):
            if constrained.get(vi, 0.0):
                continue
            if dx < 0.0:
                constrained[vi] = dx
            c[i] = ck * n.dot(p)
            for j in range(3):
                rows.append(i); cols.append(3 * vi + j); vals.append(ck * n[j])
        for i, (vi, dx, face, bary, n) in enumerate(dhits):
            if constrained.get(vi, 0.0):
                continue
            if dx < 0.0:
                constrained[vi] = dx
            ci = i + len(hits)
            for j in range(3):
                rows.append(ci); cols.append(3 * vi + j); vals.append(ck * n[j])
            for k in range(3):
                for j in range(3):
                    rows.append(ci); cols.append(3 * int(face[k]) + j); vals.append(-ck * n[j] * bary[k])
        Cm = sp.csr_matrix((vals, (rows, cols)), shape=(len(c), self.dof))
        return Cm, c

    # -- global solvers
    def solve_ldlt(self, b, x0=None):
        if self._big is not None:
            return self._big.solve(b, x0)         # (warm start: fewer iterations, the same solution to round-off)
        return self._lu.solve(b)

    def solve_uzawa(self, x, b, hits):
        """UzawaCG::solve (UzawaCG.hpp:57-125); returns (x, iters)."""
        Cm, c = self.make_matrix(hits, self._dhits)
        if self.y.shape[0] != Cm.shape[0]:
            self.y = np.zeros(Cm.shape[0])
        if Cm.nnz == 0:
            return self.solve_ldlt(b, x), 1
        Ct = Cm.T.tocsr()
        x = self._lu.solve(b - Ct @ self.y)
        r = Cm @ x - c
        d = r.copy()
        tol2 = self.uz_tol * self.uz_tol
        tiny = np.finfo(np.float64).tiny
        it = 0
        while it < self.uz_max_iters:
            q2 = self._lu.solve(Ct @ d)
            q3 = Cm @ q2
            denom = d.dot(q3)
            if abs(denom) < tiny:
                break
            alpha = d.dot(r) / denom
            x = x - alpha * q2
            self.y = self.y + alpha * d
            r = r - alpha * q3
            if r.dot(r) < tol2:
                break
            beta = r.dot(q3) / denom
            d = r - beta * d
            it += 1
        return x, it

    def solve_gs(self, x, b):
        L = lib()
        if self.gs_colors is None:
            raise RuntimeError("oracle GS needs colours (the reference's colouring library is absent)")
        nc = int(self.gs_colors.max()) + 1
        order = np.argsort(self.gs_colors, kind="stable").astype(np.int32)
        cptr = np.zeros(nc + 1, dtype=np.int32)
        np.cumsum(np.bincount(self.gs_colors, minlength=nc), out=cptr[1:])
        pin_flag = np.zeros(self.nv, dtype=np.int32); pin_xyz = np.zeros((self.nv, 3))
        for k, p in self.pins.items():
            pin_flag[k] = 1; pin_xyz[k] = p
        if self.slides:
            self._gs_nrm = np.zeros((self.nv, 3))
            for k, (p, n) in self.slides.items():
                pin_flag[k] = 2; pin_xyz[k] = p; self._gs_nrm[k] = n
            L.orc_gs_set_pin_normals(_p(self._gs_nrm))
        okind = np.array([o[0] for o in self.obstacles], dtype=np.int32)
        opar = np.ascontiguousarray(np.array([o[1] for o in self.obstacles], dtype=np.float64).reshape(-1, 4))
        x = np.ascontiguousarray(x).copy()
        b = np.ascontiguousarray(b)
        if self._dhits:
            # dynamic hits: A <- A + C^T C, b <- b + C^T c (c = 0), re-colour (NodalMultiColorGS.hpp:80-86)
            Cm, c = self.make_matrix([], self._dhits)
            M = (self.A + Cm.T @ Cm).tocsr(); M.sort_indices()
            colors = recolor_touched(self.gs_colors, self.Ah, self._dhits)
            nc2 = int(colors.max()) + 1
            order2 = np.argsort(colors, kind="stable").astype(np.int32)
            cptr2 = np.zeros(nc2 + 1, dtype=np.int32)
            np.cumsum(np.bincount(colors, minlength=nc2), out=cptr2[1:])
            b2 = np.ascontiguousarray(b + Cm.T @ c)
            it = L.orc_gs_solve_full(self.nv, _i(np.ascontiguousarray(M.indptr, dtype=np.int32)),
                                     _i(np.ascontiguousarray(M.indices, dtype=np.int32)), _p(np.ascontiguousarray(M.data)), _p(b2), _p(x),
                                     nc2, _i(cptr2), _i(order2), _i(pin_flag) if (self.pins or self.slides) else None, _p(pin_xyz), len(self.obstacles),
                                     _i(okind) if len(okind) else None, _p(opar) if len(okind) else None,
                                     self.gs_omega, self.gs_max_iters, self.gs_tol)
            return x, it
        rp = np.ascontiguousarray(self.Ah.indptr, dtype=np.int32); ci = np.ascontiguousarray(self.Ah.indices, dtype=np.int32)
        va = np.ascontiguousarray(self.Ah.data)
        it = L.orc_gs_solve(self.nv, _i(rp), _i(ci), _p(va), _p(b), _p(x), nc, _i(cptr), _i(order),
                            _i(pin_flag) if (self.pins or self.slides) else None, _p(pin_xyz), len(self.obstacles),
                            _i(okind) if len(okind) else None, _p(opar) if len(okind) else None,
                            self.gs_omega, self.gs_max_iters, self.gs_tol)
        return x, it

    def rhs(self, Mxbar, z, u):
        return Mxbar + self.DtWtW @ (z - u)          # Solver.cpp:98

    def global_solve(self, x, b):
        if self.linsolver == 1:
            return self.solve_gs(x, b)
        if self.linsolver == 2:
            return self.solve_uzawa(x, b, self._hits)
        return self.solve_ldlt(b, x), 1

    def step(self, trace=None):
        """Solver::step (Solver.cpp:35-110). trace: optional list receiving (z,u,b,x) per ADMM iteration."""
        dt = self.dt
        if abs(self.gravity) > 0:
            self.v[1::3] += dt * self.gravity                     # :57-59
        x_bar = self.x + dt * self.v                              # :65
        Mxbar = self.m * x_bar
        curr = x_bar.copy()
        z = np.zeros(self.R); u = np.zeros(self.R)                # :70-71 (z = D x is a dead store)
        self.inner_iters = 0
        for it_ in range(self.admm_iters):
            self.local_step(curr, z, u)                           # :84-87
            if it_ == 0 or not getattr(self, "freeze_active", False):   # (freeze_active: test mode, one detect per step)
                self._hits = self.detect_passive(curr) if self.linsolver != 1 else []   # :92-93
                self._dhits = self.detect_dynamic(curr)
            b = self.rhs(Mxbar, z, u)                             # :98
            curr, it = self.global_solve(curr, b)                 # :99
            self.inner_iters += it
            if trace is not None:
                trace.append((z.copy(), u.copy(), b.copy(), curr.copy()))
        self.v = (curr - self.x) / dt                             # :105
        self.x = curr                                             # :106
        self.z, self.u = z, u

    def set_pins(self, pins):
        """Solver::set_pins after initialize (Solver.cpp:113-157)."""
        self.pins = dict(pins)
        if self.linsolver in (0, 2) and self.npin:
            self.p_active[:] = 0
            where = {int(v): i for i, v in enumerate(self.p_vert)}
            for k, p in self.pins.items():
                if k not in where:
                    raise RuntimeError("**Solver::set_pins Error: Constraint for %d not found." % k)
                self.p_active[where[k]] = 1
                self.p_xyz[where[k]] = p

    # TetEnergyTerm::energy for one linear tet (TetEnergyTerm.cpp:94-100) given its 9 D-rows times x
    def tet_energy_linear(self, t, x):
        F = (self.D[9 * t:9 * t + 9] @ x)
        return lib().orc_energy_tet_linear(_p(np.ascontiguousarray(F)), self.t_k[t], self.t_vol[t])
