"""The host-side plan of the on-chip PCG for general meshes (csrc/oc_plan.cpp; replaces the prefactored solve of
src/LinearSolver.hpp:87-90): internal row order, aggregates and the dense coarse inverse, checked on the CPU by running
the two-level preconditioned CG it describes in numpy / scipy."""
import numpy as np
import pytest
import scipy.sparse as sp

import scenes


def _system(sc):
    s = sc.make_solver(init=False)
    rp, ci, va = s.host_matrix(sc.product_settings)
    nv = len(sc.x)
    A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
    return s, A


def _pcg(A, b, prec, tol=1e-8, maxit=2000):
    dinv = 1.0 / A.diagonal()
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b)
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        if r @ (dinv * r) <= tol * tol * b2:
            return x, it + 1
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit


def _check_plan(sc, G, spb, affine=True):
    s, A = _system(sc)
    nv = A.shape[0]
    plan = s.host_oc_plan(G, spb, settings=sc.product_settings)
    rv, ra, st = plan["row_vertex"], plan["row_aggregate"], plan["stats"]
    live = rv >= 0
    # every vertex has exactly one row; blocks and aggregates are what the slot says
    assert np.array_equal(np.sort(rv[live]), np.arange(nv))
    assert len(rv) == 64 * G * spb and st["coarse_unknowns"] == 4 * G
    assert np.array_equal(ra[live] // 4, np.nonzero(live)[0] // (64 * spb))
    assert (ra[~live] == -1).all()
    agg = np.empty(nv, np.int64); agg[rv[live]] = ra[live]
    # the rows of a block are sorted by length (longest first): similar lengths inside a wavefront
    deg = np.diff((A != 0).astype(int).tocsr().indptr) - 1
    for blk in rv.reshape(G, -1):
        d = deg[blk[blk >= 0]]
        assert (np.diff(d) <= 0).all()
    # the coarse inverse is the inverse of P^T A P (empty coarse unknowns: unit diagonal); row r of P = the row's weights in the
    # four coarse functions of its block: (1, x, y, z) centred and scaled per block (the scenes pass their positions as
    # desc.vert_xyz), one-hot on the aggregate when ADMM_HIP_OC_AFFINE=0
    nc = 4 * G
    wt = plan["row_weights"]
    blk_of_row = np.arange(len(rv)) // (64 * spb)
    rr = np.repeat(np.nonzero(live)[0], 4)
    P = sp.csr_matrix((wt[live].ravel().astype(np.float64), (np.repeat(rv[live], 4), (4 * blk_of_row[rr] + np.tile(np.arange(4), live.sum())))), shape=(nv, nc))
    assert (wt[~live] == 0).all()
    if affine:
        # per block: the span of {1, x, y, z} (the functions a direction without extent would give are dropped), orthonormal in
        # the energy of the block's own part of A
        X = np.concatenate([np.ones((nv, 1)), np.asarray(sc.x, dtype=np.float64).reshape(-1, 3)], axis=1)
        for b in range(G):
            m = live & (blk_of_row == b)
            if m.sum() < 4:
                continue
            Wb = wt[m].astype(np.float64); Xb = X[rv[m]]
            used = np.abs(Wb).max(axis=0) > 0
            coef = np.linalg.lstsq(Xb, Wb[:, used], rcond=None)[0]
            assert np.abs(Xb @ coef - Wb[:, used]).max() < 1e-4 * np.abs(Wb).max()          # in the span
            Abb = A[rv[m]][:, rv[m]]
            Gm = Wb[:, used].T @ (Abb @ Wb[:, used])
            assert np.abs(Gm - np.eye(used.sum())).max() < 1e-4                                  # A_bb-orthonormal (FP32 weights)
    else:
        assert ((wt[live] == 1.0).sum(axis=1) == 1).all() and np.array_equal(np.argmax(wt[live], axis=1), ra[live] % 4)
    Ac = (P.T @ A @ P).toarray()
    empty = np.diag(Ac) == 0
    Ac[empty, empty] = 1.0
    err = np.abs(plan["coarse_inv"] @ Ac - np.eye(nc)).max()
    assert err < 2e-6, err          # (cond(P^T A P) ~ 1e5..1e6 with the affine functions)
    # the two-level preconditioner needs markedly fewer iterations than Jacobi and solves the same system
    dinv = 1.0 / A.diagonal()
    b = A @ np.random.default_rng(0).standard_normal(nv)
    x0, it_j = _pcg(A, b, lambda r: dinv * r, tol=1e-11)
    x1, it_2 = _pcg(A, b, lambda r: dinv * r + P @ (plan["coarse_inv"] @ (P.T @ r)), tol=1e-11)
    assert np.abs(x1 - x0).max() <= 1e-7 * np.abs(x0).max()
    # the block-local Chebyshev smoother of the kernel is built on the plan's estimate of lambda_max(D^-1 A_bb): it must not
    # underestimate by more than the kernel's margin (coefficients from 1.1 x the estimate, polynomial positive up to 1.125 x
    # that), and S + coarse must beat D^-1 + coarse
    import scipy.sparse.linalg as spla
    blk = agg // 4
    coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
    Ab = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
    Sb = sp.diags(np.sqrt(dinv)) @ Ab @ sp.diags(np.sqrt(dinv))
    lam = spla.eigsh(Sb, k=1, which="LA", return_eigenvectors=False)[0]
    assert 0.9 * lam <= st["lambda_bb"] <= 1.0001 * lam, (lam, st["lambda_bb"])
    hi = 1.1 * st["lambda_bb"]; lo = hi / 16.0; th, de = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sg = th / de; r0 = 1.0 / sg; r1 = 1.0 / (2.0 * sg - r0)
    al = (1.0 + r1 * r0) / th + 2.0 * r1 / de; be = 2.0 * r1 / (de * th)          # admm_hip.hip: oc_sm_ab = al - be, oc_sm_b = be
    assert al - be * lam > 0.0                                                     # q(lambda) > 0 on the whole spectrum: S is SPD
    Aoff = Ab - sp.diags(Ab.diagonal())
    S = lambda r: dinv * ((al - be) * r - be * (Aoff @ (dinv * r)))
    x2, it_3 = _pcg(A, b, lambda r: S(r) + P @ (plan["coarse_inv"] @ (P.T @ r)), tol=1e-11)
    assert np.abs(x2 - x0).max() <= 1e-7 * np.abs(x0).max() and it_3 < 0.9 * it_2, (it_2, it_3)
    assert st["bank_load_placed"] <= st["bank_load_by_index"] + 1e-9
    return it_j, it_2, st


def test_plan_unstructured_body():
    sc = scenes.blob_scene(30, admm_iters=5, linsolver=0)      # 16 k tets, valences 3..26
    it_j, it_2, st = _check_plan(sc, 16, 4)
    assert it_2 < 0.75 * it_j, (it_j, it_2)
    # compact blocks: most of the couplings stay inside a block, few neighbour blocks
    assert st["block_local"] > 0.6 * st["nnz"] and 0 < st["max_neighbour_blocks"] <= 15
    # rows sorted by length at the real block size (12 wavefronts): little padding, (nearly) everything in LDS
    sc = scenes.blob_scene(44, admm_iters=5, linsolver=0)
    st = sc.make_solver(init=False).host_oc_plan(16, 12, settings=sc.product_settings, coarse=False)["stats"]
    assert st["stored"] < 1.35 * st["nnz"] and st["on_chip"] >= 0.9 * st["stored"], st


def test_plan_structured_cube_and_cloth():
    sc = scenes.mixed_cube_scene(12, admm_iters=5, linsolver=0)
    it_j, it_2, st = _check_plan(sc, 9, 4)
    assert it_2 < it_j
    sc = scenes.cloth_scene(40, admm_iters=5, linsolver=0)
    it_j, it_2, st = _check_plan(sc, 7, 4)
    assert it_2 < it_j


def test_plan_piecewise_constant_coarse_space(monkeypatch):
    """ADMM_HIP_OC_AFFINE=0 (or no desc.vert_xyz): four compact aggregates per block, the round-2 coarse space; the affine one
    needs no more iterations than it (same number of coarse unknowns)."""
    sc = scenes.blob_scene(30, admm_iters=5, linsolver=0)
    it_j, it_a, _ = _check_plan(sc, 16, 4)
    monkeypatch.setenv("ADMM_HIP_OC_AFFINE", "0")
    it_j, it_c, _ = _check_plan(sc, 16, 4, affine=False)
    assert it_a <= it_c + 2, (it_a, it_c)


def test_plan_rejects_too_few_slots():
    import pytest
    import admm_elastic_amd as pkg
    sc = scenes.cube_scene(6, pkg.TET_NEOHOOKEAN, linsolver=0)
    s = sc.make_solver(init=False)
    with pytest.raises(pkg.AdmmHipError):
        s.host_oc_plan(2, 2, settings=sc.product_settings)      # 256 slots for 343 vertices


def test_launch_path_two_level_plan():
    """admm_host_big_plan (csrc/oc_plan.cpp: build_big_plan, the plan of csrc/pcg_big.hpp), no GPU: every vertex in exactly one aggregate slot,
    aggregates of equal size, the coarse functions energy-orthonormal per aggregate (diagonal blocks of P^T A P = I), the stored single-precision
    inverse the inverse of P^T A P, and M^-1 = D^-1 + P (P^T A P)^-1 P^T a preconditioner that clusters the spectrum (CG on the host: several
    times fewer iterations than Jacobi)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    sc = scenes.blob_scene(48, admm_iters=4, linsolver=0)
    s = sc.make_solver(init=False)
    nv = len(sc.x)
    rp, ci, va = s.host_matrix(sc.product_settings)
    A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
    P = s.host_big_plan(max_aggregates=0, settings=sc.product_settings)
    G, ra, rows, nc = P["G"], P["ra"], P["rows"], P["nc"]
    assert G >= 8 and rows == G * ra and ra % 256 == 0 and nc == 4 * G
    rv = P["row_vertex"]
    live = np.nonzero(rv >= 0)[0]
    assert len(live) == nv and np.array_equal(np.sort(rv[live]), np.arange(nv))
    agg = live // ra
    assert np.bincount(agg, minlength=G).max() <= ra
    # prolongation in vertex order: column 4 g + k = function k of aggregate g
    Pm = sp.lil_matrix((nv, nc))
    for r in live:
        for k in range(4):
            Pm[rv[r], 4 * (r // ra) + k] = P["row_weights"][r, k]
    Pm = Pm.tocsr()
    Ac = (Pm.T @ A @ Pm).toarray()
    for g in range(G):
        blk = Ac[4 * g:4 * g + 4, 4 * g:4 * g + 4]
        keep = np.abs(np.diag(blk)) > 0.5          # (a function dropped by the pivoted Cholesky has a zero column)
        assert np.abs(blk[np.ix_(keep, keep)] - np.eye(keep.sum())).max() < 1e-9
    keep = np.abs(np.diag(Ac)) > 0.5
    inv = P["coarse_inv"].astype(float)
    assert np.abs(inv[np.ix_(keep, keep)] @ Ac[np.ix_(keep, keep)] - np.eye(keep.sum())).max() < 2e-3      # single precision, cond ~1e2
    dinv = 1.0 / A.diagonal()
    M2 = spl.LinearOperator((nv, nv), matvec=lambda r: dinv * r + Pm @ (inv @ (Pm.T @ r)))
    MJ = spl.LinearOperator((nv, nv), matvec=lambda r: dinv * r)
    b = np.random.default_rng(2).standard_normal(nv)
    its = {}
    for name, M in (("jacobi", MJ), ("two_level", M2)):
        n = [0]
        x, info = spl.cg(A, b, rtol=1e-8, maxiter=5000, M=M, callback=lambda xk: n.__setitem__(0, n[0] + 1))
        assert info == 0
        its[name] = n[0]
    print("host CG iterations", its, "aggregates", G)
    assert its["two_level"] * 2 < its["jacobi"], its


@pytest.mark.parametrize("kind", ["cube", "blob", "cloth"])
def test_persistent_gs_plan_swept_on_the_host(kind):
    """admm_host_gs_plan_sweeps: the plan of the persistent multi-colour GS kernel (csrc/oc_plan.cpp: build_gs_plan) run on the host in the
    kernel's order of phases -- blocks, per-colour ELLs with local columns, halo lists, outbox nodes -- against plain multi-colour SOR sweeps
    on the assembled matrix (src/NodalMultiColorGS.hpp:180-216).  A dropped entry, a misrouted halo value or boundary row shows at once."""
    import scipy.sparse as sp
    import admm_elastic_amd as pkg
    from admm_elastic_amd import capi
    import ctypes as C
    if kind == "cube": sc = scenes.cube_scene(9, pkg.TET_NEOHOOKEAN, linsolver=1)
    elif kind == "blob": sc = scenes.blob_scene(22, admm_iters=4, linsolver=1)
    else: sc = scenes.cloth_scene(30, linsolver=1)
    s = sc.make_solver(init=False)
    nv = len(sc.x)
    rp, ci, va = s.host_matrix(sc.product_settings)
    A = sp.csr_matrix((va, ci, rp), shape=(nv, nv)); A.eliminate_zeros(); A.sort_indices()
    color, nc = capi.greedy_coloring(A.indptr.astype(np.int32), A.indices.astype(np.int32))
    assert nc <= 12
    rng = np.random.default_rng(7)
    b = rng.standard_normal((nv, 3)); x0 = rng.standard_normal((nv, 3))
    masses = np.repeat(np.asarray(sc.m, float).reshape(nv, -1)[:, :1], 3, axis=1) if np.asarray(sc.m).size == nv else np.asarray(sc.m, float).reshape(nv, 3)
    omega, sweeps = 1.9, 3
    # reference: colour by colour, every row of a colour from the current x
    x = x0.copy()
    offd = A - sp.diags(A.diagonal())
    aii = A.diagonal()[:, None] + masses
    for _ in range(sweeps):
        for c in range(nc):
            rows = np.nonzero(color == c)[0]
            lux = offd[rows] @ x
            x[rows] = omega * (b[rows] - lux) / aii[rows] + (1.0 - omega) * x[rows]
    d = s.make_desc(sc.product_settings)
    xs = np.ascontiguousarray(x0.ravel().copy()); bb = np.ascontiguousarray(b.ravel())
    st = np.zeros(6, np.int32)
    capi.check(capi.lib().admm_host_gs_plan_sweeps(C.byref(d), int(nc), capi.iptr(color.astype(np.int32)), 8, 64, capi.dptr(bb), capi.dptr(xs), sweeps, omega, capi.iptr(st)))
    assert st[0] > 1 and st[1] == nc and st[3] > 0          # several blocks, a halo
    err = np.abs(xs.reshape(nv, 3) - x).max() / np.abs(x).max()
    assert err < 1e-12, err
