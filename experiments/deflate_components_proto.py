"""Free bodies (no pins): Jacobi-PCG iterations with and without deflating the per-component constant vectors (the
near-null space of Ahat: A 1_c = m restricted to component c).  CPU prototype (scipy), per axis scalar system."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import scenes
import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Lame

def pcg(A, b, dinv, tol, W=None, maxit=5000):
    x = np.zeros_like(b)
    if W is not None:
        AW = A @ W; E = W.T @ AW; Einv = np.linalg.inv(E)
        x = W @ (Einv @ (W.T @ b))
    r = b - A @ x
    def proj(v):  # P^T-ish: remove the W-component of the residual direction (deflated CG, Saad et al. 2000)
        return v - AW @ (Einv @ (W.T @ v)) if W is not None else v
    z = dinv * r; p = z.copy()
    if W is not None: p = p - W @ (Einv @ (AW.T @ p))
    rz = r @ z; b2 = b @ (dinv * b)
    for it in range(maxit):
        Ap = A @ p
        al = rz / (p @ Ap)
        x += al * p; r -= al * Ap
        z = dinv * r
        rz2 = r @ z
        if rz2 <= tol * tol * b2: return x, it + 1
        be = rz2 / rz; rz = rz2
        p = z + be * p
        if W is not None: p = p - W @ (Einv @ (AW.T @ p))
    return x, maxit

for lame, name in ((Lame(1e6, 0.3), 'E=1e6 nu=0.3'), (Lame.rubber(), 'rubber')):
    for n in (4, 8, 12):
        sc = scenes.Scene()
        offs = []
        for i in range(2):
            verts, tets = meshes.tet_blocks(n, n, n)
            verts = verts / n + np.array([0.013 * i, 1.3 * i, 0.007 * i])
            offs.append(sc.add_tet_mesh(verts, tets, lame, pkg.TET_LINEAR))
        sc.settings.update(linsolver=0)
        o = sc.make_oracle()
        Ah = o.A[0::3, :][:, 0::3].tocsr()       # one axis: diag(m) + Ahat
        nv = Ah.shape[0]
        dinv = 1.0 / Ah.diagonal()
        rng = np.random.default_rng(0)
        W = np.zeros((nv, 2)); W[:nv // 2, 0] = 1; W[nv // 2:, 1] = 1
        for bname, b in (('dense', Ah @ rng.standard_normal(nv)), ('sparse', np.eye(nv)[5] - 2 * np.eye(nv)[nv - 7])):
            x0, it0 = pcg(Ah, b, dinv, 1e-10)
            x1, it1 = pcg(Ah, b, dinv, 1e-10, W)
            xe = spla.spsolve(Ah.tocsc(), b)
            print(name, 'n', n, 'verts', nv, bname, 'jacobi its', it0, 'deflated its', it1,
                  'err %.1e %.1e' % (np.abs(x0 - xe).max() / np.abs(xe).max(), np.abs(x1 - xe).max() / np.abs(xe).max()), flush=True)
