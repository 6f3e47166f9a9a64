"""Chebyshev polynomial of D^-1 A as CG preconditioner (CPU prototype, scipy): iterations (= grid-wide reduction rounds
of the on-chip PCG) and matrix-vector products (= exchanges) against Jacobi-PCG, one axis of the n-cell cube system."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import scenes
import admm_elastic_amd as pkg

def pcg(A, b, prec, tol=1e-8, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z
    dinv = 1.0 / A.diagonal(); b2 = b @ (dinv * b)
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        if r @ (dinv * r) <= tol * tol * b2: return x, it + 1
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit

def cheb(A, dinv, m, lmin, lmax):
    # z = p_m(D^-1 A) D^-1 r : m Chebyshev iterations on D^-1 A z = D^-1 r from z = 0 (Saad, Iterative Methods, Alg. 12.1)
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    def apply(r):
        sigma = theta / delta; rho = 1.0 / sigma
        z = np.zeros_like(r); res = dinv * r; d = res / theta
        for k in range(m):
            z = z + d
            if k == m - 1: break
            res = res - dinv * (A @ d)
            rho_new = 1.0 / (2.0 * sigma - rho)
            d = rho_new * rho * d + (2.0 * rho_new / delta) * res
            rho = rho_new
        return z
    return apply

for n in (12, 20):
    sc = scenes.cube_scene(n, pkg.TET_NEOHOOKEAN, admm_iters=5)
    o = sc.make_oracle()
    Ah = o.A[0::3, :][:, 0::3].tocsr(); dinv = 1.0 / Ah.diagonal(); nv = Ah.shape[0]
    lmax = spla.eigsh(sp.diags(np.sqrt(dinv)) @ Ah @ sp.diags(np.sqrt(dinv)), k=1, which='LA', return_eigenvectors=False)[0]
    b = Ah @ np.random.default_rng(0).standard_normal(nv)
    _, it0 = pcg(Ah, b, lambda r: dinv * r)
    print('n %d verts %d  lmax(D^-1 A) %.3f  Jacobi-PCG: %d iterations = %d reductions, %d matvecs' % (n, nv, lmax, it0, it0, it0))
    for m in (2, 3, 4, 6):
        for ratio in (10, 30, 100):
            _, it = pcg(Ah, b, cheb(Ah, dinv, m, 1.05 * lmax / ratio, 1.05 * lmax))
            mv = it * m                                  # m - 1 inside the polynomial + 1 for A p
            est = it * (m * 3.6 + 4.6) / (it0 * 8.2)     # exchanges 3.6 us, reduction round 4.6 us, plain iteration 8.2 us
            print('   degree %d  interval lmax/%-3d: %3d reductions  %4d matvecs (x%.2f)   estimated time x%.2f' % (m, ratio, it, mv, mv / it0, est))
