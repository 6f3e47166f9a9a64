"""Block-Jacobi with banded blocks of 64 consecutive rows (what one wavefront could solve in registers)."""
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv)>1 else 30
w = bench.WORKLOADS["cube1m_nh"]
sc, nt, nv = bench.build_scene(w, n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
rng = np.random.default_rng(0); b = A @ rng.standard_normal(nv); d = A.diagonal()
def pcg(apply_M, tol=1e-8, maxit=3000):
    x = np.zeros(nv); r = b.copy(); z = apply_M(r); p = z.copy(); rz = r@z; rz0 = b@apply_M(b)
    for it in range(maxit):
        Ap = A@p; al = rz/(p@Ap); x += al*p; r -= al*Ap; z = apply_M(r); rzn = r@z
        if rzn <= tol*tol*rz0: return it+1
        p = z + (rzn/rz)*p; rz = rzn
    return maxit
print("n", n, "nv", nv, "jacobi", pcg(lambda r: r/d))
coo = A.tocoo()
for blk, band in ((64, 1), (64, 63), (256, 255)):
    keep = (coo.row // blk == coo.col // blk) & (np.abs(coo.row - coo.col) <= band)
    M = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
    lu = spla.splu(M)
    print(" block", blk, "band", band, "its", pcg(lambda r: lu.solve(r)))
