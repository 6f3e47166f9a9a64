"""Block-local preconditioners for the on-chip PCG (CPU prototype, scipy): blocks = the contiguous row ranges the kernel's
256 blocks own (686 rows at 1 M tets).  No exchange is needed to apply them.  Iterations of PCG at 1e-8 for: Jacobi,
block-SSOR (symmetric Gauss-Seidel on the diagonal block, 1 sweep), block-exact (the limit of any block-local method)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
G = 256
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], n)
o = sc.make_oracle(big=True) if n > 20 else sc.make_oracle()
A = o.A[0::3, :][:, 0::3].tocsr(); nv = A.shape[0]
rows_pb = -(-(-(-nv // 64)) // G) * 64 if False else ((nv + 63) // 64 + G - 1) // G * 64
print('verts', nv, 'rows per block', rows_pb, flush=True)
dinv = 1.0 / A.diagonal()
b = A @ np.random.default_rng(0).standard_normal(nv)

def pcg(prec, tol=1e-8, maxit=3000):
    x = np.zeros(nv); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b)
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        if r @ (dinv * r) <= tol * tol * b2: return it + 1
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit

# block-diagonal part of A
blk = np.arange(nv) // rows_pb
coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
Ab = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
L = sp.tril(Ab, 0).tocsr(); U = sp.triu(Ab, 0).tocsr(); D = Ab.diagonal()
def ssor(r):            # (D + L) D^-1 (D + U) z = r   (symmetric Gauss-Seidel, omega = 1)
    y = spla.spsolve_triangular(L, r, lower=True)
    return spla.spsolve_triangular(U, D * y, lower=False)
lu = spla.splu(Ab.tocsc())
print('Jacobi      ', pcg(lambda r: dinv * r), flush=True)
t = time.time(); print('block-SSOR  ', pcg(ssor), '(%.0f s)' % (time.time() - t), flush=True)
print('block-exact ', pcg(lu.solve), flush=True)
# global SSOR for reference (needs exchanges: not block-local)
Lg = sp.tril(A, 0).tocsr(); Ug = sp.triu(A, 0).tocsr(); Dg = A.diagonal()
print('global SSOR ', pcg(lambda r: spla.spsolve_triangular(Ug, Dg * spla.spsolve_triangular(Lg, r, lower=True), lower=False)), flush=True)

# multicolour ordering inside the blocks (what a GPU block can do in parallel): symmetric Gauss-Seidel with the rows
# of a block swept colour by colour
from admm_elastic_amd import capi
Ap = A.copy(); Ap.data = np.where(A.data != 0, 1.0, 0.0); Ap.eliminate_zeros()
col, ncol = capi.greedy_coloring(Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32))
print('colours', ncol, flush=True)
# permutation: by (block, colour, index); SSOR in that order restricted to the block-diagonal part
perm = np.lexsort((np.arange(nv), col, blk))
P = sp.csr_matrix((np.ones(nv), (np.arange(nv), perm)), shape=(nv, nv))
Abp = (P @ Ab @ P.T).tocsr()
Lp = sp.tril(Abp, 0).tocsr(); Up = sp.triu(Abp, 0).tocsr(); Dp = Abp.diagonal()
def mc_ssor(r):
    rp = r[perm]
    y = spla.spsolve_triangular(Lp, rp, lower=True)
    zp = spla.spsolve_triangular(Up, Dp * y, lower=False)
    z = np.empty_like(zp); z[perm] = zp
    return z
print('block multicolour SSOR', pcg(mc_ssor), flush=True)
