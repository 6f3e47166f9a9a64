"""Preconditioners of the on-chip PCG on the UNSTRUCTURED 1 M-tet body (meshes.unstructured_blob): iteration counts
(PCG to 1e-8, b = A randn) for block shapes (index strips after RCM vs compact blocks from recursive graph bisection)
x block-local methods (Jacobi, multicolour symmetric Gauss-Seidel, exact) (+ piecewise-constant coarse space)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.sparse.csgraph as csg
from admm_elastic_amd import meshes, capi
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
print('tets', nt, 'verts', nv, 'nnz/row', A.nnz / nv, flush=True)
dinv = 1.0 / A.diagonal()
b = A @ np.random.default_rng(0).standard_normal(nv)

def pcg(prec, tol=1e-8, maxit=3000):
    x = np.zeros(nv); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b)
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        if r @ (dinv * r) <= tol * tol * b2: return it + 1
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit

def bisect(Ag, G):
    """recursive graph bisection by BFS level structures from a pseudo-peripheral vertex: part id per vertex"""
    part = np.zeros(Ag.shape[0], dtype=np.int64)
    todo = [(np.arange(Ag.shape[0]), 0, G)]
    while todo:
        mem, base, g = todo.pop()
        if g == 1: part[mem] = base; continue
        sub = Ag[mem][:, mem]
        # pseudo-peripheral start: two BFS passes
        order = csg.breadth_first_order(sub, 0, directed=False, return_predecessors=False)
        far = order[-1]
        order = csg.breadth_first_order(sub, far, directed=False, return_predecessors=False)
        far = order[-1]
        order = csg.breadth_first_order(sub, far, directed=False, return_predecessors=False)
        if len(order) < len(mem):   # disconnected: append the rest
            rest = np.setdiff1d(np.arange(len(mem)), order); order = np.concatenate([order, rest])
        g0 = g // 2; n0 = (len(mem) * g0 + g - 1) // g
        todo.append((mem[order[:n0]], base, g0)); todo.append((mem[order[n0:]], base + g0, g - g0))
    return part

Ag = sp.csr_matrix((np.ones_like(A.data), A.indices, A.indptr), shape=A.shape)
Ap_ = A.copy(); Ap_.data = np.where(A.data != 0, 1.0, 0.0); Ap_.eliminate_zeros()
col, ncol = capi.greedy_coloring(Ap_.indptr.astype(np.int32), Ap_.indices.astype(np.int32))
print('colours', ncol, flush=True)

def block_methods(blk, name):
    coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
    Ab = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
    print(name, ': local fraction of nnz %.3f' % (keep.sum() / len(keep)), 'neighbour blocks max/mean',
          end=' ')
    B = sp.csr_matrix((np.ones(len(coo.row)), (blk[coo.row], blk[coo.col]))); B.sum_duplicates()
    nb = np.diff(B.indptr) - 1; print(nb.max(), nb.mean(), flush=True)
    perm = np.lexsort((np.arange(nv), col, blk))
    P = sp.csr_matrix((np.ones(nv), (np.arange(nv), perm)), shape=(nv, nv))
    Abp = (P @ Ab @ P.T).tocsr()
    Lp = sp.tril(Abp, 0).tocsr(); Up = sp.triu(Abp, 0).tocsr(); Dp = Abp.diagonal()
    def mc_ssor(r):
        rp_ = r[perm]
        y = spla.spsolve_triangular(Lp, rp_, lower=True)
        zp = spla.spsolve_triangular(Up, Dp * y, lower=False)
        z = np.empty_like(zp); z[perm] = zp
        return z
    lu = spla.splu(Ab.tocsc())
    res = dict(mcsgs=pcg(mc_ssor), exact=pcg(lu.solve))
    # + piecewise-constant coarse space (additive)
    nc = blk.max() + 1
    Pc = sp.csr_matrix((np.ones(nv), (np.arange(nv), blk)), shape=(nv, nc)); Aci = np.linalg.inv((Pc.T @ A @ Pc).toarray())
    res['mcsgs+coarse'] = pcg(lambda r: mc_ssor(r) + Pc @ (Aci @ (Pc.T @ r)))
    res['jacobi+coarse'] = pcg(lambda r: dinv * r + Pc @ (Aci @ (Pc.T @ r)))
    print('   ', res, flush=True)

print('Jacobi', pcg(lambda r: dinv * r), flush=True)
rows_pb = ((nv + 63) // 64 + G - 1) // G * 64
block_methods(np.arange(nv) // rows_pb, 'index strips (RCM order), %d rows' % rows_pb)
t = time.time(); part = bisect(Ag, G); print('bisection %.1f s' % (time.time() - t), 'sizes', np.bincount(part).min(), np.bincount(part).max())
block_methods(part, 'graph bisection blocks')
