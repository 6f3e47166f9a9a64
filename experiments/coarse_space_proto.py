"""Two-level preconditioners for the on-chip PCG on the unstructured body: compact blocks (recursive graph bisection) +
piecewise-constant coarse spaces of different sizes, additive vs deflated (A-DEF2), Jacobi vs multicolour block SGS."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'experiments')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.sparse.csgraph as csg
from admm_elastic_amd import meshes, capi
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = sys.argv[3] if len(sys.argv) > 3 else "blob1m_mix"
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
print('tets', nt, 'verts', nv, 'nnz/row', A.nnz / nv, flush=True)
dinv = 1.0 / A.diagonal()
b = A @ np.random.default_rng(0).standard_normal(nv)

def pcg(prec, tol=1e-8, maxit=3000, x0=None):
    x = np.zeros(nv) if x0 is None else x0.copy(); r = b - A @ x; z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b)
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        if r @ (dinv * r) <= tol * tol * b2: return it + 1
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit

def bisect(Ag, G):
    part = np.zeros(Ag.shape[0], dtype=np.int64)
    todo = [(np.arange(Ag.shape[0]), 0, G)]
    while todo:
        mem, base, g = todo.pop()
        if g == 1: part[mem] = base; continue
        sub = Ag[mem][:, mem]
        order = csg.breadth_first_order(sub, 0, directed=False, return_predecessors=False)
        order = csg.breadth_first_order(sub, order[-1], directed=False, return_predecessors=False)
        order = csg.breadth_first_order(sub, order[-1], directed=False, return_predecessors=False)
        if len(order) < len(mem):
            rest = np.setdiff1d(np.arange(len(mem)), order); order = np.concatenate([order, rest])
        g0 = g // 2; n0 = (len(mem) * g0 + g - 1) // g
        todo.append((mem[order[:n0]], base, g0)); todo.append((mem[order[n0:]], base + g0, g - g0))
    return part

Anz = A.copy(); Anz.data = np.where(A.data != 0, 1.0, 0.0); Anz.eliminate_zeros()
t = time.time(); fine = bisect(Anz, 8 * G); print('bisection %.1f s' % (time.time() - t), flush=True)
blk = fine // 8
coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
Ab = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
# block-local colouring
Abnz = Ab.copy(); Abnz.data = np.where(Ab.data != 0, 1.0, 0.0); Abnz.eliminate_zeros()
col, ncol = capi.greedy_coloring(Abnz.indptr.astype(np.int32), Abnz.indices.astype(np.int32))
print('block-local colours', ncol)
perm = np.lexsort((np.arange(nv), col, blk))
P = sp.csr_matrix((np.ones(nv), (np.arange(nv), perm)), shape=(nv, nv))
Abp = (P @ Ab @ P.T).tocsr()
Lp = sp.tril(Abp, 0).tocsr(); Up = sp.triu(Abp, 0).tocsr(); Dp = Abp.diagonal()
def mcsgs(r):
    rp_ = r[perm]
    y = spla.spsolve_triangular(Lp, rp_, lower=True)
    zp = spla.spsolve_triangular(Up, Dp * y, lower=False)
    z = np.empty_like(zp); z[perm] = zp
    return z
jac = lambda r: dinv * r
lu = spla.splu(Ab.tocsc())
print('Jacobi', pcg(jac), ' mcsgs', pcg(mcsgs), ' exact', pcg(lu.solve), flush=True)
for sub in (1, 2, 4, 8):
    agg = fine // (8 // sub); nc = agg.max() + 1
    Pc = sp.csr_matrix((np.ones(nv), (np.arange(nv), agg)), shape=(nv, nc))
    Ac = (Pc.T @ A @ Pc).toarray(); Aci = np.linalg.inv(Ac)
    Q = lambda r: Pc @ (Aci @ (Pc.T @ r))
    res = {}
    for name, M in (('jacobi', jac), ('mcsgs', mcsgs), ('exact', lu.solve)):
        res[name + '+add'] = pcg(lambda r: M(r) + Q(r))
        # A-DEF2: z = M (r - A Q r) + Q r, started from x0 = Q b
        res[name + '+def'] = pcg(lambda r: M(r - A @ Q(r)) + Q(r), x0=Q(b))
    print('coarse dofs %d (%d per block):' % (nc, sub), res, flush=True)
