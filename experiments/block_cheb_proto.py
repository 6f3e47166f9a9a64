"""Block-local polynomial smoothers next to the aggregate coarse space (the on-chip PCG's two-level preconditioner):
M^-1 = q_m(A_bb) + P A_c^-1 P^T with q_m = Chebyshev approximation of the inverse of the block-diagonal part A_bb (entries
whose row and column sit in the same block: data the block holds in LDS, no exchange).  CG iterations to 1e-8 on the
unstructured body.  python experiments/block_cheb_proto.py [n] [G] [workload]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'experiments')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.sparse.csgraph as csg
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "full" else None
G = int(sys.argv[2]) if len(sys.argv) > 2 else 240
wl = sys.argv[3] if len(sys.argv) > 3 else "blob1m_mix"
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
print('tets', nt, 'verts', nv, 'nnz/row', A.nnz / nv, flush=True)
d = A.diagonal(); dinv = 1.0 / d
rng = np.random.default_rng(0)
b = A @ rng.standard_normal(nv)

def pcg(prec, tol=1e-8, maxit=3000):
    x = np.zeros(nv); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b)
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
t = time.time(); fine = bisect(Anz, 4 * G); print('bisection %.1f s' % (time.time() - t), flush=True)
blk = fine // 4
coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
Ab = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
print('block-local share of the non-zeros %.3f' % (Ab.nnz / A.nnz))
Pc = sp.csr_matrix((np.ones(nv), (np.arange(nv), fine)), shape=(nv, fine.max() + 1))
Aci = np.linalg.inv((Pc.T @ A @ Pc).toarray())
Q = lambda r: Pc @ (Aci @ (Pc.T @ r))
jac = lambda r: dinv * r
# spectrum of D^-1 A_bb
Sb = sp.diags(np.sqrt(dinv)) @ Ab @ sp.diags(np.sqrt(dinv))
lmax = spla.eigsh(Sb, k=1, which='LA', return_eigenvectors=False)[0]
print('lambda_max(D^-1 A_bb) = %.3f' % lmax)

def cheb(m, ratio, lmax_used):
    """q_m(A_bb) r: m steps of the Chebyshev iteration for A_bb z = r from z = 0 with Jacobi inside, on [lmax/ratio, lmax]"""
    lo, hi = lmax_used / ratio, lmax_used
    th, de = 0.5 * (hi + lo), 0.5 * (hi - lo)
    def apply(r):
        # standard three-term Chebyshev semi-iteration (Saad, Alg. 12.1) on D^-1 A_bb
        z = np.zeros_like(r); res = r.copy()
        sig = th / de; rho = 1.0 / sig
        dvec = (1.0 / th) * (dinv * res)
        for k in range(m):
            z = z + dvec
            if k == m - 1: break
            res = res - Ab @ dvec
            rho_n = 1.0 / (2.0 * sig - rho)
            dvec = rho_n * rho * dvec + (2.0 * rho_n / de) * (dinv * res)
            rho = rho_n
        return z
    return apply

lu = spla.splu(Ab.tocsc())
print('jacobi only', pcg(jac), flush=True)
print('jacobi + coarse', pcg(lambda r: jac(r) + Q(r)), '   exact block solve + coarse', pcg(lambda r: lu.solve(r) + Q(r)), flush=True)
for m in (2, 3, 4):
    row = {}
    for ratio in (4, 8, 16, 30):
        c = cheb(m, ratio, 1.05 * lmax)
        row[ratio] = pcg(lambda r: c(r) + Q(r))
    print('Chebyshev degree %d (%d local products per application) + coarse, by lmax/lmin:' % (m, m - 1), row, flush=True)
# what a guaranteed bound costs: Gershgorin on D^-1 A_bb instead of the true lambda_max
gersh = (abs(Ab).sum(axis=1).A1 * dinv).max()
print('Gershgorin bound %.3f (true %.3f)' % (gersh, lmax))
for hi in (gersh, 1.1 * lmax):
    c = cheb(2, 8, hi)
    print('degree 2, ratio 8, hi = %.3f:' % hi, pcg(lambda r: c(r) + Q(r)), flush=True)
# the closed form the kernel uses: z = D^-1 ((alpha - beta) r - beta offdiag(A_bb) D^-1 r)
hi = 1.1 * lmax; lo = hi / 8; th, de = 0.5 * (hi + lo), 0.5 * (hi - lo); sig = th / de; r0 = 1 / sig; r1 = 1 / (2 * sig - r0)
al = (1 + r1 * r0) / th + 2 * r1 / de; be = 2 * r1 / (de * th)
Aoff = Ab - sp.diags(Ab.diagonal())
closed = lambda r: dinv * ((al - be) * r - be * (Aoff @ (dinv * r)))
v = rng.standard_normal(nv)
print('closed form vs iteration:', np.abs(closed(v) - cheb(2, 8, hi)(v)).max() / np.abs(closed(v)).max(), ' alpha %.4f beta %.4f' % (al, be))
