"""Offline: the first solves of a frame have no (or few) recycled pairs; would filling the free slots of the 4-pair Galerkin
projection with the LOWEST EIGENVECTORS of D^-1 A (computed once: A is constant) cut their iteration counts?  Joint projection
(one Gram system over recycled + static pairs), two-level (aggregate coarse space) or Jacobi PCG, the oracle's exact ADMM
trajectory.  usage: python experiments/static_pairs_proto.py [n=24] [G=32] [workload]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'experiments')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.sparse.csgraph as csg
import scenes, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
wl = sys.argv[3] if len(sys.argv) > 3 else "cube1m_mix"
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], n)
o = sc.make_oracle(mode=1, big=True)
A3 = o.A.tocsr()
As = A3[0::3, 0::3].tocsr()
d = As.diagonal(); dinv = 1.0 / d
print("nv", nv, "nt", nt, flush=True)

def bisect(Ag, G):
    part = np.zeros(Ag.shape[0], dtype=np.int64)
    todo = [(np.arange(Ag.shape[0]), 0, G)]
    while todo:
        mem, base, g = todo.pop()
        if g == 1: part[mem] = base; continue
        sub = Ag[mem][:, mem]
        order = csg.breadth_first_order(sub, 0, directed=False, return_predecessors=False)
        order = csg.breadth_first_order(sub, order[-1], directed=False, return_predecessors=False)
        if len(order) < len(mem):
            rest = np.setdiff1d(np.arange(len(mem)), order); order = np.concatenate([order, rest])
        g0 = g // 2; n0 = (len(mem) * g0 + g - 1) // g
        todo.append((mem[order[:n0]], base, g0)); todo.append((mem[order[n0:]], base + g0, g - g0))
    return part
Anz = As.copy(); Anz.data[:] = 1.0
agg = bisect(Anz, 4 * G)
Pc = sp.csr_matrix((np.ones(nv), (np.arange(nv), agg)), shape=(nv, agg.max() + 1))
Aci = np.linalg.inv((Pc.T @ As @ Pc).toarray())
M2 = lambda r: dinv * r + Pc @ (Aci @ (Pc.T @ r))

def pcg(b, x, prec, tol=1e-8, maxit=3000):
    r = b - As @ x; gb = b @ (dinv * b)
    if r @ (dinv * r) <= tol * tol * gb: return x, 0
    u = prec(r); g = r @ u; p = u.copy(); it = 0
    while it < maxit:
        s = As @ p; al = g / (p @ s)
        x = x + al * p; r = r - al * s; it += 1
        if r @ (dinv * r) <= tol * tol * gb: break
        u = prec(r); gn = r @ u
        p = u + (gn / g) * p; g = gn
    return x, it

Dh = sp.diags(np.sqrt(dinv))
lam, V = spla.eigsh((Dh @ As @ Dh).tocsc(), k=8, sigma=0.0, which='LM')
Z = Dh @ V[:, np.argsort(lam)]
AZ = As @ Z
print("lowest eigenvalues of D^-1 A:", np.sort(lam)[:4], flush=True)

for frame in range(3):
    tr = []
    o.step(trace=tr)
    if frame < 1: continue
    xs = [t[3] for t in tr]; bs = [t[2] for t in tr]
    for name, prec in (("two-level", M2), ("jacobi", lambda r: dinv * r)):
        for nstat in (0, 4, 8):
            E = []; R = []; out = []
            for s in range(1, len(tr)):
                its = []
                for ax in range(3):
                    b = bs[s][ax::3]; x = xs[s - 1][ax::3].copy()
                    r0 = b - As @ x
                    m = min(4, len(E))
                    cols_E = [e[ax::3] for e in E[-m:]] if m else []
                    cols_R = [r[ax::3] for r in R[-m:]] if m else []
                    free = (4 if nstat == 4 else 8 if nstat == 8 else 0)
                    ks = max(0, (4 - m) if nstat == 4 else (8 - m) if nstat == 8 else 0)
                    for k in range(ks): cols_E.append(Z[:, k]); cols_R.append(AZ[:, k])
                    if cols_E:
                        Em = np.array(cols_E).T; Rm = np.array(cols_R).T
                        Gm = Em.T @ Rm; cc = np.linalg.lstsq(0.5 * (Gm + Gm.T), Em.T @ r0, rcond=None)[0]
                        x = x + Em @ cc
                    x, it = pcg(b, x, prec)
                    its.append(it)
                out.append(max(its))
                e = xs[s] - xs[s - 1]; E.append(e); R.append(A3 @ e)
            print("frame", frame, "%-9s" % name, "static slots up to %d:" % nstat, out, " sum", sum(out), flush=True)
