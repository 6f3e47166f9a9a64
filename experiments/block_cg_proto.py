"""Would block CG over the three axes (same matrix, three right-hand sides) need fewer iterations than three independent CGs?"""
import os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np, scipy.sparse as sp, time
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
sc, nt, nv = bench.build_scene(dict(bench.WORKLOADS["blob1m_mix"], linsolver=0), n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m[::3] if len(sc.m) == 3*nv else sc.m)).tocsr()
G = 256 if nv > 100000 else max(4, nv // 700)
spb = -(-nv // (64 * G))
plan = s.host_oc_plan(G, spb, settings=sc.product_settings, coarse=False)
rv = plan["row_vertex"]; blk = np.zeros(nv, np.int64)
rows = np.nonzero(rv >= 0)[0]; blk[rv[rows]] = rows // (64 * spb)
d = A.diagonal(); dinv = 1.0 / d
coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
Abb = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
X = sc.x.reshape(-1, 3)
v = np.random.default_rng(1).standard_normal(nv)
for _ in range(60):
    w = dinv * (Abb @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
lam *= 1.1
lo = lam / 16; th = 0.5 * (lam + lo); de = 0.5 * (lam - lo)
def S(r):   # two Chebyshev steps, r: (nv, k)
    z = np.zeros_like(r); res = r.copy(); p = None; alpha = 0.0
    for k in range(2):
        y = dinv[:, None] * res
        if k == 0: p = y; alpha = 1.0 / th
        else:
            beta = 0.5 * (de * alpha) ** 2; alpha = 1.0 / (th - beta / alpha); p = y + beta * p
        z = z + alpha * p; res = res - alpha * (Abb @ p)
    return z
cols, vals, rws = [], [], []; nc = 0
for b in range(G):
    idx = np.nonzero(blk == b)[0]
    if len(idx) == 0: continue
    Y = X[idx] - X[idx].mean(axis=0); F = np.column_stack([np.ones(len(idx)), Y])
    for j in range(4): rws.append(idx); cols.append(np.full(len(idx), nc)); vals.append(F[:, j]); nc += 1
P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rws), np.concatenate(cols))), shape=(nv, nc))
Ac = (P.T @ A @ P).toarray(); Aci = np.linalg.inv(Ac + 1e-12 * np.trace(Ac) / nc * np.eye(nc))
M = lambda R: S(R) + P @ (Aci @ (P.T @ R))
def pcg_indep(B, tol):
    its = []
    for j in range(B.shape[1]):
        b = B[:, j:j+1]; x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = (r*z).sum(); b2 = (b*dinv[:,None]*b).sum()
        for it in range(1000):
            Ap = A @ p; al = rz / (p*Ap).sum(); x += al*p; r -= al*Ap
            if (r*dinv[:,None]*r).sum() <= tol*tol*b2: break
            z = M(r); rz2 = (r*z).sum(); p = z + (rz2/rz)*p; rz = rz2
        its.append(it+1)
    return its
def pcg_block(B, tol):
    X_ = np.zeros_like(B); R = B.copy(); Z = M(R); Pm = Z.copy(); g = R.T @ Z; b2 = (B*dinv[:,None]*B).sum(axis=0)
    for it in range(1000):
        AP = A @ Pm; al = np.linalg.solve(Pm.T @ AP, g); X_ += Pm @ al; R -= AP @ al
        if np.all((R*dinv[:,None]*R).sum(axis=0) <= tol*tol*b2): break
        Z = M(R); g2 = R.T @ Z; be = np.linalg.solve(g, g2); Pm = Z + Pm @ be; g = g2
    return it+1
rng = np.random.default_rng(0)
print("nv", nv, "blocks", G)
for name, B in (("random x", A @ rng.standard_normal((nv, 3))), ("smooth + rough", A @ (np.sin(3*X) + 0.01*rng.standard_normal((nv,3)))), ):
    for tol in (1e-3, 1e-6):
        print(name, "tol", tol, "independent", pcg_indep(B, tol), "block", pcg_block(B, tol), flush=True)
