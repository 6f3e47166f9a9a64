"""Round 4: what would a richer coarse space / a stronger block smoother buy k_pcg2 on the bench body?  CPU (scipy), the library's OWN
blocks (admm_host_oc_plan: 256 compact blocks), Chebyshev block smoother as in the kernel, additive two-level PCG.  Iterations for a
1e-6 reduction of the Jacobi-norm residual from a zero start, random right-hand side -- the ASYMPTOTIC rate, which is what a solve
consists of once the recycled projection has taken the easy part.   python experiments/coarse_space_proto2.py [n=118]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 118
sc, nt, nv = bench.build_scene(dict(bench.WORKLOADS["blob1m_mix"], linsolver=0), n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
G = 256 if nv > 100000 else max(4, nv // 700)
spb = -(-nv // (64 * G))
plan = s.host_oc_plan(G, spb, settings=sc.product_settings, coarse=False)
rv = plan["row_vertex"]; blk = np.zeros(nv, np.int64)
rows = np.nonzero(rv >= 0)[0]; blk[rv[rows]] = rows // (64 * spb)
print("tets", nt, "verts", nv, "blocks", G, "rows/block", np.bincount(blk).max(), flush=True)
d = A.diagonal(); dinv = 1.0 / d
coo = A.tocoo(); keep = blk[coo.row] == blk[coo.col]
Abb = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape)
X = sc.x

def lam_max():
    v = np.random.default_rng(1).standard_normal(nv)
    for _ in range(60):
        w = dinv * (Abb @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
    return lam
lam = 1.1 * lam_max()

def cheb(deg, ratio=16.0):
    """z = p(D^-1 A_bb) D^-1 r, p = Chebyshev of degree deg - 1 on [lam / ratio, lam]"""
    lo = lam / ratio; th = 0.5 * (lam + lo); de = 0.5 * (lam - lo)
    def S(r):
        # standard Chebyshev iteration for A_bb z = r with Jacobi scaling, deg steps from z = 0
        z = np.zeros_like(r); res = r.copy(); p = None; alpha = beta = 0.0
        for k in range(deg):
            y = dinv * res
            if k == 0: p = y; alpha = 1.0 / th
            else:
                beta = (de * alpha / 2.0) ** 2 if k > 1 else 0.5 * (de * alpha) ** 2
                alpha = 1.0 / (th - beta / alpha); p = y + beta * p
            z = z + alpha * p; res = res - alpha * (Abb @ p)
        return z
    return S

def coarse(funcs_of):
    cols, vals, rws = [], [], []; nc = 0
    for b in range(G):
        idx = np.nonzero(blk == b)[0]
        if len(idx) == 0: continue
        F = funcs_of(X[idx] - X[idx].mean(axis=0))
        for j in range(F.shape[1]):
            rws.append(idx); cols.append(np.full(len(idx), nc)); vals.append(F[:, j]); nc += 1
    P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rws), np.concatenate(cols))), shape=(nv, nc))
    Ac = (P.T @ A @ P).toarray(); Aci = np.linalg.inv(Ac + 1e-12 * np.trace(Ac) / nc * np.eye(nc))
    return P, Aci, nc

def pcg(prec, b, tol=1e-6, maxit=400):
    x = np.zeros(nv); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b); hist = []
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        q = r @ (dinv * r) / b2; hist.append(q)
        if q <= tol * tol: return it + 1, hist
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit, hist

rng = np.random.default_rng(0)
b = A @ rng.standard_normal(nv)
const = lambda Y: np.ones((len(Y), 1))
affine = lambda Y: np.column_stack([np.ones(len(Y)), Y])
quad = lambda Y: np.column_stack([np.ones(len(Y)), Y, Y[:, 0] * Y[:, 1], Y[:, 1] * Y[:, 2], Y[:, 2] * Y[:, 0], Y[:, 0] ** 2, Y[:, 1] ** 2, Y[:, 2] ** 2])
spaces = {"affine 4/block": coarse(affine), "quadratic 10/block": coarse(quad)}
for sname, (P, Aci, nc) in spaces.items():
    Q = lambda r, P=P, Aci=Aci: P @ (Aci @ (P.T @ r))
    for deg in (2, 3, 4):
        S = cheb(deg)
        t = time.time(); it, h = pcg(lambda r: S(r) + Q(r), b)
        # asymptotic rate: iterations per decade of the NORM over the second half
        k0 = len(h) // 2; rate = (len(h) - k0) / max(1e-9, 0.5 * (np.log10(h[k0]) - np.log10(h[-1])))
        print("%-20s (%4d coarse dofs)  Chebyshev steps %d: %3d iterations to 1e-6, %.1f per decade  (%.0f s)" % (sname, nc, deg, it, rate, time.time() - t), flush=True)
