"""Round 5: preconditioner variants that keep k_pcg2's synchronisation count, on the bench body with the library's OWN 256 blocks
(CPU, scipy).  Baseline = the kernel: additive two-level, affine coarse space, two Chebyshev steps on the block-diagonal part.
Variants: (a) A-DEF2 deflation-type combination  z = Q r + S (r - A Q r)  (needs (A P) y on the block's rows: the coarse coefficients
of the neighbour blocks, no extra exchange of vectors);  (b) one layer of overlap in the block smoother (restricted additive
Schwarz: needs the halo values of r, which the product's exchange already delivers for m but not for n);  (c) 3 sub-aggregates per
block with affine functions (12 per block).  Iterations to 1e-6 from zero, random solution, and per decade over the second half.
    python experiments/deflated_precond_proto.py [n=118]"""
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

def lam_max(M):
    v = np.random.default_rng(1).standard_normal(nv)
    for _ in range(60):
        w = dinv * (M @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
    return lam

def cheb(M, lam, deg, ratio=16.0):
    lo = lam / ratio; th = 0.5 * (lam + lo); de = 0.5 * (lam - lo)
    def S(r):
        z = np.zeros_like(r); res = r.copy(); p = None; alpha = beta = 0.0
        for k in range(deg):
            y = dinv * res
            if k == 0: p = y; alpha = 1.0 / th
            else:
                beta = (de * alpha / 2.0) ** 2 if k > 1 else 0.5 * (de * alpha) ** 2
                alpha = 1.0 / (th - beta / alpha); p = y + beta * p
            z = z + alpha * p; res = res - alpha * (M @ p)
        return z
    return S

def coarse(funcs_of, agg):
    cols, vals, rws = [], [], []; nc = 0
    for b in range(agg.max() + 1):
        idx = np.nonzero(agg == b)[0]
        if len(idx) == 0: continue
        F = funcs_of(X[idx] - X[idx].mean(axis=0))
        for j in range(F.shape[1]):
            rws.append(idx); cols.append(np.full(len(idx), nc)); vals.append(F[:, j]); nc += 1
    P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rws), np.concatenate(cols))), shape=(nv, nc))
    Ac = (P.T @ A @ P).toarray(); Aci = np.linalg.inv(Ac + 1e-12 * np.trace(Ac) / nc * np.eye(nc))
    return P, Aci, nc

def pcg(prec, b, x0=None, tol=1e-6, maxit=400):
    x = np.zeros(nv) if x0 is None else x0.copy(); r = b - A @ x; z = prec(r); p = z.copy(); rz = r @ z; b2 = b @ (dinv * b); hist = []
    for it in range(maxit):
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        q = r @ (dinv * r) / b2; hist.append(q)
        if q <= tol * tol: return it + 1, hist
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit, hist

def report(name, it, h, t):
    k0 = len(h) // 2; rate = (len(h) - k0) / max(1e-9, 0.5 * (np.log10(h[k0]) - np.log10(h[-1])))
    print("%-64s %3d iterations to 1e-6, %.1f per decade  (%.0f s)" % (name, it, rate, time.time() - t), flush=True)

rng = np.random.default_rng(0)
b = A @ rng.standard_normal(nv)
affine = lambda Y: np.column_stack([np.ones(len(Y)), Y])
lam = 1.1 * lam_max(Abb)
P, Aci, nc = coarse(affine, blk)
Q = lambda r: P @ (Aci @ (P.T @ r))
for deg in (2, 3):
    S = cheb(Abb, lam, deg)
    t = time.time(); it, h = pcg(lambda r: S(r) + Q(r), b); report("additive, affine, Chebyshev %d (the kernel at 2)" % deg, it, h, t)
    t = time.time(); it, h = pcg(lambda r: (lambda q: q + S(r - A @ q))(Q(r)), b, x0=Q(b)); report("A-DEF2  z = Qr + S(r - AQr), x0 = Qb, Chebyshev %d" % deg, it, h, t)
    # symmetric variant with the same cost class: z = Qr + S(r - AQr) - Q A S (r - A Q r)  (BNN without the pre-smoothing)
    def bnn(r):
        q = Q(r); y = S(r - A @ q); return q + y - Q(A @ y)
    t = time.time(); it, h = pcg(bnn, b, x0=Q(b)); report("BNN-type  (I - QA) S (I - AQ) + Q, Chebyshev %d" % deg, it, h, t)

# (b) one layer of overlap: every block's smoother acts on its rows + their neighbours, result kept on its own rows (RAS)
Apat = A.tocsr()
def ras(deg):
    Ss = []
    own = [np.nonzero(blk == g)[0] for g in range(G)]
    ext = []
    for g in range(G):
        nb = np.unique(Apat[own[g]].indices); ext.append(nb)
    subs = [Apat[e][:, e].tocsr() for e in ext]
    lam_o = 1.1 * max(np.abs(sp.linalg.eigsh(sp.diags(dinv[e] ** 0.5) @ M @ sp.diags(dinv[e] ** 0.5), k=1, which='LA', return_eigenvectors=False, tol=1e-3)[0]) for e, M in list(zip(ext, subs))[::16])
    lo = lam_o / 16.0; th = 0.5 * (lam_o + lo); de = 0.5 * (lam_o - lo)
    def S(r):
        out = np.zeros_like(r)
        for g in range(G):
            e = ext[g]; M = subs[g]; di = dinv[e]; rr = r[e]
            z = np.zeros_like(rr); res = rr.copy(); p = None; alpha = beta = 0.0
            for k in range(deg):
                y = di * res
                if k == 0: p = y; alpha = 1.0 / th
                else:
                    beta = (de * alpha / 2.0) ** 2 if k > 1 else 0.5 * (de * alpha) ** 2
                    alpha = 1.0 / (th - beta / alpha); p = y + beta * p
                z = z + alpha * p; res = res - alpha * (M @ p)
            pos = np.searchsorted(e, own[g]); out[own[g]] = z[pos]
        return out
    return S
for deg in (2, 3):
    S = ras(deg)
    t = time.time(); it, h = pcg(lambda r: S(r) + Q(r), b); report("additive + restricted overlap (1 layer), Chebyshev %d  [nonsymmetric]" % deg, it, h, t)
    t = time.time(); it, h = pcg(lambda r: (lambda q: q + S(r - A @ q))(Q(r)), b, x0=Q(b)); report("A-DEF2 + restricted overlap (1 layer), Chebyshev %d" % deg, it, h, t)

# (c) 3 sub-aggregates per block (split along the block's longest principal axis), affine on each: 12 functions per block
agg3 = np.zeros(nv, np.int64)
for g in range(G):
    idx = np.nonzero(blk == g)[0]; Y = X[idx] - X[idx].mean(axis=0)
    w, V = np.linalg.eigh(Y.T @ Y); t = Y @ V[:, -1]; o = np.argsort(t)
    part = np.zeros(len(idx), np.int64); part[o[len(o) // 3: 2 * len(o) // 3]] = 1; part[o[2 * len(o) // 3:]] = 2
    agg3[idx] = 3 * g + part
P3, Aci3, nc3 = coarse(affine, agg3)
Q3 = lambda r: P3 @ (Aci3 @ (P3.T @ r))
S = cheb(Abb, lam, 2)
t = time.time(); it, h = pcg(lambda r: S(r) + Q3(r), b); report("additive, affine on 3 sub-aggregates per block (%d), Chebyshev 2" % nc3, it, h, t)
t = time.time(); it, h = pcg(lambda r: (lambda q: q + S(r - A @ q))(Q3(r)), b, x0=Q3(b)); report("A-DEF2, affine on 3 sub-aggregates per block, Chebyshev 2", it, h, t)
