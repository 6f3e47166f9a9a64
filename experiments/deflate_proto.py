"""Offline: would projecting the initial residual of every global solve onto the K lowest eigenvectors of
D^-1/2 A D^-1/2 (computed once; A is constant) cut the Jacobi-PCG iteration counts?  Emulates the GPU's solve
sequence (warm start = previous iterate, recycled Galerkin projection on the last 4 exact pairs, stop rule
r.D^-1 r <= tol^2 b.D^-1 b on every axis) on the oracle's exact ADMM trajectory.
usage: python experiments/deflate_proto.py [n=16] [K list]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import scenes, bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Ks = [int(k) for k in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 8, 32]
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], n)
o = sc.make_oracle(mode=1, big=True)
A3 = o.A.tocsr()
As = A3[0::3, 0::3].tocsr()          # scalar system (masses are equal per axis)
d = As.diagonal(); dinv = 1.0 / d
print("nv", nv, "nt", nt)

def pcg(b, x, tol=1e-8, maxit=2000):
    r = b - As @ x; u = dinv * r
    gb = b @ (dinv * b); g = r @ u
    if g <= tol * tol * gb: return x, 0
    p = u.copy(); it = 0
    while it < maxit:
        s = As @ p; al = g / (p @ s)
        x = x + al * p; r = r - al * s; u = dinv * r
        gn = r @ u; it += 1
        if gn <= tol * tol * gb: break
        p = u + (gn / g) * p; g = gn
    return x, it

Kmax = max(Ks)
if Kmax:
    Dh = sp.diags(np.sqrt(dinv))
    S = (Dh @ As @ Dh).tocsc()
    lam, V = spla.eigsh(S, k=Kmax, sigma=0.0, which='LM')
    order = np.argsort(lam); lam = lam[order]; V = V[:, order]
    Z = Dh @ V                         # A-orthogonal: Z^T A Z = diag(lam)
    print("lowest eigenvalues of D^-1 A:", lam[:4], "...", lam[-1], " lambda_max <= 2")

for frame in range(4):
    tr = []
    o.step(trace=tr)
    if frame < 2: continue
    xs = [t[3] for t in tr]; bs = [t[2] for t in tr]
    res = {K: [] for K in Ks}
    for K in Ks:
        E = []; R = []
        for s in range(1, len(tr)):
            its = []
            for ax in range(3):
                b = bs[s][ax::3]; x = xs[s - 1][ax::3].copy()
                r0 = b - As @ x
                if K:   # eigen init-deflation
                    c = (Z[:, :K].T @ r0) / lam[:K]
                    x = x + Z[:, :K] @ c
                m = min(4, len(E))
                if m:   # recycled pairs (exact), per axis
                    Em = np.array([e[ax::3] for e in E[-m:]]).T; Rm = np.array([r[ax::3] for r in R[-m:]]).T
                    r1 = b - As @ x
                    G = Em.T @ Rm; cc = np.linalg.lstsq(0.5 * (G + G.T), Em.T @ r1, rcond=None)[0]
                    x = x + Em @ cc
                x, it = pcg(b, x)
                its.append(it)
            res[K].append(max(its))
            e = xs[s] - xs[s - 1]; E.append(e); R.append(A3 @ e)
    print("frame", frame)
    for K in Ks:
        print("  K=%3d  iterations per solve:" % K, res[K], " mean %.1f" % np.mean(res[K]))
