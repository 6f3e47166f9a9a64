"""Round 5: does a Galerkin projection of the FINAL residual of every solve on the lowest eigenvectors of K (A = K (x) I3; exact pairs
(Z, K Z), so the step is exact whatever the accuracy of Z) let the PCG tolerance be looser at the same drift?  The drift of round 4
(profiles/r04_drift_tolerance_study.txt) is the error a residual-norm stop leaves in the soft modes; a start-up projection does not
hold it out (round 4) -- an END projection removes it exactly in span(Z).
CPU, the 52 k-tet twin of blob1m_mix, the oracle's ADMM loop with its exact solves replaced by a warm-started two-level PCG (the
library's own blocks, affine coarse space, two Chebyshev steps, the kernel's stop rule r.M^-1 r <= tol^2 b.M^-1 b per axis).
    python experiments/end_deflation_proto.py [n=44] [frames=25]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 25
sc, nt, nv = bench.build_scene(dict(bench.WORKLOADS["blob1m_mix"], linsolver=0), n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
K = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
G = 256 if nv > 100000 else max(4, nv // 700)
spb = -(-nv // (64 * G))
plan = s.host_oc_plan(G, spb, settings=sc.product_settings, coarse=False)
rv = plan["row_vertex"]; blk = np.zeros(nv, np.int64)
rows = np.nonzero(rv >= 0)[0]; blk[rv[rows]] = rows // (64 * spb)
print("tets", nt, "verts", nv, "blocks", G, flush=True)
d = K.diagonal(); dinv = 1.0 / d
coo = K.tocoo(); keep = blk[coo.row] == blk[coo.col]
Kbb = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=K.shape)
v = np.random.default_rng(1).standard_normal(nv)
for _ in range(60):
    w = dinv * (Kbb @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
lam *= 1.1; lo = lam / 16.0; th = 0.5 * (lam + lo); de = 0.5 * (lam - lo)
def S(R):      # two Chebyshev steps on the block-diagonal part, columns = axes
    y = dinv[:, None] * R; al = 1.0 / th; z = al * y; res = R - al * (Kbb @ y)
    be = 0.5 * (de * al) ** 2; al2 = 1.0 / (th - be / al); p = dinv[:, None] * res + be * y
    return z + al2 * p
X0 = sc.x
cols, vals, rws = [], [], []; nc = 0
for b in range(G):
    idx = np.nonzero(blk == b)[0]
    if len(idx) == 0: continue
    Y = X0[idx] - X0[idx].mean(axis=0); F = np.column_stack([np.ones(len(Y)), Y])
    for j in range(4):
        rws.append(idx); cols.append(np.full(len(idx), nc)); vals.append(F[:, j]); nc += 1
P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rws), np.concatenate(cols))), shape=(nv, nc))
Kc = (P.T @ K @ P).toarray(); Kci = np.linalg.inv(Kc)
prec = lambda R: S(R) + P @ (Kci @ (P.T @ R))
# lowest eigenvectors of K (shift-invert; small twin)
t0 = time.time(); lu = spla.splu(K.tocsc())
ew, Z = spla.eigsh(K, k=32, sigma=0.0, which='LM', OPinv=spla.LinearOperator((nv, nv), matvec=lu.solve))
o = np.argsort(ew); ew = ew[o]; Z = Z[:, o]
print("lowest eigenvalues of K:", " ".join("%.3g" % e for e in ew[:12]), " (largest of D^-1 K ~ 2)  %.0f s" % (time.time() - t0), flush=True)

class Pcg:
    def __init__(self, tol, kdefl=0, where="end", coarse_end=False, sched=None):
        self.tol0, self.tol, self.k, self.where, self.coarse_end = tol, tol, kdefl, where, coarse_end
        self.sched = sched          # multipliers of the tolerance for the first solves of a frame (ADMM damps what early solves leave)
        self.iters = 0; self.solves = 0
        if kdefl:
            self.Z = Z[:, :kdefl]; self.KZ = K @ self.Z; self.Gi = np.linalg.inv(self.Z.T @ self.KZ)
    def defl(self, X, B):
        R = B - K @ X
        return X + self.Z @ (self.Gi @ (self.Z.T @ R))
    def solve(self, x, b):
        si = self.solves % 20
        self.tol = self.tol0 * (self.sched[si] if self.sched is not None and si < len(self.sched) else 1.0)
        X = x.reshape(-1, 3).copy(); B = b.reshape(-1, 3)
        if self.k and self.where in ("start", "both"): X = self.defl(X, B)
        R = B - K @ X; U = prec(R); Pd = U.copy(); g = np.einsum('ij,ij->j', R, U)
        gb = np.einsum('ij,ij->j', B, prec(B)); gb = np.maximum(gb, 1e-30 * gb.max())
        act = g > self.tol ** 2 * gb
        it = 0
        while act.any() and it < 600:
            W = K @ Pd; a = np.where(act, g / np.einsum('ij,ij->j', Pd, W), 0.0)
            X += a * Pd; R -= a * W; U = prec(R); g2 = np.einsum('ij,ij->j', R, U)
            be = np.where(act, g2 / g, 0.0); Pd = U + be * Pd; g = g2; it += 1
            act = act & (g > self.tol ** 2 * gb)
        if self.k and self.where in ("end", "both"): X = self.defl(X, B)
        if self.coarse_end:
            R = B - K @ X; X = X + P @ (Kci @ (P.T @ R))
        self.iters += it; self.solves += 1
        return X.ravel(), 1

def run(tag, solver):
    o = sc.make_oracle(mode=1, big=True)
    ref = sc.make_oracle(mode=1, big=True)
    if solver is not None: o.global_solve = lambda x, b: solver.solve(x, b)
    errs = []; t0 = time.time()
    for f in range(frames):
        o.step(); ref_x = REF[f]
        errs.append(scenes.rel_err(o.x, ref_x))
    print("%-44s max rel_err %.2e (frame %d)  its/solve %5.2f  | %s  (%.0f s)" % (tag, max(errs), int(np.argmax(errs)), solver.iters / max(solver.solves, 1),
          " ".join("%.1e" % e for e in errs[::max(1, frames // 12)]), time.time() - t0), flush=True)

REF = []
oref = sc.make_oracle(mode=1, big=True)
for f in range(frames):
    oref.step(); REF.append(oref.x.copy())
if len(sys.argv) > 3 and sys.argv[3] == "sched":
    for tol in (1e-9, 5e-10, 2.5e-10):
        run("tol %.1e plain" % tol, Pcg(tol))
        for name, sch in (("x20 x20", [20, 20]), ("x20 x10 x5 x2", [20, 10, 5, 2]), ("x30 x30 x10 x10 x3 x3", [30, 30, 10, 10, 3, 3]),
                          ("geometric 0.75^(19-s) capped x40", [min(40.0, 0.75 ** -(19 - q)) for q in range(20)]),
                          ("x10 for solves 0-9", [10] * 10), ("x0.5 last 10, x20 first 2", [20, 20] + [1] * 8 + [0.5] * 10)):
            run("tol %.1e, first solves %s" % (tol, name), Pcg(tol, sched=sch))
    sys.exit(0)
for tol in (1e-8, 3e-9, 1e-9, 5e-10):
    run("tol %.0e plain" % tol, Pcg(tol))
    for k in (4, 8, 16, 32):
        run("tol %.0e + END projection on %d modes" % (tol, k), Pcg(tol, k, "end"))
    run("tol %.0e + start AND end on 8 modes" % tol, Pcg(tol, 8, "both"))
    run("tol %.0e + coarse-space end correction" % tol, Pcg(tol, coarse_end=True))
