"""What does the stop rule of the global solve buy in POSITION error?  ADMM frames of the unstructured body on the CPU oracle (exact
solves); every solve is repeated with a two-level PCG (aggregates + Jacobi; recycled Galerkin start on the last 4 corrections) from
the previous iterate, recording per iteration: the Jacobi-norm relative residual (the kernel's rule), the preconditioned residual
gamma = r.M^-1 r relative to b.D^-1 b, and the true position error max|x - x*| / bbox.  python experiments/stop_rule_study.py [n] [G]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'experiments')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.sparse.csgraph as csg
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc, nt, nv = bench.build_scene(bench.WORKLOADS[sys.argv[4] if len(sys.argv) > 4 else "blob1m_mix"], n)
o = sc.make_oracle(mode=1, big=True)
Ah = o.A[0::3, :][:, 0::3].tocsr()
d = Ah.diagonal(); dinv = 1.0 / d
bbox = np.linalg.norm(sc.x.max(axis=0) - sc.x.min(axis=0))
print('tets', nt, 'verts', nv, 'bbox', bbox, flush=True)

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
Anz = Ah.copy(); Anz.data[:] = 1.0
fine = bisect(Anz, 4 * G)
if len(sys.argv) > 3 and sys.argv[3] == "affine":      # coarse space = {1, x, y, z} on each of the G blocks instead of 4 constants
    blk = fine // 4
    X3 = sc.x
    rowsP, colsP, valsP = [], [], []
    for b in range(G):
        mem = np.nonzero(blk == b)[0]
        c = X3[mem].mean(axis=0); h = np.abs(X3[mem] - c).max() + 1e-30
        for k in range(4):
            rowsP.append(mem); colsP.append(np.full(len(mem), 4 * b + k)); valsP.append(np.ones(len(mem)) if k == 0 else (X3[mem, k - 1] - c[k - 1]) / h)
    Pc = sp.csr_matrix((np.concatenate(valsP), (np.concatenate(rowsP), np.concatenate(colsP))), shape=(nv, 4 * G))
    print("affine coarse space:", Pc.shape)
else:
    Pc = sp.csr_matrix((np.ones(nv), (np.arange(nv), fine)), shape=(nv, fine.max() + 1))
Aci = np.linalg.inv((Pc.T @ Ah @ Pc).toarray())
prec = lambda R: dinv[:, None] * R + Pc @ (Aci @ (Pc.T @ R))

def pcg_trace(B, X0, Xs, pairs, maxit=60):
    """3 right-hand sides at once (columns = axes), own alpha/beta per axis"""
    X = X0.copy(); R = B - Ah @ X
    if pairs:   # Galerkin projection on the stored (E, A E) pairs, per axis
        for ax in range(3):
            E = np.stack([p[0][:, ax] for p in pairs], 1); AE = np.stack([p[1][:, ax] for p in pairs], 1)
            c = np.linalg.lstsq(E.T @ AE, E.T @ R[:, ax], rcond=None)[0]
            X[:, ax] += E @ c; R[:, ax] -= AE @ c
    b2 = (B * dinv[:, None] * B).sum(0)
    Z = prec(R); P = Z.copy(); rz = (R * Z).sum(0)
    tr = []
    for it in range(maxit):
        relD = np.sqrt(((R * dinv[:, None] * R).sum(0) / b2).max()); relM = np.sqrt((rz / b2).max())
        err = np.abs(X - Xs).max() / bbox
        Xc = X + Pc @ (Aci @ (Pc.T @ R))       # ... and after a final Galerkin coarse correction
        errc = np.abs(Xc - Xs).max() / bbox
        tr.append((it, relD, relM, err, errc))
        AP = Ah @ P; al = rz / (P * AP).sum(0); X += al * P; R -= al * AP
        Z = prec(R); rz2 = (R * Z).sum(0); P = Z + (rz2 / rz) * P; rz = rz2
    return tr, X

# ADMM frames on the oracle with a hook on every solve
frames = 2
rows = []
for f in range(frames):
    xbar = None
    # replicate OracleSolver.step with access to b (oracle.py: step)
    o.v[1::3] += o.dt * o.gravity
    x_bar = o.x + o.dt * o.v
    Mxbar = o.m * x_bar
    curr = x_bar.copy(); z = np.zeros(o.R); u = np.zeros(o.R)
    pairs = []
    for s in range(o.admm_iters):
        o.local_step(curr, z, u)
        b = o.rhs(Mxbar, z, u)
        xs = o.solve_ldlt(b)
        tr, _ = pcg_trace(b.reshape(-1, 3), curr.reshape(-1, 3), xs.reshape(-1, 3), pairs[-4:])
        e = (xs - curr).reshape(-1, 3); pairs.append((e, Ah @ e))
        rows.append((f, s, tr))
        curr = xs
    o.v = (curr - o.x) / o.dt; o.x = curr
for thr_name, col in (("Jacobi-norm rule", 1), ("M^-1-norm rule", 2)):
    for thr in (1e-8, 3e-9, 1e-9):
        its = []; errs = []; errc = []
        for f, s, tr in rows:
            k = next((t for t in tr if t[col] <= thr), tr[-1])
            its.append(k[0]); errs.append(k[3]); errc.append(k[4])
        print("%s  threshold %.0e: iterations/solve %.2f, position error at stop: max %.2e median %.2e | after a final coarse correction: max %.2e median %.2e" % (thr_name, thr, np.mean(its), max(errs), np.median(errs), max(errc), np.median(errc)))
# the error an iteration count buys, solve by solve (frame 1)
for f, s, tr in rows[20:26]:
    print("frame %d solve %d:" % (f, s), " ".join("%d:%.1e/%.1e/%.1e/%.1e" % t for t in tr[:18:2]))
