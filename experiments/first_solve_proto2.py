"""(round 3 follow-up of first_solve_proto.py: LARGER recycled bases for the first solves of a frame -- joint Galerkin systems of 6, 8 and
up to 12 pairs, and a second sequential 4-pair stage.  Result: the gain needs >= 8 pairs in ONE system; not built)
The FIRST solve of a frame takes 85 of the 194 PCG iterations of a blob1m_mix frame (no recycled pairs yet, start = x_bar).
Does projecting on the PREVIOUS frame's pairs help it with the two-level preconditioner?  CPU oracle + scipy PCG (affine coarse
space, Jacobi smoother), per frame: iterations of solves 0..3 with (a) this frame's pairs only (what the kernel does), (b) also the
previous frame's first K pairs for the solves that have fewer than 4 of their own.  python experiments/first_solve_proto.py [n] [G]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'experiments')
import numpy as np, scipy.sparse as sp, scipy.sparse.csgraph as csg
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], n)
o = sc.make_oracle(mode=1, big=True)
Ah = o.A[0::3, :][:, 0::3].tocsr(); dinv = 1.0 / Ah.diagonal()
s = sc.make_solver(init=False)
plan = s.host_oc_plan(G, 4, settings=sc.product_settings)
rv, wt = plan["row_vertex"], plan["row_weights"]; live = rv >= 0
blk = np.arange(len(rv)) // 256
rr = np.repeat(np.nonzero(live)[0], 4)
P = sp.csr_matrix((wt[live].ravel().astype(float), (np.repeat(rv[live], 4), 4 * blk[rr] + np.tile(np.arange(4), live.sum()))), shape=(nv, 4 * G))
Ai = plan["coarse_inv"]
prec = lambda R: dinv[:, None] * R + P @ (Ai @ (P.T @ R))
def solve(B, X0, pairs, tol=1e-8, maxit=400, pairs2=None):
    X = X0.copy(); R = B - Ah @ X
    for stage in (pairs, pairs2):
      for ax in range(3):
        pairs = stage
        if pairs:
            E = np.stack([p[0][:, ax] for p in pairs], 1); AE = np.stack([p[1][:, ax] for p in pairs], 1)
            c = np.linalg.lstsq(E.T @ AE, E.T @ R[:, ax], rcond=None)[0]
            X[:, ax] += E @ c; R[:, ax] -= AE @ c
    b2 = (B * dinv[:, None] * B).sum(0)
    Z = prec(R); Pd = Z.copy(); rz = (R * Z).sum(0)
    for it in range(maxit):
        if ((R * dinv[:, None] * R).sum(0) <= tol * tol * b2).all(): return it
        AP = Ah @ Pd; al = rz / (Pd * AP).sum(0); X += al * Pd; R -= al * AP
        Z = prec(R); rz2 = (R * Z).sum(0); Pd = Z + (rz2 / rz) * Pd; rz = rz2
    return maxit
prev = []
for f in range(5):
    o.v[1::3] += o.dt * o.gravity
    x_bar = o.x + o.dt * o.v; Mxbar = o.m * x_bar
    curr = x_bar.copy(); z = np.zeros(o.R); u = np.zeros(o.R)
    pairs = []; its_a = []; its_b = []; its_c = []; its_d = []; its_e = []; its_f = []
    for si in range(o.admm_iters):
        o.local_step(curr, z, u); b = o.rhs(Mxbar, z, u); xs = o.solve_ldlt(b)
        if si < 8:
            B = b.reshape(-1, 3); X0 = curr.reshape(-1, 3)
            its_a.append(solve(B, X0, pairs[-4:]))
            K = int(sys.argv[4]) if len(sys.argv) > 4 else 4
            mode = sys.argv[3] if len(sys.argv) > 3 else "same"
            free = max(0, K - len(pairs))
            extra = (prev[:free] if mode == "first" else prev[si:si + free] if mode == "same" else prev[max(0, si - 1):max(0, si - 1) + free]) if prev else []
            its_b.append(solve(B, X0, pairs[-K:] + extra))
            own = pairs[-4:]; its_c.append(solve(B, X0, own + (prev[si:si + 6 - len(own)] if prev else [])))
            its_d.append(solve(B, X0, own + (prev[si:si + 8 - len(own)] if prev else [])))
            its_e.append(solve(B, X0, own + (prev[:8] if prev else [])))
            k4 = pairs[-K:] + extra; used = set(id(p) for p in k4)
            its_f.append(solve(B, X0, k4, pairs2=[p for p in (prev[max(0, si - 1):si + 5] if prev else []) if id(p) not in used][:4]))
        e = (xs - curr).reshape(-1, 3); pairs.append((e, Ah @ e)); curr = xs
    o.v = (curr - o.x) / o.dt; o.x = curr
    print("frame", f, "own pairs only:", its_a, "| kernel (same-index fill to 4):", its_b, "| joint, filled to 6 with prev[si..]:", its_c, "| joint, filled to 8:", its_d, "| own + prev 0..7 jointly:", its_e, "| kernel basis, then 4 more prev pairs as a second stage:", its_f, flush=True)
    prev = pairs[:16]
