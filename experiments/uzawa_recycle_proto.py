"""How much of the inner right-hand sides C^T d_k of one ADMM iteration's Schur CG lies in the space of the PREVIOUS ADMM
iteration's inner solutions q2_j (A-orthogonal among themselves: d_i S d_j = 0)?  CPU oracle, cube on a floor.
ratio = energy (A^-1 norm) of what is left of C^T d_k after the Galerkin projection / energy of C^T d_k.
python experiments/uzawa_recycle_proto.py [n]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube100k_uzawa_floor"], n)
o = sc.make_oracle(mode=1, big=True)
store = {"prev": [], "cur": [], "log": []}
orig = o.solve_uzawa.__func__
def solve_uzawa(self, x, b, hits):
    Cm, c = self.make_matrix(hits, self._dhits)
    if self.y.shape[0] != Cm.shape[0]:
        self.y = np.zeros(Cm.shape[0])
    if Cm.nnz == 0:
        return self._lu.solve(b), 1
    Ct = Cm.T.tocsr()
    x = self._lu.solve(b - Ct @ self.y)
    r = Cm @ x - c; d = r.copy()
    prev = store["prev"]; cur = []
    ratios = []
    for it in range(self.uz_max_iters):
        rhs = Ct @ d
        q2 = self._lu.solve(rhs)
        e_full = rhs @ q2
        # Galerkin projection on the previous ADMM iteration's pairs (q2_j, rhs_j), mutually A-orthogonal
        left = e_full
        for (qj, gj) in prev + cur:
            cj = (qj @ rhs) / gj
            left -= cj * cj * gj
        ratios.append(np.sqrt(max(left, 0.0) / e_full))
        cur.append((q2, e_full))
        q3 = Cm @ q2; denom = d @ q3
        alpha = (d @ r) / denom
        x = x - alpha * q2; self.y = self.y + alpha * d; r = r - alpha * q3
        beta = (r @ q3) / denom; d = r - beta * d
    store["log"].append((Cm.shape[0], ratios))
    store["prev"] = cur[-20:]
    return x, self.uz_max_iters
import types
o.solve_uzawa = types.MethodType(solve_uzawa, o)
for f in range(3):
    o.step()
for i, (rows, ratios) in enumerate(store["log"][-25:]):
    print("solve", len(store["log"]) - 25 + i, "rows", rows, "left after projection:", " ".join("%.2f" % r for r in ratios[:20:2]))
