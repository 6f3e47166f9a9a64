import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv)>1 else 36
w = bench.WORKLOADS["cube1m_nh"]
sc, nt, nv = bench.build_scene(w, n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)).tocsr()
rng = np.random.default_rng(0)
b = A @ rng.standard_normal(nv); d = A.diagonal()
def pcg(apply_M, tol=1e-8, maxit=3000):
    x = np.zeros(nv); r = b.copy(); z = apply_M(r); p = z.copy(); rz = r@z; rz0 = b@apply_M(b)
    for it in range(maxit):
        Ap = A@p; al = rz/(p@Ap); x += al*p; r -= al*Ap; z = apply_M(r); rzn = r@z
        if rzn <= tol*tol*rz0: return it+1
        p = z + (rzn/rz)*p; rz = rzn
    return maxit
print("n", n, "nv", nv, "jacobi", pcg(lambda r: r/d))
for size in (64, 256, 512, 1024, 2048):
    agg = np.arange(nv)//size; nc = agg.max()+1
    P = sp.csr_matrix((np.ones(nv), (np.arange(nv), agg)), shape=(nv, nc))
    Ac = (P.T @ A @ P).toarray(); Aci = np.linalg.inv(Ac)
    print(" index-block agg", size, "nc", nc, "additive its", pcg(lambda r: r/d + P @ (Aci @ (P.T @ r))))
# geometric boxes for comparison: 8x8x8 vertex boxes
X = sc.x; g = np.floor((X - X.min(0))/(np.ptp(X,0).max()+1e-9)*(n+1)/8).astype(int)
key = (g[:,0]*1000 + g[:,1])*1000 + g[:,2]; _, agg = np.unique(key, return_inverse=True); nc = agg.max()+1
P = sp.csr_matrix((np.ones(nv), (np.arange(nv), agg)), shape=(nv, nc)); Aci = np.linalg.inv((P.T@A@P).toarray())
print(" 8x8x8 boxes nc", nc, "additive its", pcg(lambda r: r/d + P @ (Aci @ (P.T @ r))))

# hierarchical ordering by recursive graph bisection (BFS level sets), then aggregates = consecutive ranges
import collections
Ag = sp.csr_matrix((np.ones_like(A.data), A.indices, A.indptr), shape=A.shape)
indptr, indices, dat = A.indptr, A.indices, A.data
def bfs_levels(nodes_mask, start, members):
    dist = {start: 0}; q = collections.deque([start]); last = start
    while q:
        v = q.popleft(); last = v
        for k in range(indptr[v], indptr[v+1]):
            w_ = indices[k]
            if dat[k] != 0.0 and nodes_mask[w_] and w_ not in dist:
                dist[w_] = dist[v] + 1; q.append(w_)
    return dist, last
def order_rec(members, mask, out, leaf=64):
    if len(members) <= leaf:
        out.extend(members); return
    d0, far = bfs_levels(mask, members[0], members)
    d1, far2 = bfs_levels(mask, far, members)
    # vertices not reached (disconnected) go last
    keyed = sorted(members, key=lambda v: d1.get(v, 1 << 30))
    half = len(keyed) // 2
    a, b = keyed[:half], keyed[half:]
    for part in (a, b):
        for v in members: mask[v] = False
        for v in part: mask[v] = True
        order_rec(part, mask, out, leaf)
    for v in members: mask[v] = True
import sys as _s; _s.setrecursionlimit(10000)
mask = np.ones(nv, bool); out = []
order_rec(list(range(nv)), mask, out)
order = np.array(out); rank = np.empty(nv, np.int64); rank[order] = np.arange(nv)
for size in (256, 512, 1024):
    agg = rank // size; nc = agg.max()+1
    P = sp.csr_matrix((np.ones(nv), (np.arange(nv), agg)), shape=(nv, nc)); Aci = np.linalg.inv((P.T@A@P).toarray())
    print(" graph-bisection order, range aggregates", size, "nc", nc, "additive its", pcg(lambda r: r/d + P @ (Aci @ (P.T @ r))))
