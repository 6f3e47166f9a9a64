import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv)>1 else 30
w = bench.WORKLOADS["cube1m_nh"]
sc, nt, nv = bench.build_scene(w, n)
s = sc.make_solver(init=False)
rp, ci, va = s.host_matrix(sc.product_settings)
A = sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(sc.m)
print("n", n, "nv", nv, "nnz/row", A.nnz/nv)
rng = np.random.default_rng(0)
xt = rng.standard_normal(nv); b = A @ xt
d = A.diagonal()

def pcg(apply_M, tol=1e-8, maxit=2000):
    x = np.zeros(nv); r = b.copy(); z = apply_M(r); p = z.copy(); rz = r@z; rz0 = b@apply_M(b)
    for it in range(maxit):
        Ap = A@p; al = rz/(p@Ap); x += al*p; r -= al*Ap; z = apply_M(r); rzn = r@z
        if rzn <= tol*tol*rz0: return it+1
        p = z + (rzn/rz)*p; rz = rzn
    return maxit

print("jacobi its", pcg(lambda r: r/d))

# Morton order aggregates
X = sc.x
def morton(X, bits=10):
    q = ((X - X.min(0))/(np.ptp(X,0).max()+1e-12)*(2**bits-1)).astype(np.uint64)
    code = np.zeros(len(X), np.uint64)
    for b_ in range(bits):
        for a in range(3):
            code |= ((q[:,a]>>np.uint64(b_))&np.uint64(1)) << np.uint64(3*b_+a)
    return code
order = np.argsort(morton(X), kind='stable')
def aggregates(order, size):
    agg = np.empty(nv, np.int64); agg[order] = np.arange(nv)//size
    return agg
for size in (64, 32, 16, 8):
    agg = aggregates(order, size); nc = agg.max()+1
    P = sp.csr_matrix((np.ones(nv), (np.arange(nv), agg)), shape=(nv, nc))
    Ac = (P.T @ A @ P).tocsc(); lu = spla.splu(Ac)
    add = lambda r: r/d + P @ lu.solve(P.T @ r)
    print("size", size, "nc", nc, "additive 2-level its", pcg(add))
    # scaled coarse correction (over-correction factor)
    for om in (1.5, 2.0):
        print("   omega", om, "its", pcg(lambda r: r/d + om*(P @ lu.solve(P.T @ r))))
    # multiplicative symmetric: jacobi(0.7) pre, coarse, jacobi post
    wj = 0.7
    def mult(r):
        z = wj*r/d
        z = z + P @ lu.solve(P.T @ (r - A@z))
        z = z + wj*(r - A@z)/d
        return z
    print("   multiplicative V(1,1) its", pcg(mult))
    # 3-level additive: level-2 aggregates of 64 coarse
    agg2 = np.arange(nc)//64; nc2 = agg2.max()+1
    P2 = sp.csr_matrix((np.ones(nc), (np.arange(nc), agg2)), shape=(nc, nc2))
    Ac2 = (P2.T @ Ac @ P2).tocsc(); lu2 = spla.splu(Ac2); dc = Ac.diagonal()
    def add3(r):
        rc = P.T @ r
        return r/d + P @ (rc/dc + P2 @ lu2.solve(P2.T @ rc))
    print("   3-level additive (jacobi on level1) its", pcg(add3))

print("---- smoothed aggregation ----")
def est_rho(Aop, dvec, its=30):
    v = rng.standard_normal(Aop.shape[0])
    for _ in range(its):
        v = (Aop @ v)/dvec; lam = np.linalg.norm(v); v /= lam
    return lam
rho = est_rho(A, d)
print("rho(D^-1 A) ~", rho)
def build_levels(A, order, sizes, smooth=True):
    levels = []
    Acur = A.tocsr(); ordcur = order
    for size in sizes:
        nvc = Acur.shape[0]
        agg = np.empty(nvc, np.int64); agg[ordcur] = np.arange(nvc)//size
        nc = agg.max()+1
        T = sp.csr_matrix((np.ones(nvc), (np.arange(nvc), agg)), shape=(nvc, nc))
        dc = Acur.diagonal()
        if smooth:
            r = est_rho(Acur, dc, 20)
            P = (T - (4.0/(3.0*r)) * sp.diags(1.0/dc) @ (Acur @ T)).tocsr()
        else:
            P = T
        Ac = (P.T @ Acur @ P).tocsr()
        levels.append((Acur, dc, P, r if smooth else None))
        Acur = Ac; ordcur = np.arange(nc)   # coarse numbering already follows the curve
    return levels, Acur
for sizes in ([8], [16], [27], [8,8], [8,8,8], [16,16]):
    levels, Ac = build_levels(A, order, sizes)
    luc = spla.splu(Ac.tocsc())
    print("sizes", sizes, "coarse n", Ac.shape[0], "coarse nnz/row %.1f" % (Ac.nnz/Ac.shape[0]), "level nnz/row", ["%.1f"%(L[0].nnz/L[0].shape[0]) for L in levels])
    def vcycle(r, lvl=0, nu=1, wj=None):
        if lvl == len(levels): return luc.solve(r)
        Al, dl, P, rh = levels[lvl]
        om = 4.0/(3.0*rh)
        z = om*r/dl
        for _ in range(nu-1): z = z + om*(r - Al@z)/dl
        z = z + P @ vcycle(P.T @ (r - Al@z), lvl+1, nu)
        for _ in range(nu): z = z + om*(r - Al@z)/dl
        return z
    print("    V(1,1) its", pcg(lambda r: vcycle(r, 0, 1)), "  V(2,2) its", pcg(lambda r: vcycle(r, 0, 2)))
    def additive(r, lvl=0):
        if lvl == len(levels): return luc.solve(r)
        Al, dl, P, rh = levels[lvl]
        return r/dl + P @ additive(P.T @ r, lvl+1)
    print("    additive its", pcg(additive))

print("---- plain (unsmoothed) aggregation, multilevel multiplicative ----")
for sizes in ([8], [8,8], [8,8,8], [8,8,8,8], [4,4,4,4,4], [8,4,4,4]):
    levels, Ac = build_levels(A, order, sizes, smooth=False)
    luc = spla.splu(Ac.tocsc())
    rhos = [est_rho(L[0], L[1], 20) for L in levels]
    for over in (1.0, 1.5, 1.8):
        def vcycle(r, lvl=0, nu=1):
            if lvl == len(levels): return luc.solve(r)
            Al, dl, P, _ = levels[lvl]
            om = 4.0/(3.0*rhos[lvl])
            z = om*r/dl
            for _ in range(nu-1): z = z + om*(r - Al@z)/dl
            z = z + over * (P @ vcycle(P.T @ (r - Al@z), lvl+1, nu))
            for _ in range(nu): z = z + om*(r - Al@z)/dl
            return z
        print("sizes", sizes, "over", over, "coarse n", Ac.shape[0], " V(1,1) its", pcg(lambda r: vcycle(r,0,1)), " V(2,2) its", pcg(lambda r: vcycle(r,0,2)))
