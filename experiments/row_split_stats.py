"""k_pcg2: could the rows be split into a block-local part (processed while waiting for the neighbours) and a halo part?  Padding of
separate local / halo widths per slice, and the share of entries a wave-uniform split point would cover, from the plan of a workload.
python experiments/row_split_stats.py blob1m_mix   (CPU only; result: DESIGN 9)"""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scipy.sparse as sp
import bench
wl = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else None
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], n)
s = sc.make_solver(init=False)
ns = (nv + 63) // 64
G = min(256, ns); spb = (ns + G - 1) // G
while spb < 16 and (ns + spb - 1) // spb > 32 * spb: spb += 1
G = (ns + spb - 1) // spb
plan = s.host_oc_plan(G, spb, settings=sc.product_settings)
print(plan["stats"])
rv = plan["row_vertex"]; T = 64 * spb
rp, ci, va = None, None, None
import admm_elastic_amd as pkg
from admm_elastic_amd import capi
d = s.make_desc(sc.product_settings)
import ctypes as C
nnz = C.c_int32(0)
capi.check(capi.lib().admm_host_assemble_matrix(C.byref(d), None, None, None, C.byref(nnz)))
rp = np.zeros(nv + 1, np.int32); ci = np.zeros(nnz.value, np.int32); va = np.zeros(nnz.value)
capi.check(capi.lib().admm_host_assemble_matrix(C.byref(d), capi.iptr(rp), capi.iptr(ci), capi.dptr(va), C.byref(nnz)))
A = sp.csr_matrix((va, ci, rp), shape=(nv, nv))
pos = np.full(nv, -1); live = rv >= 0; pos[rv[live]] = np.nonzero(live)[0]
blk_of_v = pos // T
nloc = np.zeros(len(rv), int); nhal = np.zeros(len(rv), int)
for r in np.nonzero(live)[0]:
    v = rv[r]; cols = ci[rp[v]:rp[v+1]]; vals = va[rp[v]:rp[v+1]]
    m = (cols != v) & (vals != 0.0)
    b = blk_of_v[cols[m]] == r // T
    nloc[r] = b.sum(); nhal[r] = (~b).sum()
r4 = lambda x: (x + 3) // 4 * 4
cur = 0; split = 0; split_sorted = 0
for s0 in range(0, len(rv), 64):
    L = nloc[s0:s0+64]; H = nhal[s0:s0+64]
    cur += 64 * max(4, r4((L + H).max())); split += 64 * (max(4, r4(L.max())) + r4(H.max()))
# rows re-sorted inside each block by (halo count desc, then length desc)
for b0 in range(0, len(rv), T):
    idx = np.arange(b0, min(len(rv), b0 + T))
    order = idx[np.lexsort((-(nloc[idx] + nhal[idx]), -nhal[idx]))]
    for s0 in range(0, len(order), 64):
        q = order[s0:s0+64]
        split_sorted += 64 * (max(4, r4(nloc[q].max())) + r4(nhal[q].max()))
tot = (nloc + nhal).sum()
print("nnz offdiag", tot, "stored now", cur, "(%.3f)" % (cur / tot), "split in the current row order", split, "(%.3f)" % (split / tot), "split, rows sorted by (halo count, length)", split_sorted, "(%.3f)" % (split_sorted / tot))
print("rows with halo entries: %.1f %%" % (100.0 * (nhal > 0).sum() / live.sum()), "halo entries %.1f %%" % (100.0 * nhal.sum() / tot))
early = 0; early_sorted = 0; stored_sorted = 0
f4 = lambda x: x // 4 * 4
for s0 in range(0, len(rv), 64):
    L = nloc[s0:s0+64]; H = nhal[s0:s0+64]; lv = live[s0:s0+64]
    if lv.any(): early += 64 * f4(L[lv].min())
for b0 in range(0, len(rv), T):
    idx = np.arange(b0, min(len(rv), b0 + T)); idx = idx[live[idx]]
    # keep SELL efficiency: primary key length (desc), secondary local count
    order = idx[np.lexsort((-nloc[idx], -(nloc[idx] + nhal[idx])))]
    for s0 in range(0, len(order), 64):
        q = order[s0:s0+64]
        early_sorted += 64 * f4(nloc[q].min()); stored_sorted += 64 * max(4, r4((nloc[q] + nhal[q]).max()))
print("entries that can be processed before the halo arrives: current row order %.1f %% of stored; rows sorted by (length, local count): %.1f %% of %d stored" % (100.0 * early / cur, 100.0 * early_sorted / stored_sorted, stored_sorted))
