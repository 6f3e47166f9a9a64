"""What a bad VERTEX numbering costs: cube1m_mix with the generator's lexicographic numbering vs a random renumbering of
the vertices (same mesh, same physics)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else None
for order in ("lexicographic", "random", "random+rcm", "lexicographic+rcm"):
    sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_mix"], n)
    if order.startswith("random"):
        p = np.random.default_rng(0).permutation(nv)          # new id of old vertex i = p[i]
        inv = np.empty(nv, np.int64); inv[p] = np.arange(nv)
        sc.x = sc.x[inv]; sc.m = sc.m[inv]
        sc.tets = [(verts[inv], p[tets].astype(np.int32), lame, kind, off) for verts, tets, lame, kind, off in sc.tets]
        sc.pins = {int(p[k]): v for k, v in sc.pins.items()}
    if order.endswith("+rcm"):            # mesh preprocessing: admm_host_locality_order (reverse Cuthill-McKee)
        from admm_elastic_amd import capi
        alltets = np.concatenate([t[1] for t in sc.tets])
        new_id, before, after = capi.locality_order(nv, alltets)
        print("   mean edge span %.0f -> %.0f" % (before, after))
        inv = np.empty(nv, np.int64); inv[new_id] = np.arange(nv)
        sc.x = sc.x[inv]; sc.m = sc.m[inv]
        sc.tets = [(verts[inv], new_id[tets].astype(np.int32), lame, kind, off) for verts, tets, lame, kind, off in sc.tets]
        sc.pins = {int(new_id[k]): v for k, v in sc.pins.items()}
    s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
    s.upload()
    loc = rhs = glob = 0.0; inner = 0
    for f in range(4):
        s.step_device(stats=True)
        if f >= 2:
            rd = s.runtime_data(); loc += rd.local_ms; rhs += rd.rhs_ms; glob += rd.global_ms; inner += rd.inner_iters
    print("%-18s local %.1f us  rhs %.1f us  global %.3f ms per ADMM iteration, %.1f PCG its" % (order, 1e3 * loc / 40, 1e3 * rhs / 40, glob / 40, inner / 40), flush=True)
    s.close()
