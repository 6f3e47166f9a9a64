"""Round 6: the all-to-all floor of k_pcg2 by record layout (ADMM_HIP_PROBE_A2A_MODE, pcg_onchip2.hpp: k_sync_probe modes 0, 2, 3, 4)."""
import os, sys, subprocess, json
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import bench
    sc, nt, nv = bench.build_scene(dict(bench.WORKLOADS[sys.argv[2]]), None)
    s = sc.make_solver()
    for rep in range(3):
        a2a, xch, pst = s.probe_sync(400)
        print("mode %s rep %d: all-to-all %.3f us, exchange %.3f us" % (os.environ.get("ADMM_HIP_PROBE_A2A_MODE", "0"), rep, a2a, xch), flush=True)
else:
    wl = sys.argv[1] if len(sys.argv) > 1 else "blob1m_mix"
    for rnd in range(2):
        for m in ("0", "2", "3", "4"):
            env = dict(os.environ, ADMM_HIP_PROBE_A2A_MODE=m)
            subprocess.run([sys.executable, __file__, "child", wl], env=env)
