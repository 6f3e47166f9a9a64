import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, bench
for wl in ("blob1m_mix", "cube1m_mix"):
    sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl])
    s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600); s.upload()
    for _ in range(5): s.step_device(stats=True)
    for mode in (True, False, True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): s.step_device(stats=mode)
        s.download(); dt = time.perf_counter() - t0
        print(wl, "stats" if mode else "async", "%.1f ADMM it/s" % (20 * 20 / dt))
