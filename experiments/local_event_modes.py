"""The two ways bench.py can time the local-step kernel with HIP events: a pair recorded around the launch (kernel + two dispatch gaps) and a
pair attached to the kernel's own dispatch (hipExtLaunchKernelGGL: its begin / end time stamps); and what timing costs the frame.
python experiments/local_event_modes.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "blob1m_mix"
sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], None)
s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=600)
s.upload()
for _ in range(5):
    s.step_device(stats=True)
for mode in (0, 1, 2, 0, 1, 2):
    s.time_local_launches(mode)
    s.download(); t = time.perf_counter()
    for _ in range(10):
        s.step_device(stats=False)
    s.download(); dt = time.perf_counter() - t
    n, ms = s.local_launch_times() if mode else (0, 0.0)
    print("mode %d: %.3f ms per frame; %d pairs, %.2f us per local-step launch" % (mode, 1e2 * dt, n, 1e3 * ms / max(1, n)), flush=True)
