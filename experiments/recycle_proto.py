import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scenes, bench
w = bench.WORKLOADS["cube1m_mix"]
sc, nt, nv = bench.build_scene(w, int(sys.argv[1]) if len(sys.argv) > 1 else 14)
o = sc.make_oracle(mode=1, big=True)
A = o.A
for frame in range(4):
    tr = []
    o.step(trace=tr)
    if frame < 2: continue
    # reconstruct the sequence of solves of this frame
    xs = [t[3] for t in tr]; bs = [t[2] for t in tr]
    # x before first solve = x_bar; skip: use s>=1
    E = []; R = []
    out = []
    for s in range(1, len(tr)):
        xprev = xs[s-1]; b = bs[s]
        r0 = b - A @ xprev
        e_prev = xs[s-1] - (xs[s-2] if s >= 2 else xs[s-1])
        line = "s=%2d |r0|/|b| %.2e" % (s, np.linalg.norm(r0)/np.linalg.norm(b))
        for m in (1, 2, 4):
            if len(E) >= m:
                Em = np.array(E[-m:]).T; Rm = np.array(R[-m:]).T   # A e_j = r0_j
                G = Em.T @ Rm; c = np.linalg.lstsq(G, Em.T @ r0, rcond=None)[0]
                rnew = r0 - Rm @ c
                line += "  m=%d: x%.1f" % (m, np.linalg.norm(r0)/np.linalg.norm(rnew))
        out.append(line)
        E.append(xs[s] - xprev); R.append(r0)
    print("frame", frame); print("\n".join(out[::3]))
