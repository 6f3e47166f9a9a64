#!/bin/bash
# round 3: where a UzawaCG ADMM iteration goes (kernel stats of cube100k_uzawa_floor, solve totals)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03t
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
#( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cube100k_uzawa_floor --steps 3 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/err.txt )
#f=$(find $O -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-150
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, numpy as np
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube100k_uzawa_floor"], None)
s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
s.upload()
for f in range(4):
    t0 = s.solve_totals()
    s.step_device(stats=True)
    t1 = s.solve_totals()
    r = s.runtime_data()
    print("frame", f, "solves", t1[0]-t0[0], "converged", t1[1]-t0[1], "inner", t1[2]-t0[2], "per solve", (t1[2]-t0[2])/max(1,t1[0]-t0[0]), "global ms", r.global_ms)
PY
