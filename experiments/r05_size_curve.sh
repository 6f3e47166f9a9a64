#!/bin/bash
# Round 5: ADMM it/s of ONE body by size (round-4 review item 3): 0.25 / 0.5 / 1.0 / 1.35 / 2 / 4 M tets.  Up to 262 144 vertices the on-chip
# solver, beyond it the launch-path two-level PCG; ADMM_HIP_BIG=0 (the Jacobi PCG of rounds 1-4) at 2 M for the A/B.
cd "$(dirname "$0")/.." || exit 1
O=${1:-gpurun_out/r05_size}; mkdir -p $O
for n in 74 93 118 130 148 187; do
  timeout 1500 python bench.py --workload blob1m_mix --n $n --steps 10 --warmup 5 --no-cpu-baseline > $O/size_$n.json 2> $O/size_$n.err
  python - $O/size_$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("n=%s  %8d tets %7d verts: %7.0f ADMM it/s, %6.2f ms/frame, %5.1f inner its per ADMM it, split local %.3f rhs %.3f global %.3f ms, unconverged %d" % (
        d["config"]["workload"].split("n=")[-1].rstrip(")"), d["config"]["elements"], d["config"]["verts"], d["value"], d["ms_per_frame"], d["inner_iters_per_admm_iter"],
        d["split_ms_per_admm_iter"]["local"], d["split_ms_per_admm_iter"]["rhs"], d["split_ms_per_admm_iter"]["global"], d["unconverged_solves_in_timed_region"]))
except Exception as e:
    print("failed:", sys.argv[1], e)
PY
done | tee $O/size_curve.txt
ADMM_HIP_BIG=0 timeout 1500 python bench.py --workload blob1m_mix --n 148 --steps 4 --warmup 2 --no-cpu-baseline --pcg-max-iters 3000 > $O/size_148_jacobi.json 2> $O/size_148_jacobi.err
python -c "
import json; d=json.load(open('$O/size_148_jacobi.json')); print('n=148 Jacobi launch path (ADMM_HIP_BIG=0): %.0f ADMM it/s, %.1f inner its per ADMM it' % (d['value'], d['inner_iters_per_admm_iter']))" | tee -a $O/size_curve.txt
