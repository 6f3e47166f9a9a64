#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05j; mkdir -p $O
ADMM_PROF_SOFT=24 ADMM_PROF_TOL=7e-10 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "end projection" | tail -2
timeout 600 python -m pytest tests/test_soft_modes.py -m gpu -q 2>&1 | tail -2
for cfg in "24 7e-10" "0 5e-10" "24 7e-10"; do
  set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --soft-modes $1 --pcg-tol $2 > $O/b.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('soft %2d tol $2: %.0f ADMM it/s, %.2f ms/frame, %.2f its/solve, solve %.1f us' % (d['soft_modes'], d['value'], d['ms_per_frame'], d['inner_iters_per_admm_iter'], d['roofline_global']['solve_us']))"
done
