#!/bin/bash
# Round 5, sixth GPU session: the fused soft-mode projection with batched loads and better-converged modes.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_soft_modes.py -m gpu -q -s > $O/t.txt 2>&1
tail -8 $O/t.txt
for cfg in "0 5e-10" "32 5e-10" "32 7e-10" "0 5e-10" "32 5e-10" "32 7e-10" "24 7e-10" "32 1e-9"; do
  set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --soft-modes $1 --pcg-tol $2 > $O/b.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('soft %2d tol $2: %.0f ADMM it/s, %.2f ms/frame, %.2f its/solve, solve %.1f us, local frac %.3f' % (d['soft_modes'], d['value'], d['ms_per_frame'], d['inner_iters_per_admm_iter'], d['roofline_global']['solve_us'], d['roofline']['frac']))"
done | tee $O/ab_soft.txt
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="5e-10:SOFTLIB=32;7e-10:SOFTLIB=32;1e-9:SOFTLIB=32;7e-10:SOFTLIB=24" timeout 1800 python experiments/r05_drift.py > $O/drift_fused.txt 2>&1
cat $O/drift_fused.txt | cut -c1-330
