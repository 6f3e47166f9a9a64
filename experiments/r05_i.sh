#!/bin/bash
# Round 5, ninth GPU session: the bench settings (pcg_tol 7e-10 + 24 soft modes, single-precision modes in the fused epilogue): parity tests, A/B.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_soft_modes.py tests/test_bench_parity.py tests/test_big_pcg.py -m gpu -q -s > $O/t.txt 2>&1
grep -i "rel_err\|passed\|failed\|error" $O/t.txt | cut -c1-400 | tail -20
cp gpurun_out/drift_blob1m_frames.txt $O/ 2>/dev/null
for cfg in "24 7e-10" "0 5e-10" "24 7e-10" "0 5e-10" "16 7e-10" "0 2e-10"; do
  set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --soft-modes $1 --pcg-tol $2 > $O/b.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('soft %2d tol $2: %.0f ADMM it/s, %.2f ms/frame, %.2f its/solve, solve %.1f us, local frac %.3f' % (d['soft_modes'], d['value'], d['ms_per_frame'], d['inner_iters_per_admm_iter'], d['roofline_global']['solve_us'], d['roofline']['frac']))"
done | tee $O/ab.txt
