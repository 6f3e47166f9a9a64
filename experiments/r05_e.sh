#!/bin/bash
# Round 5, fifth GPU session: fused soft-mode end projection (tests, A/B of the bench line, 200-frame drift), launch-path PCG after the k_big_vec fix.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_soft_modes.py tests/test_big_pcg.py tests/test_multi_gpu.py -m gpu -q -s > $O/t.txt 2>&1
tail -30 $O/t.txt
for sm in 0 32 0 32; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --soft-modes $sm > $O/bench_soft$sm.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench_soft$sm.json')); print('soft %2d: %.0f ADMM it/s, %.2f ms/frame, %.2f its/solve, solve %.1f us, local frac %.3f' % (d['soft_modes'], d['value'], d['ms_per_frame'], d['inner_iters_per_admm_iter'], d['roofline_global']['solve_us'], d['roofline']['frac']))"
done | tee $O/ab_soft.txt
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="5e-10:SOFTLIB=32;5e-10:SOFTLIB=16;7e-10:SOFTLIB=32;1e-9:SOFTLIB=32;3e-10:SOFTLIB=32" timeout 1800 python experiments/r05_drift.py > $O/drift_fused.txt 2>&1
cat $O/drift_fused.txt | cut -c1-330
for n in 148 187; do
  timeout 1500 python bench.py --workload blob1m_mix --n $n --steps 6 --warmup 3 --no-cpu-baseline --soft-modes 0 > $O/size_$n.json 2> $O/size_$n.err
  python -c "
import json; d=json.load(open('$O/size_$n.json')); print('n=$n %d tets %d verts: %.0f ADMM it/s, %.2f ms/frame, %.1f its/solve, global %.3f ms' % (d['config']['elements'], d['config']['verts'], d['value'], d['ms_per_frame'], d['inner_iters_per_admm_iter'], d['split_ms_per_admm_iter']['global']))"
done | tee $O/size.txt
