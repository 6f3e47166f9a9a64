#!/bin/bash
# Is the block smoother of k_pcg2 alive in the bench context of blob1m_mix?  (ADMM_HIP_OC_CHEB knobs change the cube's iteration count and not the blob's.)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06sm; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for sm in 24 0; do
  echo "== blob1m_mix --soft-modes $sm"
  ADMM_HIP_OC_DIAG=1 ADMM_HIP_OC_DEBUG=1 timeout 600 python bench.py --workload blob1m_mix --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --soft-modes $sm 2> $O/err_$sm.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('it/s', round(d['value'],1), 'inner', d['inner_iters_per_admm_iter'])"
  grep -E "block smoother|lambda_max" $O/err_$sm.txt | head -5
  grep -c "^\[oc\] seq" $O/err_$sm.txt
done
for cfg in "X=0" "ADMM_HIP_OC_CHEB=0"; do
  for sm in 24 0; do
  echo "== [$cfg] soft modes $sm"; env $cfg timeout 600 python bench.py --workload blob1m_mix --steps 20 --warmup 5 --no-cpu-baseline --soft-modes $sm 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('it/s', round(d['value'],1), 'inner', d['inner_iters_per_admm_iter'])"
  done
done
