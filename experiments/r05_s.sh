#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05s; mkdir -p $O
for m in 0 2; do echo "mask $m"; ADMM_HIP_DEFL_START=$m python experiments/iters_log.py blob1m_mix 30 2>&1 | grep frame | tail -3; done > $O/iters.txt
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24:ADMM_HIP_DEFL_START=2;7e-10:SOFTSET=24:ADMM_HIP_DEFL_START=6;1e-9:SOFTSET=24:ADMM_HIP_DEFL_START=2;7e-10:SOFTSET=24:ADMM_HIP_DEFL_START=0" timeout 1800 python experiments/r05_drift.py > $O/drift_start.txt 2>&1
cat $O/iters.txt; grep "^tol\|reference" $O/drift_start.txt | cut -c1-200
