#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05j; mkdir -p $O
for dbg in 0 1 2 3 7; do
  echo "== dbg $dbg"; ADMM_HIP_DEFL_DBG=$dbg ADMM_PROF_SOFT=24 ADMM_PROF_TOL=7e-10 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "end projection" | tail -2
done
