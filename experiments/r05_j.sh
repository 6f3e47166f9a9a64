#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05j; mkdir -p $O
for sm in 24; do
  ADMM_PROF_SOFT=$sm ADMM_PROF_TOL=7e-10 python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -9 > $O/ocprof_soft$sm.txt
  echo "== soft $sm"; cat $O/ocprof_soft$sm.txt
done
