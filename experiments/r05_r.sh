#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05r; mkdir -p $O
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24:ADMM_HIP_TOL_SCHED=10,10;7e-10:SOFTSET=24:ADMM_HIP_TOL_SCHED=30,30;7e-10:SOFTSET=24:ADMM_HIP_TOL_SCHED=100,100;7e-10:SOFTSET=24:ADMM_HIP_TOL_SCHED=10,30;5e-10:SOFTSET=24:ADMM_HIP_TOL_SCHED=30,30,3" timeout 1800 python experiments/r05_drift.py > $O/drift_sched_soft.txt 2>&1
grep "^tol\|reference" $O/drift_sched_soft.txt | cut -c1-190
