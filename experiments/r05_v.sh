#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05v; mkdir -p $O
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24:ADMM_HIP_OC_VERIFY=1;2e-9:SOFTSET=24:ADMM_HIP_OC_VERIFY=1;5e-9:SOFTSET=24:ADMM_HIP_OC_VERIFY=1;1e-8:SOFTSET=24:ADMM_HIP_OC_VERIFY=1;2e-9:SOFTSET=24" timeout 1500 python experiments/r05_drift.py > $O/drift_verify.txt 2>&1
grep "^tol\|reference" $O/drift_verify.txt | cut -c1-200
