#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05l; mkdir -p $O
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24:ADMM_HIP_DEFL_DBG=8;7e-10:SOFTSET=24:ADMM_HIP_DEFL_DBG=0;8e-10:SOFTSET=24:ADMM_HIP_DEFL_DBG=8" timeout 1200 python experiments/r05_drift.py 2>&1 | grep "^tol\|reference" | cut -c1-175 | tee -a $O/ab2.txt
