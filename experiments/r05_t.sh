#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05t; mkdir -p $O
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24:ADMM_HIP_DEFL_EVERY=2;7e-10:SOFTSET=24:ADMM_HIP_DEFL_EVERY=4;8e-10:SOFTSET=24;6e-10:SOFTSET=24:ADMM_HIP_DEFL_EVERY=2" timeout 1500 python experiments/r05_drift.py > $O/drift_every_start.txt 2>&1
grep "^tol\|reference" $O/drift_every_start.txt | cut -c1-200
