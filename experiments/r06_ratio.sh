#!/bin/bash
# smoother interval [hi / ratio, hi] of k_pcg2's block Chebyshev polynomial (default ratio 16): larger ratios, bench + 200-frame drift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06ratio; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
STEPS=20 WARMUP=5 bash experiments/env_ab.sh "blob1m_mix cube1m_nh" "X=0" "ADMM_HIP_OC_CHEB_RATIO=400" "ADMM_HIP_OC_CHEB_RATIO=1000" "ADMM_HIP_OC_CHEB_RATIO=4000" 2>&1 | grep "^\[" | tee $O/ab.txt
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24;7e-10:SOFTSET=24:ADMM_HIP_OC_CHEB_RATIO=128;7e-10:SOFTSET=24:ADMM_HIP_OC_CHEB_RATIO=400;7e-10:SOFTSET=24:ADMM_HIP_OC_CHEB_RATIO=1000" timeout 1500 python experiments/r05_drift.py 2>&1 | grep -v "^\[" | tee $O/drift_blob.txt
ADMM_DRIFT_WORKLOAD=cube1m_nh ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="1e-7;1e-7:ADMM_HIP_OC_CHEB_RATIO=400;1e-7:ADMM_HIP_OC_CHEB_RATIO=1000" timeout 1500 python experiments/r05_drift.py 2>&1 | grep -v "^\[" | tee $O/drift_cube.txt
