#!/bin/bash
# round 3, final code: interval ratio of the block-local Chebyshev smoother (ADMM_HIP_OC_CHEB_RATIO, default 16), same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=10 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_OC_CHEB_RATIO=6" "ADMM_HIP_OC_CHEB_RATIO=10" "ADMM_HIP_OC_CHEB_RATIO=16" "ADMM_HIP_OC_CHEB_RATIO=30" > gpurun_out/r03/ag_ab.txt 2>&1
cat gpurun_out/r03/ag_ab.txt
