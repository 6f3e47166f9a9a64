#!/bin/bash
# round 3, final code: is the block-local smoother still worth its product per iteration? (ADMM_HIP_OC_CHEB=0, same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=20 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_OC_CHEB=0" "X=1" > gpurun_out/r03/af_ab.txt 2>&1
cat gpurun_out/r03/af_ab.txt
