#!/bin/bash
# round 3, final code: how many recycled pairs? (ADMM_HIP_RC_PAIRS=2/3/4, same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=10 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_RC_PAIRS=2" "ADMM_HIP_RC_PAIRS=3" "ADMM_HIP_RC_PAIRS=4" > gpurun_out/r03/ad_ab.txt 2>&1
cat gpurun_out/r03/ad_ab.txt
