#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=10 bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" "trust0=-DADMM_OC2_TRUST=0" "cur=" "marks=-DADMM_OC2_MARKS" > gpurun_out/r03/c_ab.txt 2>&1
cat gpurun_out/r03/c_ab.txt
