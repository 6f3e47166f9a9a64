#!/bin/bash
# round 3: pair count decided per context from the fourth frame's iteration counts (default) vs always four, same box; then the tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=20 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_RC_ADAPT=0" "X=1" > gpurun_out/r03/ae_ab.txt 2>&1
cat gpurun_out/r03/ae_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py -x -q -m gpu 2>&1 | tail -2
