#!/bin/bash
# round 3: Binv from rest positions, productised (runtime switch): tests + same-box A/B against ADMM_HIP_TET_REST=0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_multi_gpu.py tests/test_samples.py -x -q -m gpu -k "not big_rot and not big_trans" > gpurun_out/r03/s_tests.txt 2>&1
tail -5 gpurun_out/r03/s_tests.txt
STEPS=10 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_TET_REST=0" "X=1" > gpurun_out/r03/s_ab.txt 2>&1
cat gpurun_out/r03/s_ab.txt
