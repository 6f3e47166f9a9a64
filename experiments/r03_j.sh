#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_known_answers.py tests/test_samples.py tests/test_cpp_api.py tests/test_edge_cases.py tests/test_dynamic_collision.py -x -q -m gpu -k "not big" > gpurun_out/r03/j_tests.txt 2>&1
tail -5 gpurun_out/r03/j_tests.txt
python experiments/dbg_cloth_affine.py 2>&1 | grep "cheb 1" 
python experiments/tol_blob.py > gpurun_out/r03/j_tol_blob_affine.txt 2>&1; cat gpurun_out/r03/j_tol_blob_affine.txt
STEPS=10 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_OC_AFFINE=0" "X=1" > gpurun_out/r03/j_ab.txt 2>&1
cat gpurun_out/r03/j_ab.txt
