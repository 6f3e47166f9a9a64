#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or 48k or unstructured or mixed or beams" > gpurun_out/r03/e_tests.txt 2>&1
tail -3 gpurun_out/r03/e_tests.txt
STEPS=10 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_FUSE_RHS=0" "X=1" > gpurun_out/r03/e_ab.txt 2>&1
cat gpurun_out/r03/e_ab.txt
python experiments/oc_prof.py blob1m_mix 2>&1 | tail -4 > gpurun_out/r03/e_ocprof_blob.txt; cat gpurun_out/r03/e_ocprof_blob.txt
