#!/bin/bash
# round 3, call b: verification skip + FP32 carried smoother vectors: tests of the PCG paths, then same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "onchip or short_pass or big_blob or 48k or unstructured or uzawa" > gpurun_out/r03/b_tests.txt 2>&1
tail -15 gpurun_out/r03/b_tests.txt
STEPS=10 bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" "trust0=-DADMM_OC2_TRUST=0" "cur=" > gpurun_out/r03/b_ab.txt 2>&1
cat gpurun_out/r03/b_ab.txt
