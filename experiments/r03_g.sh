#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "onchip or short_pass or 48k or unstructured or uzawa" > gpurun_out/r03/g_tests.txt 2>&1
tail -3 gpurun_out/r03/g_tests.txt
STEPS=10 bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" "fillfirst=-DADMM_OC2_FILL_FIRST" "cur=" > gpurun_out/r03/g_ab.txt 2>&1
cat gpurun_out/r03/g_ab.txt
python experiments/oc_prof.py blob1m_mix 2>&1 | tail -4 > gpurun_out/r03/g_ocprof_blob.txt; cat gpurun_out/r03/g_ocprof_blob.txt
