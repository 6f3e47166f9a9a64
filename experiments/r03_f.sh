#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "onchip or short_pass or 48k or unstructured or uzawa" > gpurun_out/r03/f_tests.txt 2>&1
tail -3 gpurun_out/r03/f_tests.txt
STEPS=10 bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" "base=-DADMM_OC2_SMOOTH32=0 -DADMM_OC2_SPLITREC=0" "s32=-DADMM_OC2_SPLITREC=0" "split=-DADMM_OC2_SMOOTH32=0" "cur=" > gpurun_out/r03/f_ab.txt 2>&1
cat gpurun_out/r03/f_ab.txt
python experiments/oc_prof.py blob1m_mix 2>&1 | tail -4 > gpurun_out/r03/f_ocprof_blob.txt; cat gpurun_out/r03/f_ocprof_blob.txt
