#!/bin/bash
# round 3: local vector of k_pcg2 as AoS (24-byte entries) vs per-axis arrays, same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=10 bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" "cur=" "aos=-DADMM_OC2_AOS=1" > gpurun_out/r03/x_ab.txt 2>&1
cat gpurun_out/r03/x_ab.txt
ADMM_HIP_LIB=/tmp/ab/aos.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "onchip or unstructured or big_blob_bench" 2>&1 | tail -2
