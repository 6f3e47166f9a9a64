#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big" --durations=8 > gpurun_out/r03/h_tests_big.txt 2>&1
tail -15 gpurun_out/r03/h_tests_big.txt
