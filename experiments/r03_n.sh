#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_edge_cases.py -x -q -m gpu > gpurun_out/r03/n_tests.txt 2>&1
tail -25 gpurun_out/r03/n_tests.txt
