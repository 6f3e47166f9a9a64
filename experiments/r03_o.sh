#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/o_tests_all.txt 2>&1
tail -25 gpurun_out/r03/o_tests_all.txt
