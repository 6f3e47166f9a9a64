#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
python experiments/tol_blob.py > gpurun_out/r03/i_tol_blob.txt 2>&1; cat gpurun_out/r03/i_tol_blob.txt
