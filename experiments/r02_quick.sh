#!/bin/bash
# one box: GPU test suite, then same-box A/B of the libraries in ab/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" base new 2>&1 | grep "^\["
