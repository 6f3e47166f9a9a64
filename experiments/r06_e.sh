#!/bin/bash
# Round 6: the whole GPU suite on the current tree
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/suite.txt 2>&1; tail -30 $O/suite.txt
