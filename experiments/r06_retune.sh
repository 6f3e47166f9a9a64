#!/bin/bash
# blob1m_mix with the block smoother alive: are round 5's settings (7e-10, 24 modes, start step in front of the second solve) still the best that pass the 200-frame bar?
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06retune; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24;6e-10:SOFTSET=24;8e-10:SOFTSET=24;7e-10:SOFTSET=16;7e-10:SOFTSET=32;8e-10:SOFTSET=32;1e-9:SOFTSET=32;7e-10:SOFTSET=24:ADMM_HIP_DEFL_START=0;7e-10:SOFTSET=24:ADMM_HIP_DEFL_START=6;7e-10:SOFTSET=24:ADMM_HIP_DEFL_START=3" timeout 2400 python experiments/r05_drift.py 2>&1 | grep -v "^\[" | tee $O/drift.txt
