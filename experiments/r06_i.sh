#!/bin/bash
# Round 6: does the longer history of the recycled start hold the 200-frame drift bar?  (It did NOT at the blob's tolerance: 3.4e-5.)  Tolerance x history x pairs.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
V="7e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=5:ADMM_HIP_RC_ADAPT=1"
V="$V;7e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=5:ADMM_HIP_RC_PAIRS=4"
V="$V;7e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=8:ADMM_HIP_RC_PAIRS=4"
V="$V;7e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=12:ADMM_HIP_RC_PAIRS=4"
V="$V;7e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=20:ADMM_HIP_RC_PAIRS=3"
V="$V;5e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=20:ADMM_HIP_RC_PAIRS=4"
V="$V;3e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=20:ADMM_HIP_RC_PAIRS=4"
V="$V;2e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=20:ADMM_HIP_RC_PAIRS=4"
V="$V;5e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=8:ADMM_HIP_RC_PAIRS=4"
V="$V;5e-10:SOFTSET=32:ADMM_HIP_RC_HIST_N=20:ADMM_HIP_RC_PAIRS=4"
V="$V;7e-10:SOFTSET=24:ADMM_HIP_RC_HIST_N=20:ADMM_HIP_RC_PAIRS=4:ADMM_HIP_RC_ORDER=o1,o2,p0.1,p0.2"
ADMM_DRIFT_WORKLOAD=blob1m_mix ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="$V" timeout 2400 python experiments/r05_drift.py > $O/drift_blob_history.txt 2>&1
cut -c1-260 $O/drift_blob_history.txt
