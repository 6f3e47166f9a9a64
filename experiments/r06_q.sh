#!/bin/bash
# Round 6, third session: row path of k_gs_persist (own data first, single obstacle in SGPRs, parked partial after the row) vs the
# commit before it; bit-identity and contact tests on the variant; the all-to-all floor of k_pcg2 by record layout.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=r06q bash experiments/r06_l.sh "gs_base gs_rows gs_basefine gs_rowsfine"
O=gpurun_out/r06q
ADMM_HIP_LIB=$PWD/experiments/_build/gs_rows.so timeout 900 python -m pytest tests/test_gs_persist.py tests/test_f3_terms.py -m gpu -q -x > $O/t_gs_rows.txt 2>&1; tail -3 $O/t_gs_rows.txt
ADMM_HIP_LIB=$PWD/experiments/_build/gs_rows.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gs or cloth or floor or obstacle or sphere or plane" > $O/t_gs_rows2.txt 2>&1; tail -3 $O/t_gs_rows2.txt
timeout 600 python experiments/probe_a2a_modes.py blob1m_mix 2>&1 | grep "^mode" > $O/probe_a2a.txt; cat $O/probe_a2a.txt
