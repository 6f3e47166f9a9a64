#!/bin/bash
# Round 6, third session: two-granule boundary rows (ADMM_GSP_PACK2) + fine phase split of k_gs_persist; bit-identity tests on the variant
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=r06m bash experiments/r06_l.sh "gs_cur gs_pack2 gs_fine gs_pack2fine"
O=gpurun_out/r06m
ADMM_HIP_LIB=$PWD/experiments/_build/gs_pack2.so timeout 900 python -m pytest tests/test_gs_persist.py -m gpu -q -x > $O/t_gs_pack2.txt 2>&1; tail -3 $O/t_gs_pack2.txt
