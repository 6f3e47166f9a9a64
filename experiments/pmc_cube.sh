#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE separately, --kernel-trace only) for one workload: bash experiments/pmc_cube.sh cube1m_mix
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-cube1m_mix}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$W
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $O/$C.err )
done
find $O -name "*counter_collection.csv"
