#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03k
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for cfg in "ADMM_HIP_GS_THREE_KERNELS=1" "X=1"; do
( cd /tmp && env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$cfg -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cloth200k_gs_floor --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/err.txt )
echo "== $cfg"; head -8 $(find $O/s_$cfg -name "*kernel_stats.csv" | head -1) | cut -c1-60,100-190
done
rm -rf $O/s_*
