#!/bin/bash
# round 3: kernel stats of the UzawaCG workload with the cached columns
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03w
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cube100k_uzawa_floor --steps 5 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.txt )
f=$(find $O -name "*kernel_stats.csv" | head -1); head -22 $f | cut -c1-160
cp $f $O/kernel_stats_cube100k_uzawa_floor.csv
rm -rf $O/stats
