#!/bin/bash
# Round 5, seventh GPU session: whole suite at the new bench tolerance, profile of the launch-path PCG at 2 M tets.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_big -- python $OLDPWD/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$O/prof_big.log 2>&1)
f=$(find $O/prof_big -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
timeout 3400 python -m pytest tests -m gpu -q > $O/suite.txt 2>&1
tail -25 $O/suite.txt
cp gpurun_out/drift_blob1m_frames.txt $O/ 2>/dev/null
