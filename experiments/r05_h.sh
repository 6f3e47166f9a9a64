#!/bin/bash
# Round 5, eighth GPU session: launch-path PCG after the occupancy / SELL-padding changes (tests, profile at 2 M tets, size curve); oracle/_ref no longer travels.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
ls oracle/_ref 2>&1 | head -2 > $O/ref_dir.txt
timeout 1800 python -m pytest tests/test_big_pcg.py tests/test_multi_gpu.py tests/test_oracle_vs_ref.py tests/test_known_answers.py "tests/test_gpu_parity.py::test_onchip_pcg_preconditioner_modes" -m gpu -q -s > $O/t.txt 2>&1
tail -12 $O/t.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_big -- python $OLDPWD/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$O/prof_big.log 2>&1)
f=$(find $O/prof_big -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-140
bash experiments/r05_size_curve.sh $O/size > $O/size_log.txt 2>&1
cat $O/size/size_curve.txt
