#!/bin/bash
# Round 6, fourth session: rocprofv3 kernel statistics of the 2 M-tet body on the launch path -- where do the ~0.55 ms per solve outside the
# iterations go? (ADMM loop only: dispatches after the first local-step dispatch)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
O=$R/gpurun_out/r06z2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_148 -o p -- python $R/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_148.json 2> $O/stats_148.err )
cp $(find $O/stats_148 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_148.csv
python experiments/loop_stats_from_trace.py $O/stats_148 $O/kernel_stats_admm_loop_148.csv
python - $O/stats_148 <<'PY' > $O/frame_sequence_148.txt
# the kernel sequence of ONE ADMM iteration late in the run (names + durations + gaps), from the kernel trace
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
loc = [i for i, n in enumerate(names) if "k_local_tets" in n]
i0, i1 = loc[-3], loc[-2]
prev_end = int(rows[i0 - 1]["End_Timestamp"])
for r, n in zip(rows[i0:i1], names[i0:i1]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s %8.1f us   gap before %6.1f us" % (n[:40], (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("ADMM iteration: %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3))
PY
rm -rf $O/stats_148
cat $O/kernel_stats_admm_loop_148.csv | head -30
cat $O/frame_sequence_148.txt | head -120
