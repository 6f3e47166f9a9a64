#!/bin/bash
# Round 6, first GPU session: (1) the slide-normal fix; (2) same-box A/B of the timed-vs-statistics gap of the default bench line
# (instrumentation of the local-step launches 2 / 1 / 0; warm-up 5 vs 25 frames); (3) 200-frame drift of cube1m_nh / cube1m_mix at the
# bench settings and neighbours; (4) the on-chip phase table BEFORE this round's kernel work.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_f3_terms.py -m gpu -q -x > $O/t_f3.txt 2>&1; tail -5 $O/t_f3.txt
for EV in 2 0 1 2 0; do
  ADMM_BENCH_LOCAL_EVENTS=$EV timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/err_ev$EV.txt | tail -1 > $O/bench_w5_ev${EV}_$RANDOM.json
done
for EV in 2 0; do
  ADMM_BENCH_LOCAL_EVENTS=$EV timeout 300 python bench.py --steps 20 --warmup 25 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_w25_ev${EV}.json
done
ADMM_BENCH_LOCAL_EVENTS=2 timeout 300 python bench.py --steps 20 --warmup 45 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_w45_ev2.json
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys, os
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("roofline_global", {})
print("%-28s value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f stats %s  solve_us %.1f (stats %s)  local %.1f us" % (
    os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], d["stats_frames_ms_per_step"], d["inner_iters_per_admm_iter"],
    d.get("inner_iters_per_admm_iter_statistics_frames"), g.get("solve_us", 0), g.get("statistics_frames"), d["roofline"]["avg_launch_us"]))
PY
done | tee $O/ab_summary.txt
for WL in cube1m_nh cube1m_mix; do
  ADMM_DRIFT_WORKLOAD=$WL ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24;7e-10;4e-10:SOFTSET=24;2e-10:SOFTSET=24" timeout 1500 python experiments/r05_drift.py > $O/drift_$WL.txt 2>&1
  cat $O/drift_$WL.txt
done
timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -24 > $O/ocprof_blob_before.txt
tail -8 $O/ocprof_blob_before.txt
