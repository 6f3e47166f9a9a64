#!/bin/bash
# Round 6, third session: k_pcg2 records as one 64-byte chunk per block, read chunk-wise (ADMM_OC2_REC_CHUNK) vs the round-5 layout:
# parity tests on the new default, same-box A/B (prebuilt variants), phase tables, the all-to-all floor.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06s; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_soft_modes.py tests/test_known_answers.py -m gpu -q -x > $O/t_parity.txt 2>&1; tail -3 $O/t_parity.txt
summ() { python - "$1" "$2" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "no line", e); sys.exit(0)
g = d.get("roofline_global") or {}
print("%-28s value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f  solve_us %.1f  a2a floor %.2f  xch floor %.2f  unconv %s" % (
    sys.argv[2], d["value"], d["ms_per_step"], d.get("stats_frames_ms_per_step", 0), d["inner_iters_per_admm_iter"], g.get("solve_us", 0),
    g.get("floor_all_to_all_us", 0), g.get("floor_exchange_us", 0), d.get("unconverged_solves_in_timed_region")))
PY
}
for rep in 1 2 3; do
  for WL in blob1m_mix cube1m_nh; do
    for v in oc_old oc_chunk; do
      ADMM_HIP_LIB=$PWD/experiments/_build/$v.so timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_${WL}_$rep.json; summ $O/bench_${v}_${WL}_$rep.json "[$v] $WL"
    done
  done
done | tee $O/ab.txt
for v in oc_old oc_chunk; do
  echo "[$v]" >> $O/ocprof.txt
  ADMM_HIP_LIB=$PWD/experiments/_build/$v.so timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "oc_prof" | tail -8 >> $O/ocprof.txt
done
cat $O/ocprof.txt
