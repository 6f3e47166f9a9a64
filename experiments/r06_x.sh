#!/bin/bash
# Round 6, third session: k_pcg2 with the TAGGED vector exchange (ADMM_OC2_TAGGED: two-granule rows, polled; no drain / flag) vs the flag hand-off:
# parity tests on the variant, same-box A/B, phase tables, the exchange floor.  + the GS tests on the in-tree library (192 rows per block).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06x; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gs_persist.py tests/test_f3_terms.py tests/test_edge_cases.py -m gpu -q -x > $O/t_gs.txt 2>&1; tail -2 $O/t_gs.txt
ADMM_HIP_LIB=$PWD/experiments/_build/oc_tag.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_soft_modes.py tests/test_known_answers.py -m gpu -q -x > $O/t_parity_tag.txt 2>&1; tail -3 $O/t_parity_tag.txt
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
    for v in oc_base oc_tag; do
      ADMM_HIP_LIB=$PWD/experiments/_build/$v.so timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_${WL}_$rep.json; summ $O/bench_${v}_${WL}_$rep.json "[$v] $WL"
    done
  done
done | tee $O/ab.txt
for v in oc_base oc_tag; do
  echo "[$v]" >> $O/ocprof.txt
  ADMM_HIP_LIB=$PWD/experiments/_build/$v.so timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "oc_prof" | tail -4 >> $O/ocprof.txt
done
cat $O/ocprof.txt
