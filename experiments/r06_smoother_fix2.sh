#!/bin/bash
# the host-side fix alone (k_pcg2 untouched): same-box A/B against the library before it; smoother state of the contact workload
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06fix2; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); g = d.get("roofline_global") or {}
    print("%-34s value %.1f  ms/frame %.3f  inner timed %.3f  solve_us %.1f  unconv %s  column solves %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["inner_iters_per_admm_iter"], g.get("solve_us", 0), d.get("unconverged_solves_in_timed_region"), (d.get("uzawa") or {}).get("column_solves_in_timed_region")))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  for w in blob1m_mix cube1m_nh cube1m_mix; do
    for v in base new; do
      L=""; [ $v = base ] && L=$R/experiments/_build/oc_base.so
      ADMM_HIP_LIB=$L timeout 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_${w}_$rep.json; summ $O/bench_${v}_${w}_$rep.json "[$v] $w"
    done
  done
done | tee $O/ab.txt
for w in cube100k_uzawa_floor cube1m_nh; do
  echo "== $w"; ADMM_HIP_OC_DEBUG=1 timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -E "block smoother" | head -4
done | tee $O/state.txt
for v in base new; do
  L=""; [ $v = base ] && L=$R/experiments/_build/oc_base.so
  ADMM_HIP_LIB=$L timeout 600 python bench.py --workload cube100k_uzawa_floor --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_uz60.json; summ $O/bench_${v}_uz60.json "[$v] cube100k_uzawa_floor 60 frames"
done | tee -a $O/ab.txt
