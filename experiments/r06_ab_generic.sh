#!/bin/bash
# same-box A/B: experiments/_build/oc_base.so vs the in-tree library, bench lines of the PCG workloads + phase table + parity tests of the in-tree library
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06ab; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); g = d.get("roofline_global") or {}
    print("%-26s value %.1f  ms/frame %.3f  inner timed %.3f  solve_us %.1f  unconv %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["inner_iters_per_admm_iter"], g.get("solve_us", 0), d.get("unconverged_solves_in_timed_region")))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2 3; do
  for w in ${WLS:-blob1m_mix cube1m_nh}; do
    for v in base new; do
      L=""; [ $v = base ] && L=$R/experiments/_build/oc_base.so
      ADMM_HIP_LIB=$L timeout 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_${w}_$rep.json; summ $O/bench_${v}_${w}_$rep.json "[$v] $w"
    done
  done
done | tee $O/ab.txt
for v in base new; do
  L=""; [ $v = base ] && L=$R/experiments/_build/oc_base.so
  echo "[$v]"; ADMM_HIP_LIB=$L timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "oc_prof" | tail -4
done | tee $O/ocprof.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_soft_modes.py tests/test_known_answers.py -m gpu -q -x > $O/t.txt 2>&1; tail -2 $O/t.txt
