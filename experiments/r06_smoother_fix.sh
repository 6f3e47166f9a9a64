#!/bin/bash
# Round 6, fourth session: the block smoother of k_pcg2 survives the soft-mode computation (it was switched off for the context by the eigenvector
# solves of admm_hip_compute_soft_modes: the bench body ran with S = D^-1), and k_uz_persist no longer counts its products in k_pcg2's "trust revoked" word.
# oc_base = the library before the fix (experiments/_build/oc_base.so), in-tree = after.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06fix; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); g = d.get("roofline_global") or {}
    print("%-34s value %.1f  ms/frame %.3f  inner timed %.3f  solve_us %.1f  unconv %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["inner_iters_per_admm_iter"], g.get("solve_us", 0), d.get("unconverged_solves_in_timed_region")))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  for w in blob1m_mix cube1m_nh cube100k_uzawa_floor; do
    for v in base new; do
      L=""; [ $v = base ] && L=$R/experiments/_build/oc_base.so
      ADMM_HIP_LIB=$L timeout 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_${w}_$rep.json; summ $O/bench_${v}_${w}_$rep.json "[$v] $w"
    done
  done
done | tee $O/ab.txt
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="7e-10:SOFTSET=24;1e-9:SOFTSET=24;1.4e-9:SOFTSET=24;2e-9:SOFTSET=24;7e-10:SOFTSET=24:ADMM_HIP_OC_CHEB=3;1e-9:SOFTSET=24:ADMM_HIP_OC_CHEB=3" timeout 1500 python experiments/r05_drift.py 2>&1 | grep -v "^\[" | tee $O/drift.txt
timeout 1200 python -m pytest tests/test_soft_modes.py tests/test_gpu_parity.py -m gpu -q -x > $O/t1.txt 2>&1; tail -3 $O/t1.txt
