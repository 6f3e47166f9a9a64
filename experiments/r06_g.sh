#!/bin/bash
# Round 6: history length of the recycled start (solves of a frame whose pairs are kept across frames) x pairs per projection, blob and cube.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(os.path.basename(sys.argv[1]), "no line", e); sys.exit(0)
print("   value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f stats %s  unconverged %s" % (
    d["value"], d["ms_per_step"], d["stats_frames_ms_per_step"], d["inner_iters_per_admm_iter"], d.get("inner_iters_per_admm_iter_statistics_frames"), d.get("unconverged_solves_in_timed_region")))
PY
}
i=0
while read -r WL ENVS; do
  i=$((i+1))
  env $ENVS timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_$i.json
  echo "$i: $WL $ENVS"; summ $O/bench_$i.json
done <<'LIST' | tee $O/sweep.txt
blob1m_mix ADMM_X=0
blob1m_mix ADMM_HIP_RC_HIST_N=8 ADMM_HIP_RC_PAIRS=4
blob1m_mix ADMM_HIP_RC_HIST_N=12 ADMM_HIP_RC_PAIRS=4
blob1m_mix ADMM_HIP_RC_HIST_N=20 ADMM_HIP_RC_PAIRS=4
blob1m_mix ADMM_HIP_RC_HIST_N=12 ADMM_HIP_RC_PAIRS=3
blob1m_mix ADMM_HIP_RC_HIST_N=20 ADMM_HIP_RC_PAIRS=3
blob1m_mix ADMM_HIP_RC_HIST_N=20
cube1m_nh ADMM_X=0
cube1m_nh ADMM_HIP_RC_HIST_N=8 ADMM_HIP_RC_PAIRS=4
cube1m_nh ADMM_HIP_RC_HIST_N=20 ADMM_HIP_RC_PAIRS=4
cube1m_nh ADMM_HIP_RC_HIST_N=20 ADMM_HIP_RC_PAIRS=3
cube1m_nh ADMM_HIP_RC_PAIRS=4
cube1m_mix ADMM_HIP_RC_HIST_N=20 ADMM_HIP_RC_PAIRS=4
LIST
