#!/bin/bash
# Dynamic instruction counts of the local-step kernel (per wave) on a bench workload: bash experiments/local_insts.sh [workload]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=${1:-blob1m_mix}
O=$GRAFT_REPO_ROOT/gpurun_out/insts
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $O/err.txt )
python - <<PY
import csv, collections, glob
f = glob.glob("$O/**/p_counter_collection.csv", recursive=True)
if not f: print(open("$O/err.txt").read()[-2000:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "local" not in k and "gather" not in k: continue
    w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
    print(k, "waves %.0f" % w, " per wave:", {c: round(sum(x) / len(x) / w, 1) for c, x in v.items() if c != "SQ_WAVES"})
PY
