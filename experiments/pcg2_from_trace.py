"""rocprofv3 kernel trace -> the k_pcg2 launches of the ADMM loop alone.

`compute_soft_modes` (at initialize) solves with the same kernel -- 96 launches of up to 2 000 iterations -- so the `--stats` average of
k_pcg2 mixes two populations.  The ADMM loop's solves are the k_pcg2 dispatches that follow the first local-step dispatch."""
import csv
import glob
import statistics
import sys

root, out = sys.argv[1], sys.argv[2]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not files:
    open(out, "w").write("no kernel trace found\n"); sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
name = next(k for k in rows[0] if k.lower() in ("kernel_name", "name"))
t0 = next(k for k in rows[0] if k.lower().startswith("start"))
t1 = next(k for k in rows[0] if k.lower().startswith("end"))
rows.sort(key=lambda r: int(r[t0]))
first_local = next((i for i, r in enumerate(rows) if "k_local_" in r[name]), len(rows))
before = [(int(r[t1]) - int(r[t0])) / 1e3 for r in rows[:first_local] if "k_pcg2" in r[name]]
loop = [(int(r[t1]) - int(r[t0])) / 1e3 for r in rows[first_local:] if "k_pcg2" in r[name]]
local = [(int(r[t1]) - int(r[t0])) / 1e3 for r in rows if "k_local_tets" in r[name]]
with open(out, "w") as f:
    f.write("k_pcg2 launches before the first local step (soft-mode computation at initialize): %d, mean %.1f us\n" % (len(before), statistics.mean(before) if before else 0.0))
    if loop:
        f.write("k_pcg2 launches of the ADMM loop: %d, mean %.1f us, median %.1f us, min %.1f, max %.1f\n" % (len(loop), statistics.mean(loop), statistics.median(loop), min(loop), max(loop)))
    if local:
        f.write("k_local_tets* launches: %d, mean %.2f us, median %.2f us\n" % (len(local), statistics.mean(local), statistics.median(local)))
