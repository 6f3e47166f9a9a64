"""rocprofv3 kernel trace -> per-kernel statistics of the ADMM LOOP alone (the dispatches after the first local-step dispatch): what
`--stats` would print had the run not started with the soft-mode computation (96 long k_pcg2 solves at initialize).  CSV like kernel_stats.
python experiments/loop_stats_from_trace.py <rocprof output dir> <out.csv>"""
import collections, csv, glob, statistics, sys
root, out = sys.argv[1], sys.argv[2]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not files:
    open(out, "w").write("no kernel trace found\n"); sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
name = next(k for k in rows[0] if k.lower() in ("kernel_name", "name"))
t0 = next(k for k in rows[0] if k.lower().startswith("start"))
t1 = next(k for k in rows[0] if k.lower().startswith("end"))
rows.sort(key=lambda r: int(r[t0]))
first_local = next((i for i, r in enumerate(rows) if "k_local_" in r[name]), 0)
agg = collections.defaultdict(list)
for r in rows[first_local:]:
    agg[r[name].split("(")[0]].append((int(r[t1]) - int(r[t0])) / 1e3)
tot = sum(sum(v) for v in agg.values())
with open(out, "w") as f:
    f.write("# ADMM loop only: dispatches after the first local-step dispatch (%d of %d); durations in us\n" % (len(rows) - first_local, len(rows)))
    f.write("Name,Calls,TotalDuration_us,Average_us,Median_us,Min_us,Max_us,Percentage\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f,%.2f\n' % (k, len(v), sum(v), statistics.mean(v), statistics.median(v), min(v), max(v), 100.0 * sum(v) / tot))
