"""GPU idle time inside train steps from a rocprofv3 kernel trace: union of the kernels' busy intervals against the wall
time of the traced window, the largest gaps and which kernels surround them.
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 bench.py --workload c2 --steps 6 --warmup 3 ...
    python tools/gap_analysis.py <dir> [skip_fraction]"""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]                      # the timed steps (warm-up and set-up in front)
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
prev = rows[0]
for r in rows[1:]:
    if r[0] > cur_e:
        gaps.append((r[0] - cur_e, prev[2], r[2]))
        busy += cur_e - cur_s
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
    if r[1] >= cur_e:
        prev = r
busy += cur_e - cur_s
print("window %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6, len(gaps)))
import collections
by = collections.Counter()
for g, a, b in gaps:
    by[(a.split("(")[0][-40:], b.split("(")[0][-40:])] += g
print("idle by (kernel before -> kernel after), top 15:")
for (a, b), g in by.most_common(15):
    print("  %8.3f ms  %s -> %s" % (g / 1e6, a, b))
gs = sorted(g for g, _, _ in gaps)
print("gap sizes us: median %.1f, p90 %.1f, max %.1f" % (gs[len(gs) // 2] / 1e3, gs[int(len(gs) * 0.9)] / 1e3, gs[-1] / 1e3))
