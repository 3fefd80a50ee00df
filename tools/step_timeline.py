"""One train step as the GPU saw it, from a rocprofv3 kernel trace: for the LAST full step of the trace (delimited by the
optimizer's update_kernel), every kernel with its start offset, duration and queue, and the time per kernel family that is
NOT overlapped by a persistent recurrence (what the step's wall time consists of besides the recurrences).
    python tools/step_timeline.py <trace dir> [--list]"""
import csv, glob, sys, collections
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "update_kernel" in r[2]]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
print("step wall %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
def fam(n):
    for k in ("lstm_fwd_persist", "lstm_bwd_persist", "lstm_fwd_pair", "lstm_bwd_pair", "gemm_f32g", "gemm_f32_kernel", "gemm_bf16g", "gemm_bf16s", "gemm_bf16_kernel",
              "splitk_reduce", "dropout_scale", "cast_bf16", "ctc_mm", "colsum", "unit_param", "persist_verify", "l2_sumsq", "transpose", "moe_", "greedy"):
        if k in n: return k
    return "other:" + n.split("(")[0][-30:]
rec = [(s, e) for s, e, n, q in step if "persist_kernel" in n or "pair_kernel" in n or "persist_bf16" in n]
def uncovered(s, e):
    cut = e - s
    for rs, re in rec:
        lo, hi = max(s, rs), min(e, re)
        if hi > lo: cut -= hi - lo
    return max(cut, 0)
tot, unc = collections.Counter(), collections.Counter()
for s, e, n, q in step:
    f = fam(n)
    tot[f] += e - s
    if "persist" not in f and "pair" not in f: unc[f] += uncovered(s, e)
print("%-28s %10s %14s" % ("family", "total ms", "not under a recurrence"))
for f, v in tot.most_common():
    print("%-28s %10.3f %14.3f" % (f, v / 1e6, unc[f] / 1e6))
if "--list" in sys.argv:
    for s, e, n, q in step:
        print("%9.3f +%8.3f ms q%s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n.split("(")[0][-60:]))
