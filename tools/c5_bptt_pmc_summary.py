"""Per-dispatch counters of tools/c5_bptt_pmc.py's passes: mean over the 4 timed launches without / with the shadow."""
import csv, glob, sys
root, kern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "lstm_bwd_persist_bf16")
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, v in sorted(by.items()):
        v.sort()
        vals = [x for _, x in v]
        if len(vals) < 10:
            print("%-40s %d dispatches only: %s" % (name, len(vals), vals)); continue
        a, b = vals[1:5], vals[6:10]
        ma, mb = sum(a) / 4, sum(b) / 4
        print("%-40s no shadow %.4g | shadow %.4g | ratio %.3f | diff %.4g" % (name, ma, mb, mb / ma if ma else float("nan"), mb - ma))
for f in sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[:1]:
    d = [(int(r["Dispatch_Id"]) if "Dispatch_Id" in r else i, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
         for i, r in enumerate(csv.DictReader(open(f))) if kern in r["Kernel_Name"]]
    d.sort()
    t = [x for _, x in d]
    if len(t) >= 10:
        print("duration us: no shadow %.1f | shadow %.1f" % (sum(t[1:5]) / 4e3, sum(t[6:10]) / 4e3))
