"""MfmaUtil per kernel family over a profiled run: python tools/mfma_util.py <dir with *counter_collection.csv> [substr ...]
MfmaUtil = (sum SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (sum GRBM_GUI_ACTIVE / 8 XCDs), summed over the launches of a
kernel name (MI355X_MICROARCH.md: the SQ counter counts cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import csv, glob, sys
root, want = sys.argv[1], sys.argv[2:]
busy, act, n = {}, {}, {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("((")[0].split("(float")[0][:70]
        if want and not any(w in k for w in want):
            continue
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k] = busy.get(k, 0.0) + v
            n[k] = n.get(k, 0) + 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[k] = act.get(k, 0.0) + v
print("| kernel | launches | MFMA busy cycles per SIMD | active cycles per XCD | MfmaUtil |\n|---|---|---|---|---|")
for k in sorted(busy, key=lambda k_: -act.get(k_, 0.0)):
    if act.get(k):
        b, a = busy[k] / 1024.0, act[k] / 8.0
        print("| `%s` | %d | %.3g | %.3g | **%.1f %%** |" % (k, n[k], b, a, 100.0 * b / a))
