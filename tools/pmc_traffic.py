"""Turns two rocprofv3 PMC passes over the same command (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, counters only
with --kernel-trace: /opt/skills/guides/MI355X_MICROARCH.md, HBM section) into per-kernel-family HBM-side bytes per launch:

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <workload> [out.json]

FETCH_SIZE is doubled (gfx950 reports one half of the bytes of wide coalesced reads; every kernel here reads 16 bytes per
lane or gathers from rows that are read in full), WRITE_SIZE is taken 1:1 (calibrated in round 1 on the GEMM's C store);
counter unit KiB; Infinity-Cache hits are included (L2-miss traffic: an upper bound on DRAM traffic).  Merges the result
into the JSON that bench.py reads for `roofline.traffic` / `roofline_ctc.traffic` and prints a markdown table."""
import collections
import csv
import glob
import json
import os
import sys


def family(k):
    if "ctc_mm_kernel" in k:
        return "ctc_phase1" if ", 1>" in k else "ctc_phase2"
    for name in ("gemm_x3_tn", "gemm_x3", "split_x3", "gemm_f32", "gemm_bf16g", "gemm_bf16s", "gemm_bf16", "lstm_fwd_pair_x3",
                 "lstm_bwd_pair_x3", "lstm_fwd_persist_x3", "lstm_bwd_persist_x3", "lstm_fwd_pair", "lstm_bwd_pair", "lstm_fwd_persist", "lstm_bwd_persist",
                 "lstm_fwd_step", "lstm_bwd_step", "cast_bf16", "ctc_"):
        if name in k:
            return name
    return "other"


def sums(root, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[family(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]) * 1024.0
            a[1] += 1
    return agg


def main():
    fdir, wdir, workload = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "profiles", "r4_pmc_traffic.json")
    f, w = sums(fdir, "FETCH_SIZE"), sums(wdir, "WRITE_SIZE")
    rows = {}
    print("| kernel family | launches | FETCH_SIZE raw / launch | read / launch (x2) | WRITE_SIZE / launch | total / launch |")
    print("|---|---|---|---|---|---|")
    for k in sorted(set(f) | set(w)):
        n = max(f[k][1], w[k][1], 1)
        rd, wr = 2.0 * f[k][0] / n, w[k][0] / n
        rows[k] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "bytes": rd + wr}
        print("| `%s` | %d | %.1f MB | %.1f MB | %.1f MB | %.1f MB |" % (k, n, f[k][0] / n / 1e6, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
    entry = {"families": rows}
    gemms = [k for k in ("gemm_f32", "gemm_bf16g", "gemm_bf16s", "gemm_x3", "gemm_x3_tn") if k in rows]
    if gemms:                                               # the family that moves the most bytes in this workload
        entry["gemm_family"] = max(gemms, key=lambda k: rows[k]["bytes"] * rows[k]["launches"])
        entry["gemm"] = rows[entry["gemm_family"]]["bytes"]
    if "ctc_phase1" in rows and "ctc_phase2" in rows:        # one CTC call = one launch of each phase
        entry["ctc"] = rows["ctc_phase1"]["bytes"] + rows["ctc_phase2"]["bytes"]
    try:
        data = json.load(open(out))
    except (OSError, ValueError):
        data = {}
    data[workload] = entry
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print("\nwrote", out, "(%s: gemm %.4g B / launch, ctc %s B / call)" % (workload, entry.get("gemm", float("nan")), entry.get("ctc")))


if __name__ == "__main__":
    main()
