"""Sums a rocprofv3 --pmc counter (KiB units: FETCH_SIZE / WRITE_SIZE) per kernel family over a profiled bench step."""
import collections
import csv
import glob
import sys

root, counter = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        key = ("gemm_f32" if "gemm_f32" in k else "gemm_bf16" if "gemm_bf16" in k else "lstm_fwd" if "lstm_fwd" in k
               else "lstm_bwd" if "lstm_bwd" in k else "ctc" if "ctc_" in k else "other")
        a = agg[key]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%s %-10s sum = %.4g MB over %d launches, %.4g MB per launch" % (counter, k, v * 1024 / 1e6, n, v * 1024 / 1e6 / max(n, 1)))
