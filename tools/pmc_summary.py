"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-34s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
