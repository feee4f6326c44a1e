"""Average rocprofv3 --pmc counters per kernel:  python tools/pmc_avg.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if flt in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
