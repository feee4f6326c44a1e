import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_sep_u" in r["Kernel_Name"] or "k_gemm_ws" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
