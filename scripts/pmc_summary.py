"""Summarise a rocprofv3 --pmc counter_collection.csv: per (kernel, grid) mean of every counter per dispatch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = (r["Kernel_Name"][:48], r["Grid_Size"])
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in cs.items():
        print(f"   {c:36s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
