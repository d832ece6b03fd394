"""Average rocprofv3 --pmc counters per dispatch of kernels whose name contains argv[2]: python tools/pmc_summary.py counter_collection.csv substr"""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
