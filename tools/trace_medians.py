#!/usr/bin/env python3
"""Per-kernel launch statistics (count, MEDIAN, mean, total) from a rocprofv3 --kernel-trace CSV (SURVEY 8d asks for the
median of >= 50 launches).   python tools/trace_medians.py <kernel_trace.csv> [top N]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
d = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print(f"# {len(rows)} kernel launches, total kernel time {tot / 1e3:.1f} ms")
print("# name | launches | median us | mean us | total ms | %")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:n]:
    v = sorted(v)
    print(f"{k[:120]} | {len(v)} | {v[len(v) // 2]:.1f} | {sum(v) / len(v):.1f} | {sum(v) / 1e3:.2f} | {sum(v) / tot * 100:.1f}")
