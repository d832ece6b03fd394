"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (top N rows)."""
import csv, sys
path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# total kernel time {tot/1e6:.1f} ms")
print("# name | calls | total ms | avg us | %")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f'{r["Name"][:110]} | {r["Calls"]} | {float(r["TotalDurationNs"])/1e6:.2f} | {float(r["AverageNs"])/1e3:.1f} | {float(r["TotalDurationNs"])/tot*100:.1f}')
