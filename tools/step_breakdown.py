#!/usr/bin/env python3
"""Per-step breakdown of a rocprofv3 kernel trace of bench.py (developer tool): categories, GEMM list, idle time.
usage: step_breakdown.py <kernel_trace.csv> [step index] [--gemms] [--glue]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']), int(r['Workgroup_Size_X'])) for r in rows)
idx = [i for i, e in enumerate(ev) if 'multi_tensor_apply' in e[2] or 'adam_kernel' in e[2]]      # the optimizer update closes a step (torch's fused Adam or csrc/adam.hip)
groups = []
for i in idx:
    if groups and i - groups[-1][-1] <= 3: groups[-1].append(i)
    else: groups.append([i])
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(groups) - 2
s, e = groups[k][-1] + 1, groups[k + 1][-1]
t0 = ev[s][0]
def cat(n):
    if 'lstm_' in n: return 'lstm recurrence'
    if 'adam_kernel' in n: return 'adam (own)'
    if n.startswith(('Cijk', 'Custom_Cijk')): return 'gemm bf16' if ('_BSS_BH' in n or '_BBS_BH' in n) else 'gemm f32'    # BSS: bf16 in, fp32 out; BBS: bf16 out
    if 'split_bf16' in n: return 'operand split'
    if 'tsg::' in n: return 'hot-path kernels'
    if 'multi_tensor' in n: return 'adam'
    return 'torch glue'
c = collections.defaultdict(lambda: [0, 0.0])
for x in ev[s:e + 1]:
    c[cat(x[2])][0] += 1; c[cat(x[2])][1] += (x[1] - x[0]) / 1e6
span = (ev[e][1] - ev[s][0]) / 1e6
busy = sum(v[1] for v in c.values())
print(f"step {k}: {e - s + 1} launches, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms")
for name, v in sorted(c.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:18s} {v[0]:4d} launches {v[1]:7.2f} ms")
if '--gemms' in sys.argv:
    for x in ev[s:e + 1]:
        if x[2].startswith('Cijk'):
            mt = re.search(r'MT(\d+x\d+x\d+)', x[2]).group(1)
            print(f"  +{(x[0] - t0) / 1e6:6.2f} ms {'bf16' if '_BSS_BH' in x[2] else 'f32 '} {x[2][5:14]} MT{mt:12s} wgs {x[3] // x[4]:5d} {(x[1] - x[0]) / 1e3:7.1f} us")
if '--glue' in sys.argv:
    g = collections.defaultdict(lambda: [0, 0.0])
    for x in ev[s:e + 1]:
        if cat(x[2]) == 'torch glue':
            short = re.sub(r'at::native::|\(anonymous namespace\)::|void ', '', x[2])[:80]
            g[(short, x[3])][0] += 1; g[(short, x[3])][1] += (x[1] - x[0]) / 1e3
    for kk, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"  {v[1]:8.1f} us {v[0]:3d}x threads {kk[1]:>10}  {kk[0]}")
