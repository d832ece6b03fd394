#!/usr/bin/env python3
"""Shape of a captured HIP graph from its hipGraphDebugDotPrint dump: node kinds, whether it is a linear chain, the first and last nodes.
    TSG_GRAPH_DOT=/tmp/a.dot python tools/graph_grad_probe2.py f32s dot ; python tools/graph_dot_stats.py /tmp/a.dot"""
import re, sys, collections
txt = open(sys.argv[1]).read()
nodes = dict(re.findall(r'^\s*"?(\w+)"?\s*\[(.*?)\];', txt, flags=re.M | re.S))
edges = re.findall(r'^\s*"?(\w+)"?\s*->\s*"?(\w+)"?', txt, flags=re.M)
kinds = collections.Counter()
for n, attr in nodes.items():
    lab = re.search(r'label="([^"]*)"', attr)
    lab = lab.group(1) if lab else attr
    k = "memset" if "emset" in lab else "memcpy" if "emcpy" in lab else "kernel" if ("ernel" in lab or "(" in lab) else lab[:30]
    kinds[k] += 1
print(len(nodes), "nodes,", len(edges), "edges; kinds:", dict(kinds))
indeg, outdeg = collections.Counter(b for a, b in edges), collections.Counter(a for a, b in edges)
roots = [n for n in nodes if indeg[n] == 0]; leaves = [n for n in nodes if outdeg[n] == 0]
print("roots", len(roots), "leaves", len(leaves), "max in-degree", max(indeg.values(), default=0), "max out-degree", max(outdeg.values(), default=0))
def lab(n):
    m = re.search(r'label="([^"]*)"', nodes[n]); return (m.group(1) if m else nodes[n]).replace("\\n", " ")[:160]
for n in roots[:6]: print("root:", lab(n))
for n in leaves[:10]: print("leaf:", lab(n))
ms = [n for n in nodes if "emset" in nodes[n]]
for n in ms[:12]: print("memset:", lab(n), "| preds", [lab(a)[:50] for a, b in edges if b == n][:2], "| succs", [lab(b)[:50] for a, b in edges if a == n][:2])
