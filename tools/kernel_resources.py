#!/usr/bin/env python3
"""Developer tool: per-kernel VGPR / spill / occupancy table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py scdm_attn.hip [substring filter]"""
import os, re, subprocess, sys
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "shufflingvideosfortsg_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-math-errno",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_res.o"], capture_output=True, text=True)
cur, rows = None, {}
for l in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    if flt in name:
        print(f"{name[-70:]:70s}", v)
