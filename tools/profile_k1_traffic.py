#!/usr/bin/env python3
"""PMC pass of the K1g forward the f32s step launches (scdm_fwd_ws_kernel, 128 and 64 pairs per launch): runs rocprofv3 twice per
launch size (FETCH_SIZE, WRITE_SIZE: separate passes) on tools/k1_fwd_only.py and writes profiles-style JSON (the format of
profiles/r3/k1_pmc_traffic.json that bench.py cites in roofline.traffic_source).   python3 tools/profile_k1_traffic.py OUT.json"""
import csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, N, d = 128, 20, 1024


def counter(pmc, pairs):
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        subprocess.run(["rocprofv3", "--pmc", pmc, "--kernel-trace", "--output-format", "csv", "-d", td, "-o", "p", "--",
                        "python3", os.path.join(ROOT, "tools", "k1_fwd_only.py"), str(pairs), "8", "2"], check=True, capture_output=True,
                       env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        f = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)[0]
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "scdm_fwd_ws_kernel" in r["Kernel_Name"] and r["Counter_Name"] == pmc]
    return sum(v) / len(v), len(v)


out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/k1_fwd_only.py <pairs> 8 2 ; second pass with --pmc WRITE_SIZE "
                  "(tools/profile_k1_traffic.py; means over the 8 launches)",
       "correction": "gfx950: FETCH_SIZE counts half the bytes of a streaming read -> x2 (MI355X_MICROARCH.md; calibrated in round 2: tools/ubench/read_width.hip); "
                     "WRITE_SIZE exact. Counter unit KiB",
       "shape": {"B": 64, "T": T, "N": N, "d": d, "dtype": "f32"}, "kernels": {}}
for pairs in (128, 64):
    fetch, n = counter("FETCH_SIZE", pairs)
    write, _ = counter("WRITE_SIZE", pairs)
    rb, wb = int(fetch * 1024 * 2), int(write * 1024)
    alg = pairs * ((3 * T + 2 * N) * d * 4 + T * N * 4)
    out["kernels"][f"scdm_fwd_kernel[gate]@B{pairs}"] = {
        "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "read_bytes": rb, "write_bytes": wb, "hbm_bytes_per_launch": rb + wb, "algorithmic_bytes": alg,
        "traffic_over_algorithmic": round((rb + wb) / alg, 4), "launches": n,
        "note": f"dtype TSG_F32S: scdm_fwd_ws_kernel, {pairs} pairs per launch (tree of the run)"}
out["kernels"]["scdm_fwd_kernel[gate]"] = dict(out["kernels"]["scdm_fwd_kernel[gate]@B64"], note="= the @B64 entry (scaled by pairs / 64 for other launch sizes)")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: v["traffic_over_algorithmic"] for k, v in out["kernels"].items()}))
