"""Build libtsg_hip.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

    python -m shufflingvideosfortsg_amd.build [--force] [--jobs N] [--verbose]

hipcc cross-compiles without a GPU.  One object per csrc/*.hip (parallel), linked into
shufflingvideosfortsg_amd/libtsg_hip.so.  The library's only runtime dependency is
libamdhip64.so.7, which resolves to the copy PyTorch-ROCm has already loaded (same SONAME), so
torch's streams and device pointers are valid inside it.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libtsg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-ffp-contract=fast", "-fno-math-errno"]
# per-source additions.  wgrad_split: under plain -O3 the SLP vectoriser packs the split's subtractions into v_pk_add_f32, which is slow
# beside MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs"): the LSTM layer's dW 626 -> 595 us without it
# (profiles/r4/gemm_wgrad_flags_ab_v1.txt; tsg_gemm_f32s measured the same either way and keeps the default)
EXTRA_FLAGS = {"wgrad_split.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_hash(src: str) -> str:
    h = hashlib.sha1()
    for p in [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            [os.path.join(HERE, "..", "include", "tsg_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(CXXFLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])).encode())
    return h.hexdigest()


def _compile(src: str, force: bool, verbose: bool) -> str:
    os.makedirs(OBJ, exist_ok=True)
    base = os.path.splitext(os.path.basename(src))[0]
    obj, stamp = os.path.join(OBJ, base + ".o"), os.path.join(OBJ, base + ".sha1")
    want = _deps_hash(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build(force: bool = False, jobs: int = 4, verbose: bool = False) -> str:
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.jobs, a.verbose))
