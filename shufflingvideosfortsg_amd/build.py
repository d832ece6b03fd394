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


# ---- host-side sanitizer build (CPU only, never loaded on a GPU box) ---------------------------------------------------------
# The C-ABI host layer -- argument checks, workspace planners (tsg_*_ws_bytes), tile / grid arithmetic -- compiled WITHOUT device code
# (hipcc --cuda-host-only) under AddressSanitizer + UndefinedBehaviorSanitizer.  The host objects still reference the device code objects
# their kernels would be registered from (__hip_fatbin_<hash>); an empty offload bundle is linked in their place -- nothing is ever launched
# from this library: tests/abi_host_driver.py only calls entry points on paths that return before a launch.
SAN_LIB = os.path.join(HERE, "libtsg_hip_host_asan.so")
SAN_OBJ = os.path.join(CSRC, "obj", "san")
SAN_FLAGS = ["--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer", "-Wno-unused-function", "-ffp-contract=fast", "-fno-math-errno"]


def asan_runtime() -> str:
    """Path of the shared AddressSanitizer runtime the sanitizer build needs preloaded (LD_PRELOAD) into an uninstrumented python."""
    clang = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang")
    r = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return os.path.realpath(r.stdout.strip())


def build_sanitized(jobs: int = 8) -> str:
    os.makedirs(SAN_OBJ, exist_ok=True)

    def one(src):
        obj = os.path.join(SAN_OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
        stamp = obj + ".sha1"
        want = _deps_hash(src) + "san"
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
            return obj
        r = subprocess.run([HIPCC, *SAN_FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"sanitizer build failed on {src}:\n{r.stderr}")
        open(stamp, "w").write(want)
        return obj
    with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(one, _sources()))
    # empty offload bundles for the device code objects the host stubs would register
    nm = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True).stdout
    syms = sorted({ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin_" in ln})
    stub = os.path.join(SAN_OBJ, "no_device_code.S")
    with open(stub, "w") as f:
        f.write('\t.section .hip_fatbin,"a",@progbits\n')
        for sy in syms:
            f.write(f'\t.globl {sy}\n\t.p2align 12\n{sy}:\n\t.ascii "__CLANG_OFFLOAD_BUNDLE__"\n\t.quad 0\n')
    stub_o = stub[:-2] + ".o"
    subprocess.run(["gcc", "-c", stub, "-o", stub_o], check=True)
    r = subprocess.run([HIPCC, "--cuda-host-only", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libasan", *objs, stub_o, "-o", SAN_LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"sanitizer link failed:\n{r.stderr}")
    return SAN_LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--sanitize", action="store_true", help="build the host-only ASAN + UBSAN library (CPU tests) instead")
    a = ap.parse_args()
    print(build_sanitized(a.jobs) if a.sanitize else build(a.force, a.jobs, a.verbose))
