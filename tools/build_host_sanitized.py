#!/usr/bin/env python3
"""Host-side sanitizer build of libtsg_hip (round-5 review item 8).  CPU ONLY: this file is listed in .gpurunignore and never travels to a
GPU box (the pool refuses sanitizer builds there, and nothing on the GPU needs it).

The C-ABI host layer -- argument checks, workspace planners (tsg_*_ws_bytes), tile / grid arithmetic -- is compiled WITHOUT device code
(hipcc --cuda-host-only) under AddressSanitizer + UndefinedBehaviorSanitizer.  The host objects still reference the device code objects their kernels
would be registered from (__hip_fatbin_<hash>); an empty offload bundle is linked in their place -- nothing is ever launched from this library:
tests/abi_host_driver.py only calls entry points on paths that return before a launch.
    python tools/build_host_sanitized.py [--jobs N]      ->  shufflingvideosfortsg_amd/libtsg_hip_host_asan.so"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shufflingvideosfortsg_amd import build as B  # noqa: E402

SAN_LIB = os.path.join(B.HERE, "libtsg_hip_host_asan.so")
SAN_OBJ = os.path.join(B.CSRC, "obj", "san")
SAN_FLAGS = ["--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer", "-Wno-unused-function", "-ffp-contract=fast", "-fno-math-errno"]


def asan_runtime() -> str:
    """Path of the shared AddressSanitizer runtime the sanitizer build needs preloaded (LD_PRELOAD) into an uninstrumented python."""
    clang = os.path.join(os.path.dirname(os.path.realpath(B.HIPCC)), "..", "lib", "llvm", "bin", "clang")
    r = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return os.path.realpath(r.stdout.strip())


def build_sanitized(jobs: int = 8) -> str:
    os.makedirs(SAN_OBJ, exist_ok=True)

    def one(src):
        obj = os.path.join(SAN_OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
        stamp = obj + ".sha1"
        want = B._deps_hash(src) + "san"
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
            return obj
        r = subprocess.run([B.HIPCC, *SAN_FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"sanitizer build failed on {src}:\n{r.stderr}")
        open(stamp, "w").write(want)
        return obj
    with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(one, B._sources()))
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
    r = subprocess.run([B.HIPCC, "--cuda-host-only", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libasan", *objs, stub_o, "-o", SAN_LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"sanitizer link failed:\n{r.stderr}")
    return SAN_LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    print(build_sanitized(ap.parse_args().jobs))
