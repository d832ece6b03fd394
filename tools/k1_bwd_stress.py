#!/usr/bin/env python3
"""Developer tool: the K1g backward's partner exchange under stress -- LAUNCHES (default 2 000) launches per batch size (1 .. 16 column
parts per item) with a NaN-poisoned exchange workspace before every launch, two alternating operand sets, caches dirtied every 7th
launch; the first launch of each set is checked against the two-kernel path (no exchange), every later one must be bit-identical.
    python tools/k1_bwd_stress.py [launches]        (same machinery as tests/test_scdm_gpu.py::test_scdm_bwd_exchange_is_reproducible)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from shufflingvideosfortsg_amd import _lib
from test_scdm_gpu import _k1_bwd_exchange_run
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for B in (3, 9, 17, 40, 128):
    t0 = time.time()
    done = _k1_bwd_exchange_run(lib, B, True, n, dirty_every=7)
    print(f"B {B}: {done} poisoned-workspace launches clean ({time.time() - t0:.1f} s)", flush=True)
print("K1 BWD STRESS OK")
