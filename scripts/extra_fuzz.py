#!/usr/bin/env python3
"""More seeds of the differential fuzz (tests/test_gpu_fuzz.py) than the test suite runs:
python scripts/extra_fuzz.py  -- on the GPU box; 70 seeds take about a minute."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import test_gpu_fuzz as F
t0 = time.time()
bad = 0
for seed in range(100, 140):
    try:
        F.test_fuzz_large_batches_against_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("LARGE FAIL seed", seed, str(e)[:300])
    if time.time() - t0 > 500:
        print("stopped at seed", seed); break
for seed in range(100, 130):
    try:
        F.test_fuzz_against_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("FUZZ FAIL seed", seed, str(e)[:300])
    if time.time() - t0 > 1000:
        print("stopped at seed", seed); break
for seed in range(100, 130):
    try:
        F.test_fuzz_generic_widths_against_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("GENERIC FAIL seed", seed, str(e)[:300])
    if time.time() - t0 > 1500:
        print("stopped at seed", seed); break
print("extra fuzz done, failures:", bad, "elapsed", round(time.time() - t0))
