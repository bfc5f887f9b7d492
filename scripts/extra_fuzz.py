#!/usr/bin/env python3
"""More seeds of the differential fuzz (tests/test_gpu_fuzz.py) than the test suite runs:
python scripts/extra_fuzz.py [first_seed] [seeds]  -- on the GPU box; 300 seeds of each walk take a few minutes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: one HIP runtime in the process)

import test_gpu_fuzz as F  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t0 = time.time()
bad = 0
for name, fn in (("power-of-two widths", F.test_random_tables_and_queries_match_the_oracle),
                 ("odd widths", F.test_random_odd_width_tables_match_the_oracle)):
    done = 0
    for seed in range(first, first + count):
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print("FAIL (%s) seed %d: %s" % (name, seed, str(e)[:300]), flush=True)
        done += 1
        if time.time() - t0 > 1500:
            print("stopped at seed", seed)
            break
    print("%s: %d seeds, %.0f s" % (name, done, time.time() - t0), flush=True)
print("extra fuzz done, failures:", bad, "elapsed", round(time.time() - t0))
sys.exit(1 if bad else 0)
