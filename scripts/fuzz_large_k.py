#!/usr/bin/env python3
"""More seeds of the large-k differential fuzz (tests/test_gpu_fuzz.py large_k_walk) than the test suite runs:
    python scripts/fuzz_large_k.py [first_seed] [seeds]      (on the GPU box; 500 seeds take four minutes)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: one HIP runtime in the process)

import test_gpu_fuzz as F  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0, bad, stats, done = time.time(), 0, {}, 0
for seed in range(first, first + count):
    try:
        F.large_k_walk(seed, stats)
    except AssertionError as e:
        bad += 1
        print("FAIL seed %d: %s" % (seed, str(e)[:300]), flush=True)
    done += 1
    if time.time() - t0 > 1500:
        print("stopped at seed", seed)
        break
print("large-k fuzz: %d tables, %d queries against the oracle, failures %d; shard queries scanned by the publishing launch %d, handed back %d (reasons %d); %.0f s"
      % (done, stats.get("queries", 0), bad, stats.get("published", 0), stats.get("handed_back", 0), stats.get("why", 0), time.time() - t0))
