"""The synthetic table generators: the product's host twin (gsim_synth_row, csrc/gsim_synth.h -- the
same code the device kernels run) against the oracle's restatement (gso_synth_*), and the shape
of the Morgan-like kind against what the reference's fixture test/small.fsim looks like
(python/gpusim_utils.py:21,55-66: 1024-bit Morgan r=2; popcounts 20..53, mean 34.5)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd import capi
from gpusimilarity_amd.fsim import read_fsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND_MORGAN = 2


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("W", [1, 5, 32, 64, 100])
def test_host_twin_equals_oracle(kind, W):
    seed = 0x5EED0001 + W
    for first in (0, 1023, 1 << 20, (1 << 32) - 7, 999_999_999):
        want = O.synth_rows(seed, kind, first, 9, W)
        got = np.stack([capi.synth_row(seed, kind, first + i, W * 32) for i in range(9)])
        assert (got == want).all(), (kind, W, first)


def test_synth_word_serves_the_morgan_kind():
    L = O.lib()
    want = O.synth_rows(7, KIND_MORGAN, 12345, 1, 32)[0]
    got = [L.gso_synth_word(7, KIND_MORGAN, 12345, 32, j) for j in range(32)]
    assert got == list(want)


def _popc(rows):
    return np.unpackbits(np.ascontiguousarray(rows).view(np.uint8), axis=1).sum(1)


def test_morgan_kind_has_the_fixtures_shape():
    n = 200_000
    db = O.synth_rows(0x5EED0001, KIND_MORGAN, 0, n, 32)
    pc = _popc(db)
    f = read_fsim(os.path.join(ROOT, "tests", "golden", "small.fsim"))
    ref = _popc(np.concatenate(f.fp_blocks).astype(np.uint32))
    assert abs(pc.mean() - ref.mean()) < 3.0, (pc.mean(), ref.mean())
    assert pc.min() >= 12 and pc.max() <= 56
    lo, hi = np.percentile(pc, [1, 99])
    assert lo >= ref.min() - 3 and hi <= ref.max() + 3
    # a dozen bits in more than half of the rows, a long tail of rare ones
    freq = np.unpackbits(db.view(np.uint8), axis=1, bitorder="little").mean(0)
    assert 8 <= (freq > 0.5).sum() <= 16 and (freq > 0.1).sum() < 80 and (freq > 0).sum() > 900
    # exact duplicates: a few per cent of the rows
    distinct = len(np.unique(db, axis=0))
    assert 0.01 < 1 - distinct / n < 0.06
    # series of analogs: some queries have hundreds of rows above 0.5, most have a handful; scores are coarse
    big = ties = 0
    for qi in range(24):
        q = db[O.query_row(qi, n)]
        sc, _, _ = O.tanimoto_raw(q, db)
        big += int((sc >= 0.5).sum() > 100)
        o = np.sort(sc)[::-1]
        ties += int((sc == o[999]).sum() > 10)
        assert len(np.unique(o[:1000])) < 200
    assert big >= 2 and ties >= 12


def test_generate_rejects_unknown_kind():
    L = capi.load()
    import ctypes as C
    out = (C.c_uint32 * 32)()
    assert L.gsim_synth_row(1, 3, 0, 1024, out) == -1
    assert L.gsim_synth_row(1, 0, 0, 1000, out) == -1
