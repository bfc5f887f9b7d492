"""Litmus tests of the hardware behaviours the single launch (csrc/gsim_fused.hip) rests on -- run with its own
instructions through gsim_debug_litmus (csrc/gsim_litmus.hip), 256 workgroups, >= 10^8 stores per test -- so that a ROCm or
firmware change shows up as a NAMED failure, not as a flaky soak.  What the product does with each:

  1  region entries and region headers are 16-byte sc1 stores that other workgroups read with 16-byte sc1 loads (also straight
     into LDS): a selector trusts an entry whose fourth word carries the launch's tag -- only sound if the store is seen whole;
  2  the closing workgroup announces a finished query with ONE 16-byte system-scope store into pinned host memory; the host
     polls its second word and then reads count and approx from the same 16 bytes;
  3  a workgroup's header is its arrival: it is stored after the entries without waiting for their acknowledgements, so a
     reader may see the header first -- it reads the entry again until it carries the tag (GSIM_FUSED_FLAGS=4096 forces that
     path in the parity suite).  Here: how often it happens on an idle GPU (reported), and that the re-read always succeeds.

Reference behaviour these protect: fingerprintdb_cuda.cu:228-339 returns complete, ordered hits for every query."""
import pytest

from gpusimilarity_amd import capi

pytestmark = pytest.mark.gpu


def test_litmus_16_byte_stores_are_seen_whole_by_other_workgroups():
    r = capi.litmus(1, workgroups=256, iterations=13000)  # 128 writers x 64 lanes x 13 000 = 1.06e8 stores
    print("litmus 1:", r)
    assert r["stores"] >= 100_000_000 and r["loads"] > 1_000_000, r
    assert r["torn"] == 0, "a 16-byte sc1 store was seen in pieces: %r" % (r,)
    assert r["timed_out"] == 0, r


def test_litmus_16_byte_system_scope_stores_are_whole_for_the_polling_host():
    r = capi.litmus(2, workgroups=256, iterations=4000)  # 1.02e6 header-like stores, every one observed and acknowledged by the host
    print("litmus 2:", r)
    assert r["loads"] == 256 * 4000 and r["stores"] == 256 * 4000, r
    assert r["torn"] == 0, "a header-like store reached host memory in pieces: %r" % (r,)
    assert r["timed_out"] == 0, r


@pytest.mark.parametrize("test", [3, 4])
def test_litmus_header_may_overtake_its_entry_and_a_reread_always_finds_it(test):
    """3: entry and header from the same lane; 4: the product's shape -- the entries from another wave, a workgroup barrier, the header."""
    r = capi.litmus(test, workgroups=256, iterations=13000)
    print("litmus %d:" % test, r, "-- entries behind their header at the first read: %.3g of the headers seen" % (r["stale_first_read"] / max(1, r["headers_seen"])))
    assert r["stores"] >= 100_000_000 and r["headers_seen"] > 100_000, r
    assert r["torn"] == 0 and r["never_landed"] == 0 and r["timed_out"] == 0, r
    # (no bound on the rate: it is what the tags are for; the rate of this box is in the test's output and in profiles/)
