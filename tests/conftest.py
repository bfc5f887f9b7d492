import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The library built with -DGSIM_TEST_HOOKS (GSIM_TEST_ALIAS_DEVICES, GSIM_TEST_TORN_EVERY): the shipped libgsim_hip.so has no
# test hooks, so the tests that need one run a child process against this build -- Python children through GSIM_LIB,
# gpusimserver through LD_LIBRARY_PATH (same soname, found before the binary's RUNPATH).
HOOKS_DIR = os.path.join(ROOT, "gpusimilarity_amd", "testhooks")
HOOKS_LIB = os.path.join(HOOKS_DIR, "libgsim_hip.so")


def hooks_env(**extra):
    """Environment of a child process that is to load the test-hooks build of the library."""
    env = dict(os.environ, GSIM_LIB=HOOKS_LIB, **extra)
    env["LD_LIBRARY_PATH"] = HOOKS_DIR + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    return env


# Which parity level did the full-size tests reach on THIS box?  The tests record it, the session prints it after the
# summary line -- the driver's record keeps the tail of the output, and "249 passed" alone does not say whether the
# 1 B-row table was compared with the oracle over all its rows or only through its size-independent properties.
PARITY_LEVELS = []


def record_parity(test, level, detail=""):
    PARITY_LEVELS.append((test, level, detail))


def pytest_terminal_summary(terminalreporter):
    if not PARITY_LEVELS:
        return
    terminalreporter.write_line("parity level reached by the full-size tests on this box:")
    for test, level, detail in PARITY_LEVELS:
        terminalreporter.write_line("  %-58s %-18s %s" % (test, level, detail))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
