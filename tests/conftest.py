import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# r04: the suite runs the launch heuristic that ships.  (r03 forced WR_DDC_NG2_MIN_PASSES=0 for every test, so C5's and
# C1's NG = 1 launches were never the ones compared with the oracle.)  The modules below run every test twice instead:
# "shipped" -- the library's own threshold, what bench.py and the host runtime launch -- and "ng2" -- two lane groups
# per wave whenever the launch allows (wr_tune), which the tests' short streams would otherwise never reach.
NG2_MODULES = ("test_gpu_tuner", "test_gpu_timeshard", "test_gpu_ring", "test_gpu_fuzz")

# Oracle-parity modules first, process-spawning contract tests last: under `-x` a hiccup in a launcher test must not
# cost the parity run (GPUTEST_r03).
ORDER = ["test_gpu_blocks", "test_gpu_tuner", "test_gpu_reference_pin", "test_gpu_spectrum", "test_gpu_stage", "test_gpu_f4",
         "test_gpu_fuzz", "test_gpu_ring", "test_gpu_stream", "test_gpu_timeshard", "test_gpu_host", "test_gpu_c_client", "test_gpu_bench"]

PER_TEST_LIMIT_S = 300.0        # no test takes a tenth of this; see the watchdog below


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in NG2_MODULES and "dev" in metafunc.fixturenames:
        metafunc.parametrize("ddc_lane_groups", ["shipped", "ng2"], indirect=True)


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = item.module.__name__.split(".")[-1]
        return ORDER.index(name) if name in ORDER else -1        # CPU-side modules keep their place at the front
    if not os.environ.get("WR_TEST_KEEP_ORDER"):                   # (a hunt for an order-dependent failure wants its own order)
        items.sort(key=key)                                        # stable: order within a module is kept
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(PER_TEST_LIMIT_S))  # pytest-timeout, where it is installed


@pytest.fixture(autouse=True)
def _watchdog(request):
    """A test that is still running after PER_TEST_LIMIT_S + 60 s dumps every thread's stack and ends the
    run: a hang is a NAMED failure in the log, never a silent wait for the driver's limit.  (pytest-timeout's
    own signal fires first and fails just the one test; this is the backstop for code stuck in C.)"""
    import faulthandler
    sys.stderr.write("")                                            # (faulthandler writes to the real stderr)
    faulthandler.dump_traceback_later(PER_TEST_LIMIT_S + 60.0, exit=True)
    yield
    faulthandler.cancel_dump_traceback_later()


@pytest.fixture(autouse=True)
def ddc_lane_groups(request):
    """"ng2": k_tuner_ddc's two-lane-groups-per-wave variant whenever a launch allows it; "shipped": the
    library's threshold (four passes of the grid).  Only the NG2_MODULES' tests are parametrised with it."""
    which = getattr(request, "param", None)
    if which is None:
        yield None
        return
    import ctypes as C
    from webradio_amd import capi
    lib = capi.load()
    assert lib.wr_tune(1, 0 if which == "ng2" else -1, None) == 0
    yield which
    assert lib.wr_tune(1, -1, None) == 0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout)")
    # The built libraries are git-ignored; if this checkout has none (fresh clone), build them
    # once (hipcc cross-compiles gfx950 without a GPU).  Nothing is ever substituted for them.
    needed = [os.path.join(ROOT, "webradio_amd", "lib", "libwebradio_amd.so"),
              os.path.join(ROOT, "webradio_amd", "host", "libwebradio_host.so"),
              os.path.join(ROOT, "oracle", "libwr_oracle.so"),
              os.path.join(ROOT, "tests", "cxx", "libwr_host_pipeline.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


def _gpu_available():
    try:
        from webradio_amd import capi
        import ctypes as C
        n = C.c_int()
        return capi.load().wr_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def dev():
    """A wr_dev on device 0.  GPU tests FAIL (not skip) when the HIP library cannot be
    loaded on a box that has a GPU; they are only deselected by -m "not gpu"."""
    from webradio_amd.device import Device
    d = Device(0)
    yield d
    d.close()


# ---- host memory that tests page-lock (wr_dev_host_register) ---------------------------------------------------------------
# One anonymous mapping made when this file is imported -- before the process has copied anything to or from the GPU -- and
# never unmapped: every buffer a test registers is a page-aligned piece of it with a guard page behind (a test's pieces
# go back to the pool when it is over: the pool's pages are never anybody else's).
# Why: a registered numpy array from the heap shares its first and last page with whatever the allocator puts beside it, and
# its address is handed out again after the test; the HIP runtime page-locks the destination of every copy into pageable
# memory on the fly and keeps those locks for a while.  One run in a dozen of the sequence test_gpu_stage -> test_gpu_blocks
# ABORTED inside a copy into a fresh numpy array next to registered ones (profiles/r04_gate_loop.txt, pass 1).  Pages that
# are only ever ours, at addresses no earlier allocation has had, cannot collide with any of that.
_POOL_BYTES = 256 << 20
_pool = None
_pool_used = 0


def _pinned_pool():
    global _pool
    if _pool is None:
        import mmap
        _pool = mmap.mmap(-1, _POOL_BYTES)
    return _pool


_pinned_pool()


@pytest.fixture
def page_locked(dev):
    """page_locked(count, dtype) -> a zero-filled numpy array in page-aligned memory of its own, registered with the
    device (wr_dev_host_register); unregistered again when the test ends (after the device has gone idle)."""
    import ctypes as C
    import numpy as np
    global _pool_used
    made = []
    mark = _pool_used                                                   # (handed back when the test is over: see below)

    def make(count, dtype=np.uint8):
        global _pool_used
        nbytes = int(count) * np.dtype(dtype).itemsize
        span = (nbytes + 4095) // 4096 * 4096 + 4096                     # + a guard page nobody else gets
        assert _pool_used + span <= _POOL_BYTES, "tests/conftest.py: the page-locked pool is exhausted"
        arr = np.frombuffer(_pinned_pool(), dtype=np.uint8, count=nbytes, offset=_pool_used).view(dtype)
        _pool_used += span
        arr[...] = 0
        assert dev.lib.wr_dev_host_register(dev.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes) == 0, dev.lib.wr_last_error()
        made.append(arr)
        return arr

    yield make
    dev.sync()
    dev.lib.wr_dev_wait_uploads(dev.h)
    for arr in made:
        dev.lib.wr_dev_host_unregister(dev.h, arr.ctypes.data_as(C.c_void_p))
    _pool_used = mark              # the pieces go back to the pool: pages that were only ever the pool's, unregistered, idle


@pytest.fixture(scope="session")
def oracle():
    import wr_oracle
    wr_oracle.lib()
    return wr_oracle
