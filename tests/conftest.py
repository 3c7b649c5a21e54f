import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The lean DDC loop runs two lane groups per wave (k_tuner_ddc: NG = 2) only for streams long enough to fill the
# grid four times over; the tests' streams are short, so they ask for it whenever the launch allows -- an even number
# of lane groups on one channel filter -- and cover NG = 1 through odd group counts and mixed passbands.
os.environ.setdefault("WR_DDC_NG2_MIN_PASSES", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
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


@pytest.fixture(scope="session")
def oracle():
    import wr_oracle
    wr_oracle.lib()
    return wr_oracle
