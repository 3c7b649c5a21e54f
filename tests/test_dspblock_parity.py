"""Row a0 of SURVEY section 8: the host runtime's DspBlock::connect/start/run/stop must
behave exactly like the reference's.  One scenario harness (oracle/ref_harness.cxx) is
compiled against the REAL reference dspblock.cxx (oracle/_ref) and against this repo's
webradio_amd/host/dspblock.cxx; the recorded traces must be identical.  The reference
traces are also committed (tests/golden/dspblock_traces.json) so the check runs where
/root/reference does not exist."""
import ctypes as C
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "webradio_amd", "host")
CXXT = os.path.join(ROOT, "tests", "cxx")


@pytest.fixture(scope="module")
def host_harness():
    lib = os.path.join(CXXT, "libwr_host_harness.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", CXXT, "libwr_host_harness.so"])
    H = C.CDLL(lib)
    H.wr_harness_run.restype = C.c_long
    H.wr_harness_run.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    return H


def test_traces_match_committed_reference_traces(host_harness, oracle):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dspblock_traces.json")))
    assert host_harness.wr_harness_scenarios() == len(gold) >= 7
    for i, want in enumerate(gold):
        got = oracle.harness_trace(host_harness, i)
        assert got == want, "scenario %d differs from the reference:\n%s\n--- reference ---\n%s" % (i, got, want)


def test_traces_match_live_reference(host_harness, oracle):
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    for i in range(R.wr_harness_scenarios()):
        assert oracle.harness_trace(host_harness, i) == oracle.harness_trace(R, i)
