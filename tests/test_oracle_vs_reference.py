"""Pins the oracle to the REAL reference where the reference builds from its own sources
(dsp/dspblock.cxx + dsp/demodulator.cxx -> oracle/_ref/libwr_ref.so), live and through
the golden fixtures that were generated from it (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = ["AM", "FM", "USB", "LSB"]


def _need_ref(oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")


@pytest.mark.parametrize("mode", MODES)
def test_demod_bit_exact_vs_live_reference(oracle, mode):
    _need_ref(oracle)
    rng = np.random.default_rng(10 + MODES.index(mode))
    iq = rng.standard_normal(2 * 8192).astype(np.float32)
    iq[:20] = 0.0                       # signed-zero atan2f cases at the start
    iq[40:44] = [-0.0, 1.0, 0.0, -1.0]
    ref = oracle.ref_demod(mode, iq, 1024)
    got, _ = oracle.demod(MODES.index(mode), (0.0, 0.0), iq)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))


def test_demod_state_carries_across_blocks_and_mode_switch(oracle):
    _need_ref(oracle)
    rng = np.random.default_rng(99)
    iq = rng.standard_normal(2 * 4096).astype(np.float32)
    ref = oracle.ref_demod("AM", iq, 512, switch_at=3, mode2="FM")
    a, prev = oracle.demod(oracle.AM, (0.0, 0.0), iq[: 2 * 1536])
    b, _ = oracle.demod(oracle.FM, prev, iq[2 * 1536:])
    assert np.array_equal(ref, np.concatenate([a, b]))


@pytest.mark.parametrize("mode", MODES)
def test_demod_golden_fixture(oracle, mode):
    """Committed vectors produced by the reference's own Demodulator (no _ref needed)."""
    g = np.load(os.path.join(GOLDEN, "demod_reference.npz"))
    got, _ = oracle.demod(MODES.index(mode), (0.0, 0.0), g["iq"])
    assert np.array_equal(got.view(np.uint32), g["out_" + mode].view(np.uint32))


def test_dspblock_trace_golden(oracle):
    """The DspBlock scheduling traces of the real reference are committed; when _ref is
    available they must still be what it produces."""
    _need_ref(oracle)
    import json
    gold = json.load(open(os.path.join(GOLDEN, "dspblock_traces.json")))
    R = oracle.ref()
    assert R.wr_harness_scenarios() == len(gold)
    for i, want in enumerate(gold):
        assert oracle.harness_trace(R, i) == want
