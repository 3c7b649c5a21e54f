"""CPU: the conditional FM bound of tests/fm_bound.py holds for the oracle itself -- channel IQ
perturbed by differences of the size the ROTATE mode makes (1e-7 ... 1e-6) moves the oracle's own
FM audio by no more than the bound says, on channels with a carrier and on noise-only ones -- and is
not vacuous: on a carrier channel it stays below the AUDIO_ATOL the tests used to assert there."""
import numpy as np
import pytest

import fm_bound

FM = 1


def _chain(oracle, iq, taps2, d2):
    dem, _ = oracle.demod(FM, (0.0, 0.0), iq)
    return dem, oracle.Fir(1, d2, taps2).process(dem)


@pytest.mark.parametrize("level,eps", [(0.4, 1.3e-7), (0.4, 1e-6), (2e-3, 1.3e-7), (2e-3, 1e-6), (3e-5, 1e-6)])
def test_bound_holds_for_perturbed_oracle_input(oracle, level, eps):
    rng = np.random.default_rng(int(level * 1e6) + int(eps * 1e9))
    n, d2 = 20_000, 5
    taps2 = oracle.lowpass_design(8_000, 250_000)
    # a channel's output: a slowly turning carrier of the given level plus Rayleigh noise a tenth of it
    # (level 2e-3 / 3e-5: what the channel filter leaves of the -40 dBFS noise floor, and a deep fade of it)
    t = np.arange(n)
    z = level * np.exp(1j * (0.3 * t + 2.0 * np.sin(0.01 * t)))
    z += (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * level * (0.1 if level > 0.1 else 1.0)
    want = np.empty(2 * n, np.float32)
    want[0::2], want[1::2] = z.real, z.imag
    dz = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * eps / np.sqrt(2)
    got = want.copy()
    got[0::2] += dz.real.astype(np.float32)
    got[1::2] += dz.imag.astype(np.float32)
    wd, wa = _chain(oracle, want, taps2, d2)
    gd, ga = _chain(oracle, got, taps2, d2)
    worst, ratio = fm_bound.assert_fm_audio_within_iq_bound(want, got, wa, ga, taps2, d2, want_demod=wd)
    assert ratio <= 1.0
    b = fm_bound.audio_bound(want, got, taps2, d2, wd)
    if level > 0.1:
        assert np.median(b) < 1e-5 and b.max() < 1e-5          # a carrier: tighter than the old AUDIO_ATOL
    if level == 2e-3 and eps < 5e-7:
        assert np.median(b) < 1e-3                              # noise floor: still a real statement


def test_bound_sees_audio_that_its_iq_does_not_explain(oracle):
    """Audio 8e-6 cycles off while the channel IQ is identical: outside the bound (a fault behind the channel
    filter cannot hide in it)."""
    n, d2 = 4_000, 5
    taps2 = oracle.lowpass_design(8_000, 250_000)
    t = np.arange(n)
    z = 0.4 * np.exp(1j * (0.3 * t))
    want = np.empty(2 * n, np.float32)
    want[0::2], want[1::2] = z.real, z.imag
    wd, wa = _chain(oracle, want, taps2, d2)
    # audio computed from a DIFFERENT signal than the IQ handed to the bound: what a kernel bug past the
    # channel filter would look like
    z2 = z * np.exp(1j * 1e-3 * np.sin(0.05 * t))        # 8e-6 cycles of frequency error
    other = np.empty(2 * n, np.float32)
    other[0::2], other[1::2] = z2.real, z2.imag
    _, oa = _chain(oracle, other, taps2, d2)
    with pytest.raises(AssertionError):
        fm_bound.assert_fm_audio_within_iq_bound(want, want, wa, oa, taps2, d2, want_demod=wd)
