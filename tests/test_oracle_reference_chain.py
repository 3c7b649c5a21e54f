"""CPU: the oracle (oracle/wr_oracle.c) against vectors the REAL reference produced -- its own
dsp/downconverter.cxx, dsp/lowpass.cxx, dsp/demodulator.cxx and io/spectrumsink.cxx run through
oracle/ref_chain.cxx on the GPU box, with the image's hipFFTW behind their FFTW calls
(tests/golden/make_reference_chain_golden.py wrote tests/golden/reference_chain.npz there).

This pins SURVEY 8 rows a1-a4 and a6 of the oracle to the reference's code (r01-r03: "parity unpinned",
two readings by the same builder).  Stated plainly: the FFT under the reference here is rocFFT, not FFTW3;
what it computes -- a 64-point inverse DFT of a 0/1 spectrum, the N-point forward DFT of a frame -- is
defined by the API, and the tolerances below are the f32 rounding of either library."""
import os

import numpy as np
import pytest

import refcases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_chain.npz")


@pytest.fixture(scope="module")
def gold():
    assert os.path.exists(GOLDEN), "tests/golden/reference_chain.npz is missing (make_reference_chain_golden.py, on the GPU box)"
    return np.load(GOLDEN)


@pytest.mark.parametrize("pb,rate", refcases.LOWPASS)
def test_lowpass_taps(oracle, gold, pb, rate):
    want = gold["taps_%d_%d" % (pb, rate)]
    got = oracle.lowpass_design(pb, rate)
    assert want.shape == got.shape == (64,)
    assert np.abs(got - want).max() <= refcases.TAPS_TOL
    if oracle.lowpass_maxbin(pb, rate) == 0:
        assert not want.any() and not got.any()               # lowpass.cxx:167: the all-zero filter


@pytest.mark.parametrize("name", sorted(refcases.MIXES))
def test_mixer_bit_exact(oracle, gold, name):
    c = refcases.MIXES[name]
    iq = refcases.mix_input(c)
    assert np.array_equal(refcases.sha(iq), gold["sha_mix_" + name]), "this host generates a different input"
    got, _ = oracle.mix(oracle.sin_table(), 0, oracle.phase_step(c["if_hz"], c["fs"]), iq)
    want = gold["mix_" + name]
    # the same table (sinf of the same float), the same unfused float arithmetic, the same order: the same bits --
    # up to the host's own sinf (glibc picks an FMA variant by CPU): at most one table ulp
    assert got.shape == want.shape and np.abs(got - want).max() <= 1.3e-7
    assert np.mean(got.view(np.uint32) == want.view(np.uint32)) > 0.99


@pytest.mark.parametrize("name", sorted(refcases.CHAINS))
def test_receiver_chain(oracle, gold, name):
    c = refcases.CHAINS[name]
    iq = refcases.chain_input(c)
    assert np.array_equal(refcases.sha(iq), gold["sha_chain_" + name]), "this host generates a different input"
    rx = oracle.Receiver(c["fs"], c["if_hz"], c["cpb"], c["crate"], c["mode"], c["apb"], c["arate"])
    audio, chan, dem = [], [], []
    n = c["block"]
    for b in range(c["blocks"]):
        a, z, d = rx.run(iq[2 * n * b: 2 * n * (b + 1)])
        audio.append(a), chan.append(z), dem.append(d)
    audio, chan, dem = np.concatenate(audio), np.concatenate(chan), np.concatenate(dem)
    for got, key, tol in ((chan, "chan", refcases.CHAN_TOL), (dem, "demod", refcases.AUDIO_TOL), (audio, "audio", refcases.AUDIO_TOL)):
        want = gold["chain_%s_%s" % (name, key)]
        assert got.shape == want.shape and want.size > 0, key
        assert np.abs(got - want).max() <= tol, (key, float(np.abs(got - want).max()))
    assert np.abs(audio).max() > 1e-3                         # not a comparison of silences


@pytest.mark.parametrize("name", sorted(refcases.SPECTRA))
def test_spectrum_db(oracle, gold, name):
    c = refcases.SPECTRA[name]
    iq = refcases.spectrum_input(c)
    assert np.array_equal(refcases.sha(iq), gold["sha_spec_" + name]), "this host generates a different input"
    s = oracle.Spectrum(c["n"])
    n = c["block"]
    for b in range(c["blocks"]):
        s.process(iq[2 * n * b: 2 * n * (b + 1)])
    got, want = s.get(), gold["spec_" + name]
    strong = want >= want.max() - refcases.DB_MASK
    assert strong.sum() >= 3
    assert np.abs(got - want)[strong].max() <= refcases.DB_TOL
    assert int(np.argmax(got)) == int(np.argmax(want))
    if name == "n512_tone":
        assert int(np.argmax(want)) == 277                    # SURVEY 8a's probe: 256 + 21


def test_c1_capture_against_the_reference(oracle):
    """BASELINE config 1: the recorded RTL-SDR capture (131 072 bytes, io/rtlsdrtuner.cxx:106's (u8 - 128) / 128) through the
    oracle's Receiver against what the REAL DownConverter -> LowPass -> Demodulator -> LowPass gave for it on the GPU box
    (tests/golden/reference_c1.npz, made by tests/golden/make_c1_reference_golden.py).  r01-r04 compared the C1 tests'
    results with the oracle's own output for this capture; they compare with the reference's now."""
    from webradio_amd import synth
    path = os.path.join(os.path.dirname(GOLDEN), "reference_c1.npz")
    assert os.path.exists(path), "tests/golden/reference_c1.npz is missing (make_c1_reference_golden.py, on the GPU box)"
    g = np.load(path)
    c1, n = synth.C1, int(g["block_frames"])
    iq = oracle.u8_to_float(g["u8"])
    rx = oracle.Receiver(c1["input_rate"], c1["if_hz"], c1["chan_passband"], c1["chan_rate"], oracle.FM, c1["audio_passband"],
                         c1["audio_rate"])
    audio, chan, dem = [], [], []
    for b in range(iq.size // 2 // n):
        a, z, d = rx.run(iq[2 * n * b: 2 * n * (b + 1)])
        audio.append(a), chan.append(z), dem.append(d)
    for got, key, tol in ((np.concatenate(chan), "chan_iq", refcases.CHAN_TOL), (np.concatenate(dem), "demod", refcases.AUDIO_TOL),
                          (np.concatenate(audio), "audio", refcases.AUDIO_TOL)):
        assert got.shape == g[key].shape and g[key].size > 0, key
        assert np.abs(got - g[key]).max() <= tol, (key, float(np.abs(got - g[key]).max()))
    assert np.abs(g["audio"]).max() > 1e-3
