"""The C oracle (oracle/wr_oracle.c) fuzzed against a second, independent restatement of the same
reference functions (tests/restate_np.py: numpy float32 array arithmetic + numpy's FFT, written from
the reference text by another route).  DownConverter, LowPass and SpectrumSink cannot be pinned to a
build of the reference here (they need FFTW); two independent readings that agree bit for bit --
wherever only float multiplies and adds are involved -- at least rule out a slip in either one.
This is NOT a reference pin."""
import numpy as np
import pytest

import restate_np as R

MODES = ["AM", "FM", "USB", "LSB"]


def _ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def test_sin_table_two_readings(oracle):
    t, r = oracle.sin_table(), R.sin_table()
    # libm's sinf (what the reference calls, whatever its version) vs the correctly rounded sine of the
    # same float argument: never more than one ulp apart, and equal for all but a per cent or two
    bad = _ulp_diff(t, r) > 1
    assert not np.any(bad & (np.abs(t - r) > 1e-12))
    assert np.count_nonzero(t != r) < 0.02 * t.size


def test_phase_step_two_readings(oracle):
    rng = np.random.default_rng(1)
    for _ in range(2000):
        rate = int(rng.integers(8_000, 1_000_000_001))
        hz = int(rng.integers(-rate // 2, rate // 2 + 1))
        assert oracle.phase_step(hz, rate) == R.phase_step(hz, rate)


def test_mixer_two_readings_bit_exact(oracle):
    rng = np.random.default_rng(2)
    table = oracle.sin_table()
    for _ in range(40):
        n = int(rng.integers(1, 3000))
        iq = rng.standard_normal(2 * n).astype(np.float32)
        step = int(rng.integers(-(1 << 30), 1 << 30))
        phase = int(rng.integers(0, 1 << 31))
        a, pa = oracle.mix(table, phase, step, iq)
        b, pb = R.mix(table, phase, step, iq)
        assert pa == pb
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("length", [16, 32, 64, 128, 256])
def test_lowpass_design_two_readings(oracle, length):
    rng = np.random.default_rng(3 + length)
    w = oracle.lowpass_window_n(length)
    # cosf vs the correctly rounded cosine: at most an ulp in the window, i.e. 1.2e-7 / L
    assert np.abs(w - R.lowpass_window(length)).max() <= 1.3e-7 / length
    for _ in range(60):
        rate = int(rng.integers(8_000, 1_000_000_001))
        pb = int(rng.integers(0, rate))
        assert oracle.lowpass_maxbin_n(length, pb, rate) == R.lowpass_maxbin(pb, rate, length)
        a = oracle.lowpass_design(pb, rate, length)
        b = R.lowpass_design(pb, rate, length)
        # two inverse DFTs in double, each narrowed once, times windows an ulp apart
        assert np.abs(a - b).max() <= 4e-7 * max(1.0, float(np.abs(a).max()) * length) / length


def test_lowpass_q6_wrap_two_readings(oracle):
    # 64 * passband wraps modulo 2^32 (lowpass.cxx:167); C5's 64 MHz just fits
    for pb, rate in ((70_000_000, 1_000_000), (64_000_000, 1_000_000_000), (67_108_864, 100_000_000)):
        assert oracle.lowpass_maxbin(pb, rate) == R.lowpass_maxbin(pb, rate)


@pytest.mark.parametrize("channels", [1, 2])
def test_fir_two_readings_bit_exact(oracle, channels):
    """LowPass::process incl. history across ragged blocks: multiplies and adds only -> same bits."""
    rng = np.random.default_rng(4 + channels)
    for _ in range(25):
        L = int(rng.choice([2, 8, 16, 64, 128]))
        d = int(rng.integers(1, 50))
        coeff = (rng.standard_normal(L) / L).astype(np.float32)
        a, b = oracle.Fir(channels, d, coeff), R.Fir(channels, d, coeff)
        for _blk in range(4):
            frames = int(rng.integers(0, 12)) * d      # blocks are whole multiples of the decimation
            x = rng.standard_normal(frames * channels).astype(np.float32)
            ya, yb = a.process(x), b.process(x)
            assert ya.size == yb.size
            assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))


def test_demod_two_readings(oracle):
    rng = np.random.default_rng(5)
    codes = {"AM": oracle.AM, "FM": oracle.FM, "USB": oracle.USB, "LSB": oracle.LSB}
    for m in MODES:
        prev_a = prev_b = (0.0, 0.0)
        for _ in range(6):
            n = int(rng.integers(1, 600))
            iq = rng.standard_normal(2 * n).astype(np.float32)
            a, prev_a = oracle.demod(codes[m], prev_a, iq)
            b, prev_b = R.demod(m, prev_b, iq)
            assert prev_a == (float(prev_b[0]), float(prev_b[1]))
            if m == "FM":
                assert np.abs(a - b).max() <= 6e-8        # atan2f within an ulp of pi, then / 2 pi
            else:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n", [8, 64, 512, 4096])
def test_spectrum_two_readings(oracle, n):
    rng = np.random.default_rng(6 + n)
    assert np.abs(oracle.spectrum_window(n) - R.spectrum_window(n)).max() <= 1.3e-7
    a, b = oracle.Spectrum(n), R.Spectrum(n)
    # ragged pushes: partial frames carry over between calls (spectrumsink.cxx:101-121)
    stream = rng.standard_normal(2 * (3 * n + n // 3)).astype(np.float32)
    stream[0::2] += np.cos(2 * np.pi * 0.1 * np.arange(stream.size // 2)).astype(np.float32) * 4
    pos = 0
    while pos < stream.size // 2:
        take = int(rng.integers(1, n))
        a.process(stream[2 * pos: 2 * (pos + take)])
        b.process(stream[2 * pos: 2 * (pos + take)])
        pos += take
    assert a.frames_done == b.frames == 3
    bins = a.bins()
    ref = b.out
    scale = float(np.abs(ref).max())
    assert np.abs(bins[0::2] - ref.real).max() <= 2e-6 * scale
    assert np.abs(bins[1::2] - ref.imag).max() <= 2e-6 * scale
    da, db = a.get(), b.get_db()
    strong = db >= db.max() - 60
    assert np.abs(da - db)[strong].max() <= 0.02


def test_receiver_chain_two_readings(oracle):
    """The whole Receiver chain (radio.cxx:62-90) from the restated parts, against the oracle's
    wro_receiver_run: channel IQ and AM audio bit for bit over three blocks."""
    rng = np.random.default_rng(7)
    fs, chan_rate, audio_rate = 2_000_000, 250_000, 50_000
    table = R.sin_table() if False else oracle.sin_table()       # one table for both: the libm question is tested above
    for mode in ("AM", "USB"):
        rx = oracle.Receiver(fs, -123_456, 160_000, chan_rate, getattr(oracle, mode), 8_000, audio_rate)
        step, phase, prev = R.phase_step(-123_456, fs), 0, (0.0, 0.0)
        f1 = R.Fir(2, fs // chan_rate, oracle.lowpass_design(160_000, fs))
        f2 = R.Fir(1, chan_rate // audio_rate, oracle.lowpass_design(8_000, chan_rate))
        for _ in range(3):
            iq = (0.3 * rng.standard_normal(2 * 8000)).astype(np.float32)
            wa, wc, wd = rx.run(iq)
            mixed, phase = R.mix(table, phase, step, iq)
            c = f1.process(mixed)
            d, prev = R.demod(mode, prev, c)
            a = f2.process(d)
            assert np.array_equal(c.view(np.uint32), wc.view(np.uint32))
            assert np.array_equal(d.view(np.uint32), wd.view(np.uint32))
            assert np.array_equal(a.view(np.uint32), wa.view(np.uint32))
