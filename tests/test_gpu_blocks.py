"""-m gpu: the one-kernel-per-reference-block tier, through the C ABI, against the oracle.
Integer phase and the unfused float arithmetic make these BIT-EXACT; only the FM
detector depends on a device libm function (atan2f) and gets a stated tolerance."""
import numpy as np
import pytest

from webradio_amd import capi

pytestmark = pytest.mark.gpu

# FM: out = atan2f(..)/pi/2; device atan2f is within a few ulp of glibc's -> <= 4 ulp of
# the result range [-0.5, 0.5], i.e. 2.4e-7 absolute
FM_ATOL = 2.4e-7


@pytest.mark.parametrize("if_hz,rate,n", [(100_000, 2_400_000, 102_400), (-39_843_750, 100_000_000, 50_001),
                                          (0, 2_048_000, 1), (1_199_999, 2_400_000, 4096)])
def test_mix_bit_exact(dev, oracle, if_hz, rate, n):
    rng = np.random.default_rng(n)
    iq = rng.uniform(-1, 1, 2 * n).astype(np.float32)
    step = oracle.phase_step(if_hz, rate)
    t = oracle.sin_table()
    phase_g, phase_o = 0x12345678 & 0x7FFFFFFF, 0x12345678 & 0x7FFFFFFF
    for blk in range(3):                               # phase carries across blocks
        want, phase_o = oracle.mix(t, phase_o, step, iq)
        got, phase_g = dev.mix(iq, phase_g, step)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert phase_g == phase_o


def test_mix_empty_block(dev):
    got, ph = dev.mix(np.zeros(0, np.float32), 5, 7)
    assert got.size == 0 and ph == 5


@pytest.mark.parametrize("channels,decim,blocks", [(2, 10, [20480] * 3), (1, 5, [2048] * 2),
                                                   (2, 400, [40000] * 2), (2, 8, [8] * 12),
                                                   (1, 1, [10] * 9), (2, 8, [21] * 5),
                                                   (3, 4, [402] * 3), (2, 100, [63] * 4)])
def test_fir_bit_exact_streaming(dev, oracle, channels, decim, blocks):
    # constant block size per stream, as every reference source produces (blockSize is
    # frozen while running, dspblock.cxx:242-249); ragged, odd and shorter-than-history
    # sizes included.  (A size CHANGE mid-stream is quirk Q7: see the next test.)
    rng = np.random.default_rng(channels * 1000 + decim)
    coeff = oracle.lowpass_design(200_000, 2_048_000)
    fir = oracle.Fir(channels, decim, coeff)
    hist = dev.malloc(63 * channels * 4)
    try:
        for frames in blocks:                          # ragged and shorter-than-history blocks
            x = rng.uniform(-1, 1, frames * channels).astype(np.float32)
            want = fir.process(x)
            got = dev.fir_decimate(x, channels, decim, coeff, hist)
            assert got.size == want.size
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    finally:
        dev.free(hist)


@pytest.mark.parametrize("length", [2, 16, 128, 512, 1024])
def test_fir_other_lengths_bit_exact(dev, oracle, length):
    """LowPass::_firLength as a run-time value (the reference's FIXME at
    lowpass.cxx:38-39): wr_fir_decimate_n against the oracle's LowPass::process with the same
    length, streaming, blocks shorter than the history included."""
    rng = np.random.default_rng(length)
    coeff = oracle.lowpass_design(150_000, 2_048_000, length)
    for channels, decim in ((1, 5), (2, 8)):
        fir = oracle.Fir(channels, decim, coeff)
        hist = dev.malloc(max(4, (length - 1) * channels * 4))
        try:
            for _ in range(4):
                frames = decim * 37
                x = rng.uniform(-1, 1, frames * channels).astype(np.float32)
                want = fir.process(x)
                got = dev.fir_decimate(x, channels, decim, coeff, hist)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        finally:
            dev.free(hist)


def test_fir_block_size_change_keeps_true_history(dev, oracle):
    """Deliberate deviation: when the block size changes mid-stream
    the reference resizes its buffer BEFORE saving the history (lowpass.cxx:138-141) and so
    filters garbage; the GPU path keeps the true last 63 frames.  With sizes that are
    multiples of the decimation the result then equals one uninterrupted block."""
    rng = np.random.default_rng(8)
    coeff = oracle.lowpass_design(200_000, 2_048_000)
    x = rng.uniform(-1, 1, 2 * 1760).astype(np.float32)
    want = oracle.Fir(2, 8, coeff).process(x)
    hist = dev.malloc(63 * 2 * 4)
    got, pos = [], 0
    for frames in (8, 16, 40, 800, 24, 872):
        got.append(dev.fir_decimate(x[2 * pos: 2 * (pos + frames)], 2, 8, coeff, hist))
        pos += frames
    dev.free(hist)
    assert np.array_equal(np.concatenate(got).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("mode", [capi.WR_AM, capi.WR_USB, capi.WR_LSB])
def test_demod_bit_exact_modes(dev, oracle, mode):
    rng = np.random.default_rng(mode)
    prev_g, prev_o = (0.0, 0.0), (0.0, 0.0)
    for blk in range(3):
        iq = rng.standard_normal(2 * 5000).astype(np.float32)
        want, prev_o = oracle.demod(mode, prev_o, iq)
        got, prev_g = dev.demod(mode, iq, prev_g)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert tuple(prev_g) == prev_o


def test_demod_fm(dev, oracle):
    rng = np.random.default_rng(77)
    prev_g, prev_o = (0.0, 0.0), (0.0, 0.0)
    for blk in range(3):
        iq = rng.standard_normal(2 * 20000).astype(np.float32)
        if blk == 0:
            iq[:8] = [0.0, 0.0, -0.0, 1.0, 0.0, -1.0, -1.0, -0.0]   # signed-zero atan2f inputs
        want, prev_o = oracle.demod(oracle.FM, prev_o, iq)
        got, prev_g = dev.demod(capi.WR_FM, iq, prev_g)
        assert np.abs(got - want).max() <= FM_ATOL
        assert tuple(prev_g) == prev_o
    # the demod golden vectors of the REAL reference
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demod_reference.npz"))
    got, _ = dev.demod(capi.WR_FM, g["iq"], (0.0, 0.0))
    assert np.abs(got - g["out_FM"]).max() <= FM_ATOL
    for name, mode in (("AM", capi.WR_AM), ("USB", capi.WR_USB), ("LSB", capi.WR_LSB)):
        got, _ = dev.demod(mode, g["iq"], (0.0, 0.0))
        assert np.array_equal(got.view(np.uint32), g["out_" + name].view(np.uint32))


def test_demod_bad_mode(dev):
    import ctypes as C
    prev = np.zeros(2, np.float32)
    rc = dev.lib.wr_demod(dev.h, 7, None, 0, capi.ptr(prev), None)
    assert rc == capi.WR_ERR_ARG


def test_u8_ingest(dev, oracle):
    import ctypes as C
    b = np.arange(256, dtype=np.uint8).repeat(3)
    din = dev.upload(b)
    dout = dev.malloc(b.size * 4)
    capi.check(dev.lib.wr_u8_to_f32(dev.h, C.c_void_p(din), C.c_void_p(dout), b.size))
    got = dev.download(dout, b.size)
    assert np.array_equal(got, oracle.u8_to_float(b))
    dev.free(din)
    dev.free(dout)


def test_u8_ingest_from_page_locked_host_memory(dev, oracle, page_locked):
    """wr_u8_to_f32_from_host: the bytes cross PCIe on the library's own stream (DMA + conversion kernel) and the
    device's stream waits for them.  Blocks alternate between two device buffers and two host buffers the way the host
    runtime stages a RawU8Block source (gpubatch.cxx), one size that is not a multiple of 16, the same buffer twice in a
    row; wr_dev_wait_uploads_but(1) lets the host refill one buffer while the other is in flight.  Every block is the
    reference's (u8 - 128) / 128 bit for bit (rtlsdrtuner.cxx:106)."""
    import ctypes as C
    rng = np.random.default_rng(11)
    n = 1_000_003
    hosts = [page_locked(n, np.uint8), page_locked(n, np.uint8)]          # (registered; tests/conftest.py says where they live)
    douts = [dev.malloc(n * 4), dev.malloc(n * 4)]
    try:
        for b in range(7):
            which = b & 1 if b != 4 else 1                       # block 4 reuses the buffers of block 3
            capi.check(dev.lib.wr_dev_wait_uploads_but(dev.h, 0 if b == 4 else 1))
            hosts[which][:] = rng.integers(0, 256, n, dtype=np.uint8)
            count = n if b != 2 else 4096 + 48
            capi.check(dev.lib.wr_u8_to_f32_from_host(dev.h, capi.ptr(hosts[which]), C.c_void_p(douts[which]), count))
            got = dev.download(douts[which], count)             # on the device's stream: waits for the conversion
            assert np.array_equal(got, oracle.u8_to_float(hosts[which][:count])), b
        capi.check(dev.lib.wr_dev_wait_uploads(dev.h))
        assert dev.lib.wr_dev_wait_uploads_but(dev.h, 4) == capi.WR_ERR_ARG
        other = np.zeros(64, np.uint8)                           # not page-locked: refused, nothing enqueued
        assert dev.lib.wr_u8_to_f32_from_host(dev.h, capi.ptr(other), C.c_void_p(douts[0]), 64) == capi.WR_ERR_ARG
    finally:
        dev.sync()
        for d in douts:
            dev.free(d)


def test_upload_ahead_alternating_buffers(dev, page_locked):
    """wr_dev_upload_ahead: the copy runs on the library's upload stream and the device's stream waits for it.  Blocks
    alternate between two device buffers (what the host runtime does with a float source's block vector), each is read
    back on the device's stream before the next call, the same buffer twice in a row works too."""
    import ctypes as C
    rng = np.random.default_rng(12)
    n = 300_001
    hosts = [page_locked(n, np.float32), page_locked(n, np.float32)]
    douts = [dev.malloc(n * 4), dev.malloc(n * 4)]
    try:
        for b in range(8):
            which = b & 1 if b != 5 else 0
            capi.check(dev.lib.wr_dev_wait_uploads(dev.h))       # a float source refills the vector it swapped out at once
            hosts[which][:] = rng.standard_normal(n).astype(np.float32)
            capi.check(dev.lib.wr_dev_upload_ahead(dev.h, C.c_void_p(douts[which]), capi.ptr(hosts[which]), hosts[which].nbytes))
            got = dev.download(douts[which], n)
            assert np.array_equal(got.view(np.uint32), hosts[which].view(np.uint32)), b
        assert dev.lib.wr_dev_upload_ahead(dev.h, None, capi.ptr(hosts[0]), 16) == capi.WR_ERR_ARG
    finally:
        dev.sync()
        capi.check(dev.lib.wr_dev_wait_uploads(dev.h))
        for d in douts:
            dev.free(d)


def test_reference_blocks_chain_equals_oracle_receiver(dev, oracle):
    """mix -> fir -> demod -> fir with the per-block kernels == the oracle's Receiver."""
    from webradio_amd import synth
    c1 = synth.C1
    rx = oracle.Receiver(c1["input_rate"], c1["if_hz"], c1["chan_passband"], c1["chan_rate"], oracle.AM,
                         c1["audio_passband"], c1["audio_rate"])
    step = oracle.phase_step(c1["if_hz"], c1["input_rate"])
    tc = oracle.lowpass_design(c1["chan_passband"], c1["input_rate"])
    ta = oracle.lowpass_design(c1["audio_passband"], c1["chan_rate"])
    h1, h2 = dev.malloc(63 * 2 * 4), dev.malloc(63 * 4)
    phase, prev = 0, (0.0, 0.0)
    for b in range(3):
        iq = synth.fm_stream(16384, c1["input_rate"], [c1["if_hz"]], start_frame=b * 16384, amp=0.5)
        want_audio, want_chan, want_dem = rx.run(iq)
        mixed, phase = dev.mix(iq, phase, step)
        chan = dev.fir_decimate(mixed, 2, rx.d1, tc, h1)
        dem, prev = dev.demod(capi.WR_AM, chan, prev)
        audio = dev.fir_decimate(dem, 1, rx.d2, ta, h2)
        assert np.array_equal(chan, want_chan)
        assert np.array_equal(dem, want_dem)
        assert np.array_equal(audio, want_audio)
    dev.free(h1)
    dev.free(h2)
