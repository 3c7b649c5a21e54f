"""-m gpu: the REAL reference blocks run live on this box -- dsp/downconverter.cxx, dsp/lowpass.cxx,
dsp/demodulator.cxx, io/spectrumsink.cxx behind oracle/ref_chain.cxx (oracle/_ref/libwr_ref_chain.so), their FFTW
calls served by the image's hipFFTW, which needs the GPU -- and are compared with (1) the oracle, (2) the HIP path
through the C ABI, (3) the vectors committed under tests/golden/ (the same reference, run when they were made).

oracle/_ref/ travels to the GPU box prebuilt (the sources it is built from exist only in the build container).
Where it is missing the tests SKIP and say so: the committed vectors still hold the oracle (CPU test
tests/test_oracle_reference_chain.py) and, below, the HIP path."""
import os

import numpy as np
import pytest

import refcases
from webradio_amd import capi
from webradio_amd.device import Tuner, Spectrum

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_chain.npz")


@pytest.fixture(scope="module")
def live(oracle, tmp_path_factory):
    """What the reference produces on THIS box, now: tests/golden/make_reference_chain_golden.py in a process of its
    own (hipFFTW brings the system's HIP runtime, the rest of the suite runs on torch's: one process cannot hold both)."""
    if not os.path.exists(oracle.REF_CHAIN_LIB):
        pytest.skip("oracle/_ref/libwr_ref_chain.so not built (needs /root/reference at build time)")
    import sys
    import _proc
    out = str(tmp_path_factory.mktemp("refchain") / "live.npz")
    _proc.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_reference_chain_golden.py"), out], timeout=240,
              env=dict(os.environ, WEBRADIO_QUIET="1"))
    return np.load(out)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN) if os.path.exists(GOLDEN) else None


@pytest.mark.parametrize("pb,rate", refcases.LOWPASS)
def test_live_lowpass_taps(live, oracle, gold, pb, rate):
    """LowPass::init/recalculate (lowpass.cxx:81-116,164-197) observed as the impulse response of the running
    block, against the oracle's design AND the C ABI's (wr_lowpass_design is what the product uploads)."""
    import ctypes as C
    want = live["taps_%d_%d" % (pb, rate)]
    assert np.abs(oracle.lowpass_design(pb, rate) - want).max() <= refcases.TAPS_TOL
    ours = np.empty(64, np.float32)
    maxbin = C.c_uint()
    assert capi.load().wr_lowpass_design(pb, rate, capi.ptr(ours), C.byref(maxbin)) == 0
    assert np.abs(ours - want).max() <= refcases.TAPS_TOL
    assert (maxbin.value == 0) == (not want.any())
    if gold is not None:
        assert np.abs(gold["taps_%d_%d" % (pb, rate)] - want).max() <= 1e-9      # the same library, the same answer


@pytest.mark.parametrize("name", sorted(refcases.MIXES))
def test_live_mixer(live, oracle, dev, name):
    """DownConverter::process (downconverter.cxx:91-114): the oracle and wr_mix (k_mix, EXACT arithmetic) give the
    reference's bits."""
    import ctypes as C
    import torch
    c = refcases.MIXES[name]
    iq = refcases.mix_input(c)
    want = live["mix_" + name]
    step = oracle.phase_step(c["if_hz"], c["fs"])
    got, _ = oracle.mix(oracle.sin_table(), 0, step, iq)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    x = torch.from_numpy(iq).cuda()
    y = torch.empty_like(x)
    ph = C.c_uint(0)
    assert dev.lib.wr_mix(dev.h, capi.ptr(x), capi.ptr(y), iq.size // 2, C.byref(ph), step) == 0
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint32), want.view(np.uint32))


def _hip_chain(dev, c, iq, nco):
    t = Tuner(dev, c["fs"], 1, c["block"], nco)
    ch = t.add_receiver(c["if_hz"], c["cpb"], c["crate"], c["mode"], c["apb"], c["arate"])
    t.keep_stages(capi.WR_STAGE_DEMOD)
    n = c["block"]
    audio, chan, dem = [], [], []
    for b in range(c["blocks"]):
        t.submit_host(iq[2 * n * b: 2 * n * (b + 1)])
        chan.append(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n))
        dem.append(t.fetch(ch, capi.WR_STAGE_DEMOD, n))
        audio.append(t.fetch(ch, capi.WR_STAGE_AUDIO, n))
    t.destroy()
    return np.concatenate(audio), np.concatenate(chan), np.concatenate(dem)


def _stream_cut(c):
    """The block size a streaming launch takes (wr_tuner_set_streaming: at least 64 channel-rate frames, whole audio
    frames per block): the case's own where that qualifies, otherwise consecutive blocks merged -- a frame's bits do not
    depend on where the stream is cut into blocks (lowpass.cxx:138-142 carries the history over)."""
    d1, d2 = c["fs"] // c["crate"], c["crate"] // c["arate"]
    for m in range(1, c["blocks"] + 1):
        n = c["block"] * m
        if c["blocks"] % m == 0 and n >= 64 * d1 and n % (d1 * d2) == 0:
            return n
    raise AssertionError("no cut of this case is eligible for a streaming launch")


def _hip_chain_stream(dev, c, iq):
    """The same Receiver through k_tuner_stream -- the kernel bench.py times: the input resident in device memory,
    Tuner.streaming(True), one submit_device per block (the first opens the launch, the others ring its doorbell), the
    audio of EVERY block out of the pinned ring, the last block's channel IQ.  Asserts that the launch was live and took
    every block (no silent fall-back to a launch per block)."""
    import torch
    n = _stream_cut(c)
    nblk = c["block"] * c["blocks"] // n
    x = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    torch.cuda.synchronize()
    t = Tuner(dev, c["fs"], 1, n, capi.WR_NCO_ROTATE)
    ch = t.add_receiver(c["if_hz"], c["cpb"], c["crate"], c["mode"], c["apb"], c["arate"])
    t.audio_ring(nblk)
    t.streaming(True)
    for b in range(nblk):
        t.submit_device(x[2 * n * b: 2 * n * (b + 1)], n)
    live, launches, blocks = t.stream_info()
    assert live and launches == 1 and blocks == nblk, (live, launches, blocks)
    t.flush()
    assert t.stream_info()[0] is False
    slot, audio = t.slot(ch), []
    for b in range(nblk):
        a, seq = t.ring_acquire()
        assert seq == b
        audio.append(a[slot].copy())
        t.ring_release()
    chan_last = t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n)
    t.destroy()
    del x
    return np.concatenate(audio), chan_last


@pytest.mark.parametrize("name", sorted(refcases.CHAINS))
def test_live_receiver_chain(live, oracle, dev, gold, name):
    """The Receiver chain of radio.cxx:68-83 on the reference's own blocks: channel IQ, demodulator output and
    audio of the oracle and of the HIP path (EXACT and ROTATE NCO) within the stated tolerances."""
    c = refcases.CHAINS[name]
    iq = refcases.chain_input(c)
    assert np.array_equal(refcases.sha(iq), live["sha_chain_" + name])
    w_audio, w_chan, w_dem = (live["chain_%s_%s" % (name, k)] for k in ("audio", "chan", "demod"))
    assert w_audio.size == c["blocks"] * (c["block"] // (c["fs"] // c["crate"]) // (c["crate"] // c["arate"]))
    rx = oracle.Receiver(c["fs"], c["if_hz"], c["cpb"], c["crate"], c["mode"], c["apb"], c["arate"])
    n = c["block"]
    parts = [rx.run(iq[2 * n * b: 2 * n * (b + 1)]) for b in range(c["blocks"])]
    o_audio, o_chan, o_dem = (np.concatenate([p[i] for p in parts]) for i in range(3))
    for got, want, tol in ((o_chan, w_chan, refcases.CHAN_TOL), (o_dem, w_dem, refcases.AUDIO_TOL), (o_audio, w_audio, refcases.AUDIO_TOL)):
        assert got.shape == want.shape and np.abs(got - want).max() <= tol
    for nco in (capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE):
        g_audio, g_chan, g_dem = _hip_chain(dev, c, iq, nco)
        assert g_chan.shape == w_chan.shape and np.abs(g_chan - w_chan).max() <= refcases.CHAN_TOL, nco
        assert g_audio.shape == w_audio.shape and np.abs(g_audio - w_audio).max() <= refcases.AUDIO_TOL, nco
        assert np.abs(g_dem - w_dem).max() <= refcases.AUDIO_TOL, nco
    # ... and through the streaming launch (k_tuner_stream, what bench.py times), against the reference itself
    s_audio, s_chan = _hip_chain_stream(dev, c, iq)
    assert s_audio.shape == w_audio.shape and np.abs(s_audio - w_audio).max() <= refcases.AUDIO_TOL
    assert s_chan.size and np.abs(s_chan - w_chan[-s_chan.size:]).max() <= refcases.CHAN_TOL
    if gold is not None:
        assert np.abs(gold["chain_%s_audio" % name] - w_audio).max() <= 1e-7


@pytest.mark.parametrize("name", sorted(refcases.SPECTRA))
def test_live_spectrum(live, oracle, dev, gold, name):
    """SpectrumSink (spectrumsink.cxx:60-142) on the reference's own code: dB with fft-shift after the last block,
    against the oracle and wr_spectrum_* (k_fft*)."""
    c = refcases.SPECTRA[name]
    iq = refcases.spectrum_input(c)
    want = live["spec_" + name]
    strong = want >= want.max() - refcases.DB_MASK
    o = oracle.Spectrum(c["n"])
    s = Spectrum(dev, c["n"])
    n = c["block"]
    for b in range(c["blocks"]):
        o.process(iq[2 * n * b: 2 * n * (b + 1)])
        s.push_host(iq[2 * n * b: 2 * n * (b + 1)])
    assert np.abs(o.get() - want)[strong].max() <= refcases.DB_TOL
    got = s.get_db()
    s.destroy()
    assert np.abs(got - want)[strong].max() <= refcases.DB_TOL
    assert int(np.argmax(got)) == int(np.argmax(want))
    if gold is not None:
        assert np.abs(gold["spec_" + name] - want)[strong].max() <= 1e-4


@pytest.fixture(scope="module")
def live_full(oracle, tmp_path_factory):
    if not os.path.exists(oracle.REF_CHAIN_LIB):
        pytest.skip("oracle/_ref/libwr_ref_chain.so not built (needs /root/reference at build time)")
    import sys
    import _proc
    out = str(tmp_path_factory.mktemp("refchain") / "full.npz")
    _proc.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_reference_chain_golden.py"), "--full", out],
              timeout=280, env=dict(os.environ, WEBRADIO_QUIET="1"))
    return np.load(out)


@pytest.mark.parametrize("stream", [False, True], ids=["launch-per-block", "streaming-launch"])
@pytest.mark.parametrize("name", sorted(refcases.FULL))
def test_full_size_tuner_against_the_live_reference(live_full, dev, name, stream):
    """BASELINE config 2 at its full size (256 receivers, one 4 000 000-frame block off 100 Msps) and config 5's parameters
    (1 Gsps, D1 = 4000, two blocks): the whole tuner through the HIP path in its default mode, a few of its receivers through
    the reference's OWN DownConverter -> LowPass -> Demodulator -> LowPass on the same input -- channel IQ within 1e-6, FM
    audio within 1e-5 (the probed receivers hold carriers; three of the C2 probes demodulate AM, USB and LSB).
    [streaming-launch]: the blocks resident in device memory through ONE k_tuner_stream launch -- the kernel and the
    configuration bench.py's headline times -- every block's audio out of the pinned ring, the last block's channel IQ."""
    c = refcases.FULL[name]
    iq = refcases.full_input(c)
    assert np.array_equal(refcases.sha(iq), live_full["sha_full_" + name])
    ifs = refcases.full_ifs(c)
    n = c["block"]
    t = Tuner(dev, c["fs"], c["channels"], n, capi.WR_NCO_ROTATE)
    chans = [t.add_receiver(f, c["cpb"], c["crate"], refcases.full_mode(c, i), c["apb"], c["arate"]) for i, f in enumerate(ifs)]
    got = {ch: ([], []) for ch in c["probe"]}
    if stream:
        import torch
        x = torch.from_numpy(iq).cuda()
        torch.cuda.synchronize()
        t.audio_ring(c["blocks"])
        t.streaming(True)
        for b in range(c["blocks"]):
            t.submit_device(x[2 * n * b: 2 * n * (b + 1)], n)
        live, launches, blocks = t.stream_info()
        assert live and launches == 1 and blocks == c["blocks"], (live, launches, blocks)
        t.flush()
        for b in range(c["blocks"]):
            a, seq = t.ring_acquire()
            assert seq == b
            for ch in c["probe"]:
                got[ch][1].append(a[t.slot(chans[ch])].copy())
            t.ring_release()
        for ch in c["probe"]:
            got[ch][0].append(t.fetch(chans[ch], capi.WR_STAGE_CHAN_IQ, 2 * n))      # the LAST block's
    else:
        for b in range(c["blocks"]):
            t.submit_host(iq[2 * n * b: 2 * n * (b + 1)])
            for ch in c["probe"]:
                got[ch][0].append(t.fetch(chans[ch], capi.WR_STAGE_CHAN_IQ, 2 * n))
                got[ch][1].append(t.fetch(chans[ch], capi.WR_STAGE_AUDIO, n))
    t.destroy()
    for ch in c["probe"]:
        w_chan, w_audio = live_full["full_%s_%d_chan" % (name, ch)], live_full["full_%s_%d_audio" % (name, ch)]
        g_chan, g_audio = np.concatenate(got[ch][0]), np.concatenate(got[ch][1])
        if stream:
            w_chan = w_chan[-g_chan.size:]
        assert g_chan.shape == w_chan.shape and g_audio.shape == w_audio.shape and w_audio.size > 0
        assert np.abs(g_chan - w_chan).max() <= refcases.CHAN_TOL, (ch, float(np.abs(g_chan - w_chan).max()))
        assert np.abs(g_audio - w_audio).max() <= refcases.AUDIO_TOL, (ch, float(np.abs(g_audio - w_audio).max()))
        assert np.abs(w_audio).max() > 1e-3


def test_full_size_waterfall_against_the_live_reference(live_full, dev):
    """BASELINE config 3 at its full size: the 121-row waterfall of a 4 000 000-frame block (65536-point frames every 32768
    frames, wr_spectrum_batch_db: k_fft64k_pass1/2), four of its rows against the reference's own SpectrumSink fed the row's
    65536 frames (the reference itself does not overlap: SURVEY section 0)."""
    import torch
    c3 = refcases.C3_FULL
    c = refcases.FULL[c3["case"]]
    iq = refcases.full_input(c)
    nrows = (c["block"] - c3["n"]) // c3["hop"] + 1
    assert nrows == 121
    x = torch.from_numpy(iq).cuda()
    out = torch.empty(nrows * c3["n"], dtype=torch.float32, device="cuda")
    s = Spectrum(dev, c3["n"], c3["hop"])
    s.batch_db(x, nrows, out)
    torch.cuda.synchronize()
    rows = out.cpu().numpy().reshape(nrows, c3["n"])
    s.destroy()
    for r in c3["rows"]:
        want = live_full["c3_row_%d" % r]
        strong = want >= want.max() - refcases.DB_MASK
        assert strong.sum() >= 3
        assert np.abs(rows[r] - want)[strong].max() <= refcases.DB_TOL, r
        assert int(np.argmax(rows[r])) == int(np.argmax(want))


@pytest.mark.parametrize("name", sorted(refcases.CHAINS))
def test_hip_path_against_committed_reference_vectors(dev, name):
    """No oracle/_ref needed: the HIP path against what the reference produced when the vectors were made."""
    assert os.path.exists(GOLDEN), "tests/golden/reference_chain.npz is missing"
    g = np.load(GOLDEN)
    c = refcases.CHAINS[name]
    iq = refcases.chain_input(c)
    assert np.array_equal(refcases.sha(iq), g["sha_chain_" + name])
    audio, chan, dem = _hip_chain(dev, c, iq, capi.WR_NCO_ROTATE)
    assert np.abs(chan - g["chain_%s_chan" % name]).max() <= refcases.CHAN_TOL
    assert np.abs(audio - g["chain_%s_audio" % name]).max() <= refcases.AUDIO_TOL
    # the streaming launch (k_tuner_stream) against the same committed vectors
    s_audio, s_chan = _hip_chain_stream(dev, c, iq)
    assert s_audio.shape == g["chain_%s_audio" % name].shape
    assert np.abs(s_audio - g["chain_%s_audio" % name]).max() <= refcases.AUDIO_TOL
    assert s_chan.size and np.abs(s_chan - g["chain_%s_chan" % name][-s_chan.size:]).max() <= refcases.CHAN_TOL
