"""-m gpu: the streaming launch (include/webradio_amd.h wr_tuner_set_streaming; csrc/wr_stream_kernel.inc).

One persistent launch takes a tuner's device blocks through a doorbell instead of a kernel launch per block.  What
the reference fixes is the semantics -- a block's output leaves within its own run() (dsp/dspblock.cxx:169-212), the
state a block leaves behind is the next block's (downconverter.cxx:103, lowpass.cxx:138-142,
demodulator.cxx:110-113) -- so the tests hold the streamed path to the SAME BITS as one launch per block, which
tests/test_gpu_tuner.py holds to the oracle and the live reference: every block's audio, the channel IQ of the last
block, the per-channel state behind the stream, and whatever follows the stream."""
import time

import numpy as np
import pytest

from webradio_amd import capi, synth
from webradio_amd.device import Tuner

pytestmark = pytest.mark.gpu

FS, N = 2_000_000, 40_000                     # D1 = 400, D2 = 5: BASELINE config 2's decimations in miniature


def _ifs(nch):
    return [(-(nch // 2) + c) * 6250 + 99 for c in range(nch)]


def _tuner(dev, nch, modes=(capi.WR_FM, capi.WR_USB, capi.WR_AM, capi.WR_LSB), max_frames=N):
    t = Tuner(dev, FS, nch, max_frames, capi.WR_NCO_ROTATE)
    chans = [t.add_receiver(f, 128_000, 5_000, modes[c % len(modes)], 160, 1_000) for c, f in enumerate(_ifs(nch))]
    return t, chans


def _stream_dev(nblk, nch, n=N, seed=0):
    import torch
    iq = synth.fm_stream(nblk * n, FS, _ifs(nch)[::5], amp=0.1, fm_base=30.0, beta=2.0)
    x = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    return x


def _drain(t, count):
    out = []
    for b in range(count):
        audio, seq = t.ring_acquire()
        out.append((seq, audio.copy()))
        t.ring_release()
    return out


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("nch", [8, 70, 200])
def test_streamed_blocks_give_the_same_bits(dev, nch):
    """nblk consecutive blocks through one launch per block and through ONE streaming launch: the audio of every
    block (by the ring), the last block's channel IQ, every channel's state behind the stream, and two further
    blocks submitted the ordinary way after a retune."""
    nblk = 7
    x = _stream_dev(nblk + 2, nch)

    def run(stream):
        t, chans = _tuner(dev, nch)
        t.audio_ring(nblk + 2)
        t.streaming(stream)
        for b in range(nblk):
            t.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        if stream:
            live, launches, blocks = t.stream_info()
            assert live and launches == 1 and blocks == nblk
        t.flush()
        assert t.stream_info()[0] is False
        got = _drain(t, nblk)
        # the tuner's DEVICE audio array behind the stream is the LAST block's (two blocks' post stages can run
        # side by side in the drain; they store into two arrays by turns) -- by wr_tuner_fetch_audio_all and per channel
        last_dev = t.fetch_audio_all()
        assert np.array_equal(_bits(last_dev[:nch]), _bits(got[-1][1][:nch]))
        assert np.array_equal(_bits(t.fetch(chans[-1], capi.WR_STAGE_AUDIO, N)), _bits(got[-1][1][t.slot(chans[-1])]))
        iq = np.stack([t.fetch(c, capi.WR_STAGE_CHAN_IQ, 2 * N) for c in chans[::3]])
        state = [t.state(c) for c in chans]
        t.streaming(False)
        t.set_if(chans[1], _ifs(nch)[1] + 777)
        for b in range(nblk, nblk + 2):
            t.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        t.flush()
        got += _drain(t, 2)
        t.destroy()
        return got, iq, state

    one, iq1, st1 = run(False)
    many, iq2, st2 = run(True)
    assert [s for s, _ in one] == [s for s, _ in many] == list(range(nblk + 2))
    for (_, a), (_, b) in zip(one, many):
        assert a.shape == b.shape
        assert np.array_equal(_bits(a[:nch]), _bits(b[:nch]))
    assert float(np.abs(one[-1][1]).max()) > 0.0
    assert np.array_equal(_bits(iq1), _bits(iq2))
    for (p1, v1), (p2, v2) in zip(st1, st2):
        assert p1 == p2 and np.array_equal(_bits(np.asarray(v1, np.float32)), _bits(np.asarray(v2, np.float32)))


def test_whatever_touches_the_tuner_closes_the_launch(dev):
    """A retune between two blocks, a fetch in the middle, a block of another size, a block out of host memory: each
    closes the launch at ITS block boundary and the next eligible block opens another.  Same bits as the ordinary
    path, call for call."""
    nch, nblk = 70, 12
    x = _stream_dev(nblk, nch)
    host = x.cpu().numpy()

    def run(stream):
        t, chans = _tuner(dev, nch, modes=(capi.WR_USB, capi.WR_FM), max_frames=N)
        t.streaming(stream)
        out = []
        pos = 0

        def sub(frames, from_host=False):
            nonlocal pos
            if from_host:
                t.submit_host(host[2 * pos: 2 * (pos + frames)])
            else:
                t.submit_device(x[2 * pos: 2 * (pos + frames)], frames)
            pos += frames

        sub(N); sub(N); sub(N)
        t.set_if(chans[3], _ifs(nch)[3] + 1234)                 # staged: applies from the fourth block on
        sub(N); sub(N)
        out.append(t.fetch_audio_all().copy())                    # audio of the fifth block
        sub(N // 2); out.append(t.fetch_audio_all().copy())       # another size (50 channel-rate frames: fewer than a streaming
                                                                  # launch takes): the ordinary path
        sub(N // 2); sub(N // 2)
        out.append(t.fetch_audio_all().copy())
        sub(N, from_host=True)
        out.append(t.fetch_audio_all().copy())
        sub(N); sub(N)
        t.set_mode(chans[0], capi.WR_AM)
        sub(N)
        out.append(t.fetch_audio_all().copy())
        info = t.stream_info()
        t.destroy()
        return out, info

    a, _ = run(False)
    b, info = run(True)
    assert info[1] >= 4 and info[2] >= 8                          # launches opened, blocks they took
    for u, v in zip(a, b):
        assert u.shape == v.shape and np.array_equal(_bits(u), _bits(v))


def test_two_tuners_on_one_device_both_asked_to_stream(dev):
    """One streaming launch per device at a time (its workgroups must all be resident, and it holds the device's stream):
    with two tuners on one wr_dev submitting in turn, each submit closes the other tuner's launch at its block boundary
    and opens its own -- a launch per block again, no worse -- and a tuner left alone streams as usual.  Both tuners'
    audio: the bits of the ordinary path."""
    nch, nblk = 70, 6
    x = _stream_dev(2 * nblk, nch)

    def run(stream):
        ta, ca = _tuner(dev, nch)
        tb, cb = _tuner(dev, nch, modes=(capi.WR_AM, capi.WR_FM))
        for t in (ta, tb):
            t.streaming(stream)
            t.audio_ring(nblk + 3)
        for b in range(nblk):                                    # in turn: the other tuner's launch is open at every submit
            ta.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
            tb.submit_device(x[2 * N * (nblk + b): 2 * N * (nblk + b + 1)], N)
        ta.flush()
        tb.flush()
        oa, ob = _drain(ta, nblk), _drain(tb, nblk)
        ia, ib = ta.stream_info(), tb.stream_info()
        for b in range(3):                                        # b alone: one launch for the three
            tb.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        tb.flush()
        ob += _drain(tb, 3)
        ib2 = tb.stream_info()
        ta.destroy()
        tb.destroy()
        return oa, ob, ia, ib, ib2

    pa, pb, _, _, _ = run(False)
    sa, sb, ia, ib, ib2 = run(True)
    assert ia[1] >= nblk - 1 and ib[1] >= nblk - 1, (ia, ib)      # a launch per block each while they took turns
    assert ib2[1] == ib[1] + 1 and ib2[2] == ib[2] + 3, (ib, ib2)  # ... and ONE for the three blocks b submitted alone
    for want, got in ((pa, sa), (pb, sb)):
        assert [q for q, _ in want] == [q for q, _ in got]
        for (_, u), (_, v) in zip(want, got):
            assert u.shape == v.shape and np.array_equal(_bits(u), _bits(v))


def test_a_second_context_on_the_same_gpu_goes_the_ordinary_way(dev):
    """... and one streaming launch per GPU and process: a tuner of ANOTHER wr_dev on the same GPU that asks for streaming
    while the first context's launch is open is not given a launch of its own (both could not be resident: they would
    wait for each other until their deadlines) -- its blocks go the ordinary way, their kernels wait for room on the
    GPU until the open launch has ended, and the bits are the ordinary path's."""
    from webradio_amd.device import Device
    nch, nblk = 70, 3
    x = _stream_dev(nblk + 1, nch)
    want_a, want_b = None, None
    for stream in (False, True):
        devb = Device(0)
        ta, _ = _tuner(dev, nch)
        tb, _ = _tuner(devb, nch, modes=(capi.WR_AM, capi.WR_FM))
        for t in (ta, tb):
            t.streaming(stream)
        for b in range(nblk):
            ta.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        tb.submit_device(x[2 * N * nblk: 2 * N * (nblk + 1)], N)   # a's launch is open: enqueued the ordinary way
        got_a = ta.fetch_audio_all().copy()                        # closes a's launch; b's kernels find room
        got_b = tb.fetch_audio_all().copy()
        ia, ib = ta.stream_info(), tb.stream_info()
        ta.destroy()
        tb.destroy()
        devb.close()
        if not stream:
            want_a, want_b = got_a, got_b
        else:
            assert ia[2] >= nblk - 1 and ib[1] == 0 and ib[2] == 0, (ia, ib)
            assert np.array_equal(_bits(got_a), _bits(want_a)) and np.array_equal(_bits(got_b), _bits(want_b))


def test_a_blocks_audio_arrives_without_a_flush(dev):
    """dspblock.cxx:169-212: a block's output leaves within its own run().  The ring entry of a streamed block
    becomes ready when the launch's post stage has finished THAT block -- nothing has to follow it, nobody has to
    flush -- and the launch stays open for the next block."""
    nch, nblk = 70, 4
    x = _stream_dev(nblk, nch)
    want = []
    t, _ = _tuner(dev, nch)
    for b in range(nblk):
        t.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        want.append(t.fetch_audio_all().copy())
    t.destroy()
    t, _ = _tuner(dev, nch)
    t.audio_ring(2)
    t.streaming(True)
    for b in range(nblk):
        t.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        audio, seq = t.ring_acquire()                             # waits for the block's audio, not for the stream
        assert seq == b and np.array_equal(_bits(audio), _bits(want[b]))
        t.ring_release()
        assert t.stream_info() == (True, 1, b + 1)               # still the same launch
    t.destroy()                                                   # closes it


def test_an_idle_launch_ends_by_itself_and_the_host_knows(dev):
    """Nobody rings, nobody closes: after WR_STREAM_IDLE_MS the launch ends on its own (a caller that waits for
    the stream behind the library's back waits that long, not forever), and the next block -- the host does not ring
    a launch it has left alone for a tenth of that time -- opens a new one.  Same bits."""
    import torch
    nch, nblk = 8, 3
    x = _stream_dev(nblk, nch)
    want = []
    t, _ = _tuner(dev, nch)
    for b in range(nblk):
        t.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        want.append(t.fetch_audio_all().copy())
    t.destroy()
    t, _ = _tuner(dev, nch)
    t.audio_ring(nblk)
    t.streaming(True)
    t.submit_device(x[0: 2 * N], N)
    t0 = time.time()
    torch.cuda.synchronize()                                      # behind the library's back
    waited = time.time() - t0
    assert 0.2 < waited < 5.0, waited
    t.submit_device(x[2 * N: 4 * N], N)                           # stale: a new launch
    assert t.stream_info()[1] == 2
    time.sleep(0.25)                                              # stale again, the launch still open
    t.submit_device(x[4 * N: 6 * N], N)
    assert t.stream_info()[1] == 3
    t.flush()
    for b, (seq, audio) in enumerate(_drain(t, nblk)):
        assert seq == b and np.array_equal(_bits(audio), _bits(want[b]))
    t.destroy()


def test_u8_blocks_stream_too(dev):
    """the RTL-SDR byte format (io/rtlsdrtuner.cxx:106) converted in the launch's load stage, as k_tuner_ddc does"""
    import torch
    nch, nblk = 70, 5
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, size=2 * N * nblk, dtype=np.uint8)
    x = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()

    def run(stream):
        t, _ = _tuner(dev, nch)
        t.audio_ring(nblk)
        t.streaming(stream)
        for b in range(nblk):
            t.submit_u8_device(x[2 * N * b: 2 * N * (b + 1)], N)
        t.flush()
        got = _drain(t, nblk)
        t.destroy()
        return got

    for (s1, a), (s2, b) in zip(run(False), run(True)):
        assert s1 == s2 and np.array_equal(_bits(a), _bits(b))


@pytest.mark.parametrize("on_time", [False, True], ids=["ahead", "on-time"])
def test_byte_blocks_out_of_page_locked_host_memory_stream(dev, page_locked, on_time):
    """r06: what an RTL-SDR delivers (io/rtlsdrtuner.cxx:86-117: bytes in host memory, a ring of buffers) goes through the
    streaming launch too when asked (wr_tuner_set_streaming(tuner, 2)): wr_tuner_submit_u8(..., WR_HOST) out of page-locked
    memory -- the bytes cross PCIe as a DMA copy on the upload stream, the doorbell is rung by a stream memory operation behind them, the launch reads the windows at agent scope (its L2
    may hold what the buffer held four blocks ago).  Nine blocks through four rotating host buffers: the same bits as the
    same bytes resident on the device with a launch per block -- submitted all ahead of the GPU, or each block's audio taken
    before the next is handed over (what an on-time run() does); pageable memory goes the ordinary way."""
    import torch
    nch, nblk = 70, 9
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, size=2 * N * nblk, dtype=np.uint8)
    x = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()
    t, _ = _tuner(dev, nch)
    t.audio_ring(nblk)
    for b in range(nblk):
        t.submit_u8_device(x[2 * N * b: 2 * N * (b + 1)], N)
    t.flush()
    want = _drain(t, nblk)
    t.destroy()

    bufs = [page_locked(2 * N) for _ in range(4)]
    t, chans = _tuner(dev, nch)
    t.audio_ring(nblk)
    t.streaming(True)
    t.submit_u8_host(bufs[0])                               # (level 1: host blocks do not stream)
    assert t.last_staging() != 3 and t.stream_host_blocks() == 0
    t.destroy()
    t, chans = _tuner(dev, nch)
    t.audio_ring(nblk)
    t.streaming(2)                                          # wr_tuner_set_streaming(tuner, 2): byte blocks out of host memory too
    got = []
    for b in range(nblk):
        h = bufs[b % 4]
        if b >= 4:
            dev.lib.wr_dev_wait_uploads(dev.h)              # (the buffer's last copy has run: as a source's run() does)
        h[:] = raw[2 * N * b: 2 * N * (b + 1)]
        t.submit_u8_host(h)
        assert t.last_staging() == 3 and t.stream_info()[0], (b, t.last_staging(), t.stream_info())
        if on_time:
            got += _drain(t, 1)
            assert t.stream_info()[0]                       # (taking a block's audio needs no flush: WrStreamCtl::done)
    live, launches, blocks = t.stream_info()
    assert launches == 1 and blocks == nblk and t.stream_host_blocks() == nblk
    t.flush()
    if not on_time:
        got = _drain(t, nblk)
    st = t.state(chans[3])
    t.destroy()
    assert [s for s, _ in got] == [s for s, _ in want] == list(range(nblk))
    for (_, a), (_, b) in zip(want, got):
        assert np.array_equal(_bits(a[:nch]), _bits(b[:nch]))
    assert float(np.abs(want[-1][1]).max()) > 0.0 and st is not None
    # pageable memory: not streamed (the runtime would stage the copy synchronously), the same bits all the same
    t, _ = _tuner(dev, nch)
    t.audio_ring(2)
    t.streaming(2)
    t.submit_u8_host(raw[: 2 * N].copy())
    assert t.last_staging() != 3 and t.stream_host_blocks() == 0
    t.flush()
    a0 = _drain(t, 1)[0][1]
    t.destroy()
    assert np.array_equal(_bits(a0[:nch]), _bits(want[0][1][:nch]))


@pytest.mark.parametrize("paced,mixed", [(False, False), (True, False), (False, True)], ids=["host-ahead", "host-paced", "host-ahead-all-modes"])
def test_streaming_at_c2_size(dev, paced, mixed):
    """bench.py's configuration: 256 receivers, 4 M-frame blocks off 100 Msps.  Ten resident blocks through a tuner
    that launches each on its own and through ONE streaming launch: every audio sample of every receiver is the
    same bits, and so is what a further block gives after the stream.
    r06: with the host AHEAD (all blocks rung at once) the launch cuts a block's post stage into long runs of tiles, with a
    host-PACED stream (a block rung when the one before is long done) into short ones (wr_tuner_stream_long_blocks) --
    the same bits either way.  `all-modes`: receiver c demodulates AM / FM / USB / LSB by c mod 4 -- the long runs with every
    demodulator (the seeded fuzz streams are too small to be cut into long runs)."""
    import time
    import torch
    c2 = synth.C2
    fs, n = c2["input_rate"], c2["block_frames"]
    ifs = synth.c2_ifs()
    nblk = 10
    x = synth.fm_stream_torch(n * nblk, fs, ifs[::4], "cuda")
    torch.cuda.synchronize()

    def run(stream):
        t = Tuner(dev, fs, 256, n, capi.WR_NCO_ROTATE)
        for c, f in enumerate(ifs):
            t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], (capi.WR_AM, capi.WR_FM, capi.WR_USB, capi.WR_LSB)[c % 4] if mixed else capi.WR_FM,
                           c2["audio_passband"], c2["audio_rate"])
        if mixed:                                               # af_gain and squelch ride in the post stage's audio write (f-4)
            for c in range(0, 256, 8):
                t.set_af_gain(c, -6.0 + 0.25 * c)
            for c in range(4, 256, 16):
                t.set_squelch(c, -60.0 + 0.1 * c)
        t.audio_ring(nblk)
        t.streaming(stream)
        out = []
        for rep in range(2):        # (twice: the first pass page-locks the ring's slots as it goes, which paces it whatever the host does)
            for b in range(nblk - 1):
                t.submit_device(x[2 * n * b: 2 * n * (b + 1)], n)
                if paced and stream:
                    time.sleep(0.002)                           # (a block takes the launch 30 us: it idles till the next ring)
            t.flush()
            out += [a for _, a in _drain(t, nblk - 1)]
        #  the device audio array after the close holds the LAST streamed block's audio, all 256 rows
        assert np.array_equal(_bits(t.fetch_audio_all()), _bits(out[-1]))
        t.streaming(False)
        t.submit_device(x[2 * n * (nblk - 1): 2 * n * nblk], n)
        out.append(t.fetch_audio_all().copy())
        info = t.stream_info() + (t.stream_long_blocks(),)
        t.destroy()
        return np.concatenate(out, axis=1), info

    one, _ = run(False)
    many, info = run(True)
    assert info[1] == 2 and info[2] == 2 * (nblk - 1)
    # nine blocks rung at once: all but the last two (the drain) find two further blocks rung; paced, none does
    assert (info[3] == 0) if paced else (nblk - 3 <= info[3] <= 2 * (nblk - 3)), info
    assert one.shape == many.shape == (256, (2 * nblk - 1) * n // 400 // 5)
    assert np.array_equal(_bits(one), _bits(many))
    assert float(np.abs(one[::4]).max()) > 0.0                  # the carrier channels carry audio (every 4th: AM in `all-modes`)
    assert not mixed or float(np.abs(one[1::4]).max()) > 0.0


RATES = [(2_000_000, 250_000, 50_000), (2_400_000, 240_000, 48_000), (1_920_000, 240_000, 24_000), (2_048_000, 256_000, 32_000)]


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("WR_FUZZ_SEEDS", "10"))))
def test_random_streams_with_streaming(dev, seed):
    """The streaming launch under a random stream of calls (after tests/test_gpu_fuzz.py's blocks-per-launch fuzz): the same
    sequence of submits (device blocks of the usual size, of another size, out of host memory), setters (IF, mode, a channel
    filter of its own, af_gain), flushes and state reads goes through a tuner that launches every block on its own and one
    with wr_tuner_set_streaming on -- which streams what it may and closes the launch for everything else.  Everything either
    hands out through the audio ring, laid end to end, is the same bits; so are the NCO phases read on the way."""
    import torch
    rng = np.random.default_rng(9100 + seed)
    fs, crate, arate = RATES[seed % len(RATES)]
    d1, d2 = fs // crate, crate // arate
    q = d1 * d2                                             # frames per audio frame
    nchan = int(rng.choice([3, 64, 130, 200]))
    whole = q * int(rng.integers((64 + d2 - 1) // d2, (64 + d2 - 1) // d2 + 6))     # >= 64 channel-rate frames: streams
    total = whole * 36
    ifs = [int(v) for v in rng.integers(-fs // 2 + 1, fs // 2, nchan)]
    iq = synth.fm_stream(total, fs, ifs[:3], amp=0.15, fm_base=fs / 70_000.0, beta=2.0, seed=seed)
    x = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    modes = [int(m) for m in rng.integers(0, 4, nchan)]

    ops, pos = [], 0
    while pos + 2 * whole < total and len(ops) < 60:
        r = rng.random()
        if r < 0.70:
            ops.append(("dev", pos, whole)); pos += whole
        elif r < 0.76:                                      # another size (whole audio frames): a launch of its own
            n = whole + q * int(rng.integers(1, 3))
            ops.append(("dev", pos, n)); pos += n
        elif r < 0.80:                                      # ragged: not a whole number of audio frames -- never streamed
            n = whole + int(rng.integers(1, q))
            ops.append(("dev", pos, n)); pos += n
        elif r < 0.84:
            ops.append(("host", pos, whole)); pos += whole
        elif r < 0.90:
            ops.append(("set_if", int(rng.integers(0, nchan)), int(rng.integers(-fs // 2 + 1, fs // 2))))
        elif r < 0.93:
            ops.append(("set_mode", int(rng.integers(0, nchan)), int(rng.integers(0, 4))))
        elif r < 0.95 and seed % 3 == 2:                    # (a third of the seeds: a filter of its own ends the streaming for good)
            ops.append(("set_filter", int(rng.integers(0, nchan)), int(rng.choice([fs // 40, fs // 8]))))
        elif r < 0.96:
            ops.append(("gain", int(rng.integers(0, nchan)), float(rng.choice([-6.0, 0.0, 3.5]))))
        elif r < 0.98:
            ops.append(("flush",))
        else:
            ops.append(("state", int(rng.integers(0, nchan))))

    def play(stream):
        t = Tuner(dev, fs, nchan, whole + 3 * q, capi.WR_NCO_ROTATE)
        chans = [t.add_receiver(f, fs // 16, crate, m, crate // 8, arate) for f, m in zip(ifs, modes)]
        t.audio_ring(128)
        t.streaming(stream)
        rows, states = [], []

        def drain():
            while t.ring_stats()[0]:
                a, _ = t.ring_acquire()
                rows.append(a.copy())
                t.ring_release()
        for op in ops:
            if op[0] == "dev":
                t.submit_device(x[2 * op[1]: 2 * (op[1] + op[2])], op[2])
            elif op[0] == "host":
                t.submit_host(iq[2 * op[1]: 2 * (op[1] + op[2])])
            elif op[0] == "set_if":
                t.set_if(chans[op[1]], op[2])
            elif op[0] == "set_mode":
                t.set_mode(chans[op[1]], op[2])
            elif op[0] == "set_filter":
                t.set_filter(chans[op[1]], 0, op[2], crate)
            elif op[0] == "gain":
                t.set_af_gain(chans[op[1]], op[2])
            elif op[0] == "flush":
                t.flush()
            elif op[0] == "state":
                states.append(t.state(chans[op[1]])[0])
            drain()
        t.flush()
        dev.sync()
        drain()
        slots = [t.slot(c) for c in chans]
        info = t.stream_info()
        t.destroy()
        return np.concatenate([r[slots] for r in rows], axis=1) if rows else np.zeros((nchan, 0), np.float32), states, info

    one, st1, _ = play(False)
    many, st2, info = play(True)
    assert one.shape == many.shape and one.shape[1] > 0
    assert st1 == st2
    assert np.array_equal(_bits(one), _bits(many)), (seed, info)
    if seed % 3 != 2:
        assert info[2] >= 3, info                           # blocks that did go through streaming launches


# ---- the streaming launch against the ORACLE and the REFERENCE'S OWN vectors, directly: k_tuner_stream
#      is the kernel bench.py times; the tests above hold it to another HIP path's bits, these to the checker itself ----------

def test_c2_full_size_stream_against_the_oracle(dev, oracle):
    """BASELINE config 2 at full size, as bench.py runs it: 256 receivers, three resident 4 M-frame blocks through ONE
    k_tuner_stream launch, every block's audio out of the pinned ring.
      - three receivers against the ORACLE (radio.cxx:68-83 over downconverter.cxx:91-114, lowpass.cxx:131-162,
        demodulator.cxx:77-115) on the stream's first 200 000 frames: channel IQ within IQ_ATOL, audio within AUDIO_ATOL;
      - ALL 256 receivers against the bit-exact mode (WR_NCO_EXACT, which test_gpu_tuner.py holds to the oracle bit for
        bit): the last block's channel IQ within IQ_ATOL (2.5e-7, what DESIGN.md quotes), every block's FM audio on the
        carrier channels within AUDIO_ATOL, and the last block's on EVERY channel -- the 192 noise-only ones too --
        within what the channel's own IQ difference allows (tests/fm_bound.py);
      - the phases behind the stream equal the closed form (downconverter.cxx:103)."""
    import torch
    import fm_bound
    IQ_ATOL, AUDIO_ATOL = 1e-6, 1e-5
    c2 = synth.C2
    fs, n = c2["input_rate"], c2["block_frames"]
    ifs = synth.c2_ifs()
    nblk = 3
    k1, k2 = n // 400, n // 2000
    x = synth.fm_stream_torch(n * nblk, fs, ifs[::4], "cuda")
    torch.cuda.synchronize()

    def tuner(mode):
        t = Tuner(dev, fs, 256, n, mode)
        chans = [t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"], c2["audio_rate"])
                 for f in ifs]
        return t, chans

    # the stream
    t, chans = tuner(capi.WR_NCO_ROTATE)
    t.audio_ring(nblk)
    t.streaming(True)
    for b in range(nblk):
        t.submit_device(x[2 * n * b: 2 * n * (b + 1)], n)
    live, launches, blocks = t.stream_info()
    assert live and launches == 1 and blocks == nblk, (live, launches, blocks)
    t.flush()
    s_audio = np.concatenate([a for _, a in _drain(t, nblk)], axis=1)            # [slot][frames of the whole stream]
    assert s_audio.shape == (256, nblk * k2)
    slots = [t.slot(ch) for ch in chans]
    s_iq_last = [t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * k1) for ch in chans]
    for ch, f in zip(chans[::37], ifs[::37]):
        ph, _ = t.state(ch)
        assert ph == (nblk * n * oracle.phase_step(f, fs)) % (1 << 31)
    t.destroy()
    # a one-block stream: block 0's channel IQ can be fetched (a stream keeps the last block's)
    t, chans = tuner(capi.WR_NCO_ROTATE)
    t.streaming(True)
    t.submit_device(x[: 2 * n], n)
    assert t.stream_info() == (True, 1, 1)
    s_iq_first = {c: t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * k1) for c in (0, 128, 252)}
    t.destroy()

    # the bit-exact mode, a launch per block
    t, chans = tuner(capi.WR_NCO_EXACT)
    e_audio = []
    for b in range(nblk):
        t.submit_device(x[2 * n * b: 2 * n * (b + 1)], n)
        e_audio.append(t.fetch_audio_all().copy())
    e_audio = np.concatenate(e_audio, axis=1)
    e_iq_last = [t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * k1) for ch in chans]
    e_slots = [t.slot(ch) for ch in chans]
    t.destroy()
    worst = max(float(np.abs(a - b).max()) for a, b in zip(e_iq_last, s_iq_last))
    assert worst <= 2.5e-7 <= IQ_ATOL, worst
    for c in range(0, 256, 4):                                                   # carrier channels: every block's audio
        assert np.abs(e_audio[e_slots[c]] - s_audio[slots[c]]).max() <= AUDIO_ATOL, c
        assert float(np.abs(s_audio[slots[c]]).max()) > 1e-3
    # every channel, the last block: the audio filter's first 13 frames reach into the block before (63 demodulator rows),
    # whose channel IQ a stream does not keep -- the bound is applied from frame 13 on
    taps2 = oracle.lowpass_design(c2["audio_passband"], c2["chan_rate"])
    skip = 13
    worst_ratio = 0.0
    for c in range(256):
        want_a, got_a = e_audio[e_slots[c]][-k2:], s_audio[slots[c]][-k2:]
        bound = fm_bound.audio_bound(e_iq_last[c], s_iq_last[c], taps2, 5,
                                     want_demod=oracle.demod(oracle.FM, (0.0, 0.0), e_iq_last[c])[0])
        diff = np.abs(want_a.astype(np.float64) - got_a.astype(np.float64))
        ratio = float((diff[skip:] / bound[skip:k2]).max())
        worst_ratio = max(worst_ratio, ratio)
        assert ratio <= 1.0, (c, ratio)
    assert worst_ratio <= 1.0
    # the oracle itself on the stream's first frames
    m = 200_000
    xh = x[: 2 * m].cpu().numpy()
    for c in (0, 128, 252):
        rx = oracle.Receiver(fs, ifs[c], c2["chan_passband"], c2["chan_rate"], oracle.FM, c2["audio_passband"], c2["audio_rate"])
        wa, wc, _ = rx.run(xh)
        assert wc.size == 2 * (m // 400) and wa.size == m // 2000
        assert np.abs(s_iq_first[c][: wc.size] - wc).max() <= IQ_ATOL
        assert np.abs(s_audio[slots[c]][: wa.size] - wa).max() <= AUDIO_ATOL
        assert float(np.abs(wa).max()) > 1e-3


def test_c1_capture_streams_against_the_reference_vectors(dev, oracle):
    """BASELINE config 1 through the streaming launch: the recorded RTL-SDR capture's bytes (io/rtlsdrtuner.cxx:106's format)
    resident in device memory, its four blocks through ONE k_tuner_stream launch (D1 = D2 = 8), against what the
    REAL reference chain gave for the same bytes (tests/golden/reference_c1.npz: radio.cxx:68-83 on the reference's own
    DownConverter / LowPass / Demodulator): audio of every block within AUDIO_ATOL, the last block's channel IQ within
    IQ_ATOL -- and the float form of the same capture gives the same bits as the bytes do."""
    import os
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_c1.npz"))
    c1 = synth.C1
    n = int(g["block_frames"])
    nblk = g["u8"].size // (2 * n)
    assert nblk == 4
    u8 = np.ascontiguousarray(g["u8"][: 2 * n * nblk])
    outs = []
    for as_bytes in (True, False):
        x = torch.from_numpy(u8 if as_bytes else oracle.u8_to_float(u8)).cuda()
        torch.cuda.synchronize()
        t = Tuner(dev, c1["input_rate"], 1, n, capi.WR_NCO_ROTATE)
        ch = t.add_receiver(c1["if_hz"], c1["chan_passband"], c1["chan_rate"], capi.WR_FM, c1["audio_passband"], c1["audio_rate"])
        t.audio_ring(nblk)
        t.streaming(True)
        for b in range(nblk):
            (t.submit_u8_device if as_bytes else t.submit_device)(x[2 * n * b: 2 * n * (b + 1)], n)
        assert t.stream_info() == (True, 1, nblk)
        t.flush()
        audio = np.concatenate([a[t.slot(ch)] for _, a in _drain(t, nblk)])
        chan = t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n)
        t.destroy()
        assert audio.shape == g["audio"].shape
        assert np.abs(audio - g["audio"]).max() <= 1e-5
        assert chan.size and np.abs(chan - g["chan_iq"][-chan.size:]).max() <= 1e-6
        assert float(np.abs(g["audio"]).max()) > 1e-3
        outs.append((audio, chan))
    assert np.array_equal(_bits(outs[0][0]), _bits(outs[1][0])) and np.array_equal(_bits(outs[0][1]), _bits(outs[1][1]))


@pytest.mark.parametrize("nfft,hop", [(512, 0), (4096, 2048)])
def test_a_front_ends_spectrum_sink_beside_the_stream(dev, nfft, hop):
    """radio.cxx:120-133: every FrontEnd wires a SpectrumSink to its tuner, and DspBlock::run hands it every block
    (dspblock.cxx:207-209).  While the tuner's streaming launch is open a pushed block's newest frame is kept, not
    transformed (that would close the launch every block); a poll -- waterfallhandler.cxx:56-61, 5 Hz -- transforms it and
    costs ONE closed launch.  The same blocks through the ordinary way (a launch per block, a transform per push): the
    same audio bits, the same spectrum bits at every poll."""
    from webradio_amd.device import Spectrum
    nch, nblk, poll_every = 70, 12, 5
    x = _stream_dev(nblk, nch)

    def run(stream):
        t, chans = _tuner(dev, nch)
        sp = Spectrum(dev, nfft, hop)
        t.audio_ring(nblk)
        t.streaming(stream)
        polls, launches_at_poll = [], []
        for b in range(nblk):
            blk = x[2 * N * b: 2 * N * (b + 1)]
            t.submit_device(blk, N)
            sp.push_device(blk, N)
            if stream:
                assert t.stream_info()[0], "a pushed block closed the launch"
            if (b + 1) % poll_every == 0:
                polls.append((sp.get_db().copy(), sp.get_bins().copy(), sp.frames_done()))
                launches_at_poll.append(t.stream_info()[1])
        t.flush()
        audio = _drain(t, nblk)
        info, lazy = t.stream_info(), sp.lazy_info()
        sp.destroy()
        t.destroy()
        return audio, polls, info, lazy, launches_at_poll

    a1, p1, _, lazy1, _ = run(False)
    a2, p2, info2, lazy2, lp = run(True)
    assert lazy1 == (0, 0)
    assert lazy2 == (nblk, nblk // poll_every)              # every push kept, one transform per poll
    assert info2[1] == 1 + nblk // poll_every and lp == [1, 2]      # a poll closes the launch; the next block opens the next
    assert info2[2] == nblk                                 # ... and every block went through a streaming launch
    for (s1, u), (s2, v) in zip(a1, a2):
        assert s1 == s2 and np.array_equal(_bits(u[:nch]), _bits(v[:nch]))
    assert len(p1) == len(p2) == nblk // poll_every
    for (d1, b1, f1), (d2, b2, f2) in zip(p1, p2):
        assert f1 == f2 and np.array_equal(_bits(b1), _bits(b2)) and np.array_equal(_bits(d1), _bits(d2))
