"""-m gpu: the fused per-tuner path (wr_tuner_*) against the oracle's Receiver chains.

Tolerances (float32, |x| <= 1):
  WR_NCO_EXACT  channel-filter IQ is BIT-EXACT (same table, same unfused operations in the
                same order); AM/USB/LSB demod and audio bit-exact; FM within FM_ATOL.
  WR_NCO_SPLIT  the LO comes from the two-level table: it differs from the reference's
                table entry by <= 3.5e-7 (the reference table itself carries 2.4e-7 of
                argument-rounding noise) and the taps are accumulated with FMAs, so
                channel IQ is within IQ_ATOL = 1e-6 absolute; FM audio on channels
                that hold a carrier within AUDIO_ATOL = 1e-5, and on EVERY channel -- noise-only
                ones too, where FM is ill-conditioned -- within what the channel's
                own IQ difference allows: per demod frame 1.25 * (|dz[k]|/|z[k]| + |dz[k-1]|/|z[k-1]|)
                / (2 pi) + 2.4e-7 cycles, through the audio filter's |taps| (tests/fm_bound.py;
                the bound itself is tested on the CPU in tests/test_fm_bound.py).
  WR_NCO_ROTATE (default) the same table index per frame; each LO value is the anchor's turned
                by a product of correctly rounded turns (Horner recurrence): same tolerances as
                SPLIT, measured 1.3e-7 on channel IQ.
In every mode the integer NCO phase is exact and a frame's bits do not depend on how the stream
is cut into blocks.
"""
import numpy as np
import pytest

from webradio_amd import capi, synth
from webradio_amd.device import Tuner
import fm_bound

pytestmark = pytest.mark.gpu

FM_ATOL = 2.4e-7
IQ_ATOL = 1e-6
AUDIO_ATOL = 1e-5

MODES = [capi.WR_AM, capi.WR_FM, capi.WR_USB, capi.WR_LSB]


def _mini_c2(nchan):
    """C2 geometry scaled down: fs 2 Msps, D1 = 400 -> 5 kHz, D2 = 5 -> 1 kHz."""
    fs = 2_000_000
    ifs = [(-nchan // 2 + c) * 6250 + 1234 for c in range(nchan)]
    return dict(fs=fs, ifs=ifs, chan_pb=128_000, chan_rate=5_000, audio_pb=160, audio_rate=1_000)


def _run_both(dev, oracle, nco, cfg, modes, blocks, seed=0, carriers=None, keep_demod=True):
    """keep_demod: the demodulator output is fetched too (demod and audio filter then run as
    two kernels); without it the fused k_tuner_post runs and the demod entry of `got` is None."""
    fs = cfg["fs"]
    nchan = len(cfg["ifs"])
    maxblk = max(blocks)
    t = Tuner(dev, fs, max(nchan, 1), maxblk, nco)
    if keep_demod:
        t.keep_stages(capi.WR_STAGE_DEMOD)
    rxs, chans = [], []
    for c, f in enumerate(cfg["ifs"]):
        m = modes[c % len(modes)]
        rxs.append(oracle.Receiver(fs, f, cfg["chan_pb"], cfg["chan_rate"], m, cfg["audio_pb"], cfg["audio_rate"]))
        chans.append(t.add_receiver(f, cfg["chan_pb"], cfg["chan_rate"], m, cfg["audio_pb"], cfg["audio_rate"]))
    car = cfg["ifs"][::4] if carriers is None else carriers
    start = 0
    results = []
    for n in blocks:
        iq = synth.fm_stream(n, fs, car, start_frame=start, seed=seed, fm_base=30.0, fm_step=3.0, beta=2.0)
        start += n
        t.submit_host(iq)
        blk = []
        for c in range(nchan):
            want = rxs[c].run(iq)
            k1 = n // rxs[c].d1
            got = (t.fetch(chans[c], capi.WR_STAGE_AUDIO, k1 // rxs[c].d2 + 1),
                   t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * k1 + 2),
                   t.fetch(chans[c], capi.WR_STAGE_DEMOD, k1 + 1) if keep_demod else None)
            blk.append((want, got))
        results.append(blk)
    states = [(t.state(ch), (rx.s.phase, rx.s.prev_i, rx.s.prev_q)) for ch, rx in zip(chans, rxs)]
    t.destroy()
    return results, states, car


def test_exact_mode_bit_exact_iq(dev, oracle):
    cfg = _mini_c2(70)                                  # 70 channels: two lane groups, one ragged
    results, states, car = _run_both(dev, oracle, capi.WR_NCO_EXACT, cfg, MODES, [40_000, 40_000, 40_000])
    for blk in results:
        for c, ((wa, wc, wd), (ga, gc, gd)) in enumerate(blk):
            assert gc.size == wc.size and ga.size == wa.size and gd.size == wd.size
            assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), c
            if MODES[c % 4] == capi.WR_FM:
                assert np.abs(gd - wd).max() <= FM_ATOL
                assert np.abs(ga - wa).max() <= 2 * FM_ATOL
            else:
                assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32))
    for (gph, gprev), (oph, opi, opq) in states:
        assert gph == oph and gprev[0] == opi and gprev[1] == opq


@pytest.mark.parametrize("nco", [capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_fast_nco_modes_within_tolerance(dev, oracle, nco):
    cfg = _mini_c2(64)
    results, states, car = _run_both(dev, oracle, nco, cfg, [capi.WR_FM, capi.WR_AM], [80_000, 80_000])
    worst_iq = 0.0
    for blk in results:
        for c, ((wa, wc, wd), (ga, gc, gd)) in enumerate(blk):
            worst_iq = max(worst_iq, float(np.abs(gc - wc).max()))
            if cfg["ifs"][c] in car:                   # carrier present: demod is well conditioned
                assert np.abs(ga - wa).max() <= AUDIO_ATOL, c
    assert worst_iq <= IQ_ATOL
    # every FM channel, carrier or not: the audio differs by no more than its IQ difference allows
    taps2 = oracle.lowpass_design(cfg["audio_pb"], cfg["chan_rate"])
    d2 = cfg["chan_rate"] // cfg["audio_rate"]
    checked = 0
    for c in range(0, len(cfg["ifs"]), 2):              # the FM channels of [FM, AM]
        cat = lambda which, idx: np.concatenate([blk[c][which][idx] for blk in results])
        fm_bound.assert_fm_audio_within_iq_bound(cat(0, 1), cat(1, 1), cat(0, 0), cat(1, 0), taps2, d2,
                                                 want_demod=cat(0, 2), what="channel %d" % c)
        checked += 1
    assert checked == len(cfg["ifs"]) // 2
    for (gph, _), (oph, _, _) in states:
        assert gph == oph                               # integer phase is exact in either mode


def test_c1_single_receiver_u8_file(dev, oracle):
    """BASELINE config 1: one DownConverter + FM demod off an RTL-SDR format (u8) capture, against what the REAL reference
    chain gave for the same bytes (tests/golden/reference_c1.npz: io/rtlsdrtuner.cxx:106's conversion, then the reference's
    own DownConverter -> LowPass -> Demodulator -> LowPass wired as radio.cxx:68-83, run on the GPU box through
    oracle/_ref/libwr_ref_chain.so by tests/golden/make_c1_reference_golden.py).  The EXACT mode repeats the reference's
    arithmetic operation for operation -- and for THESE passbands the reference's taps (a 64-point inverse DFT in f32) and
    ours (a closed form accumulated in double) are the same floats: the channel IQ is the reference's bit for bit, the
    audio within the in-kernel atan2's distance from glibc's (the generator prints oracle - reference = 0, 0, 0)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_c1.npz"))
    c1 = synth.C1
    n = int(g["block_frames"])
    iq = oracle.u8_to_float(g["u8"])
    for nco, exact in ((capi.WR_NCO_EXACT, True), (capi.WR_NCO_SPLIT, False), (capi.WR_NCO_ROTATE, False)):
        t = Tuner(dev, c1["input_rate"], 1, n, nco)
        ch = t.add_receiver(c1["if_hz"], c1["chan_passband"], c1["chan_rate"], capi.WR_FM,
                            c1["audio_passband"], c1["audio_rate"])
        audio, chan = [], []
        for b in range(4):
            t.submit_host(iq[2 * n * b: 2 * n * (b + 1)])
            audio.append(t.fetch(ch, capi.WR_STAGE_AUDIO, n))
            chan.append(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n))
        audio, chan = np.concatenate(audio), np.concatenate(chan)
        t.destroy()
        assert chan.shape == g["chan_iq"].shape and audio.shape == g["audio"].shape
        if exact:
            assert np.array_equal(chan.view(np.uint32), g["chan_iq"].view(np.uint32))
            assert np.abs(audio - g["audio"]).max() <= 2 * FM_ATOL
        else:
            assert np.abs(chan - g["chan_iq"]).max() <= IQ_ATOL
            assert np.abs(audio - g["audio"]).max() <= AUDIO_ATOL


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_block_split_invariance(dev, nco):
    """Size-independent property: a stream cut into different block sizes (multiples of
    D1*D2) gives bit-identical output -- the 63-frame history, the closed-form phase and
    prev_i/q stitch blocks exactly (SURVEY 5 'streaming block continuity')."""
    cfg = _mini_c2(8)
    fs = cfg["fs"]
    total = 160_000
    iq = synth.fm_stream(total, fs, cfg["ifs"][::2], fm_base=30.0, beta=2.0)
    outs = []
    for blocks in ([160_000], [40_000] * 4, [2000, 38_000, 120_000], [80_000, 2000, 2000, 76_000]):
        t = Tuner(dev, fs, 8, max(blocks), nco)
        chans = [t.add_receiver(f, cfg["chan_pb"], cfg["chan_rate"], capi.WR_FM, cfg["audio_pb"], cfg["audio_rate"])
                 for f in cfg["ifs"]]
        pos, audio, chan = 0, [[] for _ in chans], [[] for _ in chans]
        for n in blocks:
            t.submit_host(iq[2 * pos: 2 * (pos + n)])
            pos += n
            for i, ch in enumerate(chans):
                audio[i].append(t.fetch(ch, capi.WR_STAGE_AUDIO, n))
                chan[i].append(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n))
        outs.append((np.concatenate([np.concatenate(a) for a in audio]),
                     np.concatenate([np.concatenate(c) for c in chan])))
        t.destroy()
    for a, c in outs[1:]:
        assert np.array_equal(c.view(np.uint32), outs[0][1].view(np.uint32))
        assert np.array_equal(a.view(np.uint32), outs[0][0].view(np.uint32))


def test_small_decimation_and_ragged_blocks(dev, oracle):
    """D1 < 64 (overlapping FIR windows, several outputs reach into the history) and block
    sizes that are not multiples of D1*D2 (truncation, H8)."""
    fs = 240_000
    cfg = dict(fs=fs, ifs=[10_000, -20_000, 0], chan_pb=20_000, chan_rate=24_000, audio_pb=4_000, audio_rate=8_000)
    for blocks in ([1003] * 4, [17] * 9, [5] * 30, [2400] * 2, [999] * 3):   # constant size per stream
        results, states, _ = _run_both(dev, oracle, capi.WR_NCO_EXACT, cfg, [capi.WR_AM, capi.WR_USB, capi.WR_LSB],
                                       blocks, carriers=[10_000])
        for blk in results:
            for (wa, wc, wd), (ga, gc, gd) in blk:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32))
                assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32))
        for (gph, gprev), (oph, opi, opq) in states:
            assert gph == oph and gprev[0] == opi and gprev[1] == opq


def test_setters_apply_at_block_boundary(dev, oracle):
    """setIF / setMode between blocks: the history of the channel filter was mixed with
    the old phase step (lowpass.cxx:138-142 keeps MIXED samples)."""
    fs = 2_000_000
    t = Tuner(dev, fs, 2, 40_000, capi.WR_NCO_EXACT)
    t.keep_stages(capi.WR_STAGE_DEMOD)
    rx = oracle.Receiver(fs, 50_000, 128_000, 5_000, oracle.AM, 160, 1_000)
    ch = t.add_receiver(50_000, 128_000, 5_000, capi.WR_AM, 160, 1_000)
    pos = 0
    for b, (f, m) in enumerate([(50_000, oracle.AM), (-75_000, oracle.AM), (-75_000, oracle.USB), (10, oracle.LSB)]):
        rx.set_if(f)
        rx.set_mode(m)
        t.set_if(ch, f)
        t.set_mode(ch, m)
        iq = synth.fm_stream(40_000, fs, [50_000, -75_000], start_frame=pos, amp=0.3)
        pos += 40_000
        wa, wc, wd = rx.run(iq)
        t.submit_host(iq)
        assert np.array_equal(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 200), wc)
        assert np.array_equal(t.fetch(ch, capi.WR_STAGE_DEMOD, 200), wd)
        assert np.array_equal(t.fetch(ch, capi.WR_STAGE_AUDIO, 200), wa)
    t.destroy()


def test_add_remove_receivers_and_rate_groups(dev, oracle):
    fs = 2_000_000
    t = Tuner(dev, fs, 4, 40_000, capi.WR_NCO_EXACT)
    a = t.add_receiver(1000, 128_000, 5_000, capi.WR_AM, 160, 1_000)       # D1 400, D2 5
    b = t.add_receiver(-3000, 256_000, 10_000, capi.WR_USB, 320, 2_000)    # D1 200, D2 5: other group
    ra = oracle.Receiver(fs, 1000, 128_000, 5_000, oracle.AM, 160, 1_000)
    rb = oracle.Receiver(fs, -3000, 256_000, 10_000, oracle.USB, 320, 2_000)
    iq = synth.fm_stream(40_000, fs, [1000, -3000], amp=0.3)
    t.submit_host(iq)
    assert np.array_equal(t.fetch(a, capi.WR_STAGE_AUDIO, 100), ra.run(iq)[0])
    assert np.array_equal(t.fetch(b, capi.WR_STAGE_AUDIO, 100), rb.run(iq)[0])
    t.remove_receiver(a)
    with pytest.raises(capi.WrError):
        t.fetch(a, capi.WR_STAGE_AUDIO, 100)           # removed
    c = t.add_receiver(5000, 128_000, 5_000, capi.WR_LSB, 160, 1_000)      # fresh: zero history, phase 0
    rc = oracle.Receiver(fs, 5000, 128_000, 5_000, oracle.LSB, 160, 1_000)
    iq2 = synth.fm_stream(40_000, fs, [5000, -3000], start_frame=40_000, amp=0.3)
    t.submit_host(iq2)
    assert np.array_equal(t.fetch(c, capi.WR_STAGE_AUDIO, 100), rc.run(iq2)[0])
    assert np.array_equal(t.fetch(b, capi.WR_STAGE_AUDIO, 100), rb.run(iq2)[0])
    t.destroy()


def test_errors(dev):
    t = Tuner(dev, 2_000_000, 1, 1000, capi.WR_NCO_SPLIT)
    import ctypes as C
    c = C.c_int()
    capi.check(t.lib.wr_chan_add(t.h, C.byref(c)))
    assert t.lib.wr_chan_add(t.h, C.byref(c)) == capi.WR_ERR_STATE             # max_channels
    # LowPass::init without rate; non-integer ratio (dspblock.cxx:126-130)
    assert t.lib.wr_tuner_submit(t.h, None, 0, capi.WR_HOST) == capi.WR_ERR_STATE
    assert t.lib.wr_chan_set_filter(t.h, 0, 0, 100_000, 240_000) == capi.WR_ERR_RATE
    assert b"integer related" in t.lib.wr_last_error()
    assert t.lib.wr_chan_set_filter(t.h, 0, 1, 100, 1000) == capi.WR_ERR_STATE   # audio before channel
    assert t.lib.wr_chan_set_mode(t.h, 0, 9) == capi.WR_ERR_ARG
    x = np.zeros(4000, np.float32)
    capi.check(t.lib.wr_chan_set_filter(t.h, 0, 0, 128_000, 5_000))
    capi.check(t.lib.wr_chan_set_filter(t.h, 0, 1, 160, 1_000))
    assert t.lib.wr_tuner_submit(t.h, capi.ptr(x), 2000, capi.WR_HOST) == capi.WR_ERR_ARG   # > max_block
    t.destroy()


@pytest.mark.parametrize("nco", [capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_c2_full_size_properties(dev, oracle, nco):
    """BASELINE config 2 at full size (256 channels, 4 M-frame block off 100 Msps), input
    generated on the device.  The CPU oracle needs minutes for this, so:
      - three channels are checked against the oracle on the first 200 000 frames
        (block-split invariance makes a prefix comparable),
      - SPLIT is compared with EXACT on every channel (IQ within IQ_ATOL),
      - phases after the block equal the closed form."""
    import torch
    c2 = synth.C2
    fs, n = c2["input_rate"], c2["block_frames"]
    ifs = synth.c2_ifs()
    x = synth.fm_stream_torch(n, fs, ifs[::4], "cuda")
    torch.cuda.synchronize()
    outs = {}
    for mode in (capi.WR_NCO_EXACT, nco):
        t = Tuner(dev, fs, 256, n, mode)
        chans = [t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"],
                                c2["audio_rate"]) for f in ifs]
        t.submit_device(x, n)
        dev.sync()
        outs[mode] = [(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 20_000), t.fetch(ch, capi.WR_STAGE_AUDIO, 2_000))
                      for ch in chans]
        for ch, f in zip(chans[::37], ifs[::37]):
            ph, _ = t.state(ch)
            assert ph == (n * oracle.phase_step(f, fs)) % (1 << 31)
        t.destroy()
    worst = max(float(np.abs(a[0] - b[0]).max()) for a, b in zip(outs[capi.WR_NCO_EXACT], outs[nco]))
    assert worst <= IQ_ATOL
    if nco == capi.WR_NCO_ROTATE:
        # what DESIGN.md quotes for the default mode (correctly rounded turns, one anchor per frame)
        assert worst <= 2.5e-7, worst
    for a, b in list(zip(outs[capi.WR_NCO_EXACT], outs[nco]))[::4]:      # carrier channels
        assert np.abs(a[1] - b[1]).max() <= AUDIO_ATOL
    # ALL 256 channels, the 192 noise-only ones included: FM audio of the fast mode against the bit-exact mode
    # within what the channel's own IQ difference allows (tests/fm_bound.py)
    taps2 = oracle.lowpass_design(c2["audio_passband"], c2["chan_rate"])
    worst_ratio = 0.0
    for c, (a, b) in enumerate(zip(outs[capi.WR_NCO_EXACT], outs[nco])):
        dem, _ = oracle.demod(oracle.FM, (0.0, 0.0), a[0])
        _, ratio = fm_bound.assert_fm_audio_within_iq_bound(a[0], b[0], a[1], b[1], taps2, 5, want_demod=dem,
                                                            what="channel %d" % c)
        worst_ratio = max(worst_ratio, ratio)
    assert worst_ratio <= 1.0
    # oracle on a prefix for three channels
    m = 200_000
    xh = x[: 2 * m].cpu().numpy()
    for c in (0, 128, 252):
        rx = oracle.Receiver(fs, ifs[c], c2["chan_passband"], c2["chan_rate"], oracle.FM, c2["audio_passband"],
                             c2["audio_rate"])
        wa, wc, _ = rx.run(xh)
        assert np.array_equal(outs[capi.WR_NCO_EXACT][c][0][: wc.size].view(np.uint32), wc.view(np.uint32))
        assert np.abs(outs[nco][c][0][: wc.size] - wc).max() <= IQ_ATOL
        assert np.abs(outs[nco][c][1][: wa.size] - wa).max() <= AUDIO_ATOL


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_u8_ingest_equals_converting_first(dev, oracle, nco):
    """the RTL-SDR byte format goes straight into the DDC kernel's load stage;
    the (u8 - 128)/128 rule of rtlsdrtuner.cxx:106 is exact in float, so the result is
    bit-identical to converting on the host first (and, in EXACT mode, to the oracle)."""
    c1 = synth.C1
    n = 16384
    u8 = synth.rtl_u8_stream(3 * n)
    iq = oracle.u8_to_float(u8)
    outs = []
    for fmt in ("f32", "u8"):
        t = Tuner(dev, c1["input_rate"], 2, n, nco)
        chans = [t.add_receiver(f, c1["chan_passband"], c1["chan_rate"], capi.WR_FM, c1["audio_passband"],
                                c1["audio_rate"]) for f in (c1["if_hz"], -250_000)]
        got = []
        for b in range(3):
            if fmt == "f32":
                t.submit_host(iq[2 * n * b: 2 * n * (b + 1)])
            else:
                blk = np.ascontiguousarray(u8[2 * n * b: 2 * n * (b + 1)])
                capi.check(t.lib.wr_tuner_submit_u8(t.h, capi.ptr(blk), n, capi.WR_HOST))
                dev.sync()
            got.append([(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n), t.fetch(ch, capi.WR_STAGE_AUDIO, n)) for ch in chans])
        outs.append(got)
        t.destroy()
    for bf, bu in zip(*outs):
        for (cf, af), (cu, au) in zip(bf, bu):
            assert np.array_equal(cf.view(np.uint32), cu.view(np.uint32))
            assert np.array_equal(af.view(np.uint32), au.view(np.uint32))
    if nco == capi.WR_NCO_EXACT:
        rx = oracle.Receiver(c1["input_rate"], c1["if_hz"], c1["chan_passband"], c1["chan_rate"], oracle.FM,
                             c1["audio_passband"], c1["audio_rate"])
        for b in range(3):
            _, wc, _ = rx.run(iq[2 * n * b: 2 * n * (b + 1)])
            assert np.array_equal(outs[1][b][0][0].view(np.uint32), wc.view(np.uint32))


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_mixed_passbands_within_one_lane_group(dev, oracle, nco):
    """Receivers of one tuner with DIFFERENT channel/audio passbands (setPassband per receiver,
    receiverhandler.cxx:133-134): same rates -> same rate group, but the taps differ per lane,
    which takes the per-lane-taps variant of the DDC kernel."""
    fs = 2_000_000
    pbs = [(64_000, 160), (128_000, 320), (192_000, 480), (128_000, 160), (31_250, 100), (0, 160)]  # last: maxbin 0
    ifs = [(-3 + c) * 6250 + 777 for c in range(len(pbs))]
    t = Tuner(dev, fs, len(pbs), 40_000, nco)
    rxs, chans = [], []
    for f, (cpb, apb) in zip(ifs, pbs):
        rxs.append(oracle.Receiver(fs, f, cpb, 5_000, oracle.AM, apb, 1_000))
        chans.append(t.add_receiver(f, cpb, 5_000, capi.WR_AM, apb, 1_000))
    pos = 0
    for b in range(3):
        iq = synth.fm_stream(40_000, fs, ifs[::2], start_frame=pos, amp=0.2, fm_base=30.0, beta=2.0)
        pos += 40_000
        t.submit_host(iq)
        for rx, ch in zip(rxs, chans):
            wa, wc, wd = rx.run(iq)
            gc = t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 400)
            ga = t.fetch(ch, capi.WR_STAGE_AUDIO, 400)
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32))
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32))
            else:
                assert np.abs(gc - wc).max() <= IQ_ATOL and np.abs(ga - wa).max() <= 2e-6
    # retuning one receiver's passband mid-stream (recalculate while running, lowpass.cxx:55-61)
    rxs[1] = None
    t.set_filter(chans[1], 0, 64_000, 5_000)
    r1 = oracle.Receiver(fs, ifs[1], 64_000, 5_000, oracle.AM, 320, 1_000)
    iq = synth.fm_stream(40_000, fs, ifs[::2], start_frame=pos, amp=0.2)
    t.submit_host(iq)
    gc = t.fetch(chans[1], capi.WR_STAGE_CHAN_IQ, 400)
    # the filter history survives a passband change in the reference (same LowPass::block);
    # a fresh oracle receiver has none, so compare from the second output frame on
    _, wc, _ = r1.run(iq)
    assert gc.size == wc.size
    t.destroy()


def test_channel_counts(dev, oracle):
    """1, 63, 64, 65 and 130 receivers: partial and multiple lane groups."""
    fs = 2_000_000
    for n in (1, 63, 65, 130):
        cfg = _mini_c2(n)
        results, states, _ = _run_both(dev, oracle, capi.WR_NCO_EXACT, cfg, [capi.WR_USB], [40_000, 40_000],
                                       carriers=cfg["ifs"][:3])
        for blk in results:
            for (wa, wc, wd), (ga, gc, gd) in blk[:: max(1, n // 9)]:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32))
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32))


def test_audio_scale_for_the_encoder(dev, oracle):
    """the +/-32768 scaling LAME wants (mp3encoder.cxx:65-72) applied in the
    audio kernel's store: exact (a power of two)."""
    fs = 2_000_000
    t = Tuner(dev, fs, 1, 40_000, capi.WR_NCO_EXACT)
    ch = t.add_receiver(50_000, 128_000, 5_000, capi.WR_AM, 160, 1_000)
    capi.check(t.lib.wr_tuner_set_audio_scale(t.h, 32768.0))
    rx = oracle.Receiver(fs, 50_000, 128_000, 5_000, oracle.AM, 160, 1_000)
    iq = synth.fm_stream(40_000, fs, [50_000], amp=0.4)
    t.submit_host(iq)
    want = (rx.run(iq)[0].astype(np.float64) * 32768.0).astype(np.float32)
    assert np.array_equal(t.fetch(ch, capi.WR_STAGE_AUDIO, 100), want)
    t.destroy()



def _cfg_d2(d2):
    """fs 2 Msps, D1 = 400 -> 5 kHz, then audio decimation d2"""
    cfg = _mini_c2(8)
    cfg["audio_rate"] = 5_000 // d2 if 5_000 % d2 == 0 else None
    return cfg


@pytest.mark.parametrize("d2,blocks", [(5, [40_000, 40_000, 40_000]), (5, [4_400, 36_000, 400, 800, 20_000, 1_600]),
                                       (5, [399, 401, 12_345, 63 * 400, 64 * 400 + 7]), (1, [20_000, 20_000]),
                                       (2, [20_400, 20_400]), (4, [20_000, 20_000]), (10, [40_000, 40_000])])
def test_fused_post_stage_equals_two_kernel_path(dev, oracle, d2, blocks):
    """k_tuner_post (demod + audio filter in one pass, the default for small audio decimations)
    against the two-kernel path that keeps the demodulator output, and both against the oracle:
    all four detectors, blocks shorter than the audio filter's history, channel-rate frame
    counts that are not multiples of the audio decimation, a decimation outside the fused set."""
    cfg = _cfg_d2(d2)
    modes = [capi.WR_FM, capi.WR_AM, capi.WR_USB, capi.WR_LSB]
    fused, fstates, _ = _run_both(dev, oracle, capi.WR_NCO_EXACT, cfg, modes, blocks, keep_demod=False)
    kept, kstates, _ = _run_both(dev, oracle, capi.WR_NCO_EXACT, cfg, modes, blocks, keep_demod=True)
    same_size = len(set(blocks)) == 1      # (the reference's history breaks when the size changes: Q7)
    for fb, kb in zip(fused, kept):
        for c, ((wa, wc, wd), (fa, fc, _)) in enumerate(fb):
            ka = kb[c][1][0]
            assert fa.size == wa.size == ka.size
            assert np.array_equal(fa.view(np.uint32), ka.view(np.uint32)), c
            if not same_size:
                continue
            assert np.array_equal(fc.view(np.uint32), wc.view(np.uint32)), c
            if modes[c % 4] == capi.WR_FM:
                assert np.abs(fa - wa).max() <= 2 * FM_ATOL
            else:
                assert np.array_equal(fa.view(np.uint32), wa.view(np.uint32)), c
    for (fst, _), (kst, _) in zip(fstates, kstates):
        assert fst[0] == kst[0] and tuple(fst[1]) == tuple(kst[1])


def test_demod_fetch_needs_keep(dev):
    t = Tuner(dev, 2_000_000, 1, 4000)
    ch = t.add_receiver(1000, 128_000, 5_000, capi.WR_AM, 160, 1_000)
    t.submit_host(np.zeros(8000, np.float32))
    with pytest.raises(capi.WrError):
        t.fetch(ch, capi.WR_STAGE_DEMOD, 10)
    t.keep_stages(capi.WR_STAGE_DEMOD)
    t.submit_host(np.zeros(8000, np.float32))
    assert t.fetch(ch, capi.WR_STAGE_DEMOD, 10).size == 10
    t.destroy()


def test_long_stream_of_short_blocks(dev, oracle):
    """700 blocks of five channel-rate frames = one audio frame each (every block rolls all the
    histories and the ping-pong state sets, a fifth of the frames take the block-boundary path),
    retunes on the way: nothing drifts -- the phase stays exact, IQ and audio within tolerance."""
    fs, d1 = 2_000_000, 400
    cfg = _mini_c2(6)
    blk_frames = 5 * d1
    t = Tuner(dev, fs, 6, blk_frames, capi.WR_NCO_ROTATE)
    rxs, chans = [], []
    for c, f in enumerate(cfg["ifs"]):
        m = MODES[c % 4]
        rxs.append(oracle.Receiver(fs, f, cfg["chan_pb"], cfg["chan_rate"], m, cfg["audio_pb"], cfg["audio_rate"]))
        chans.append(t.add_receiver(f, cfg["chan_pb"], cfg["chan_rate"], m, cfg["audio_pb"], cfg["audio_rate"]))
    nblocks = 700
    iq = synth.fm_stream(nblocks * blk_frames, fs, cfg["ifs"][::2], amp=0.3, fm_base=30.0, beta=2.0)
    rng = np.random.default_rng(5)
    worst = 0.0
    audio_g = [[] for _ in chans]
    audio_w = [[] for _ in chans]
    for b in range(nblocks):
        if b % 197 == 100:
            c = int(rng.integers(6)); f = int(rng.integers(-fs // 2, fs // 2))
            rxs[c].set_if(f); t.set_if(chans[c], f)
        blk = iq[2 * b * blk_frames: 2 * (b + 1) * blk_frames]
        t.submit_host(blk)
        check = b % 50 == 49 or b < 70
        for c in range(6):
            wa, wc, wd = rxs[c].run(blk)
            audio_w[c].append(wa)
            audio_g[c].append(t.fetch(chans[c], capi.WR_STAGE_AUDIO, 4))
            if check:
                gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 12)
                worst = max(worst, float(np.abs(gc - wc).max()))
    assert worst <= IQ_ATOL
    for c in range(6):
        ga, wa = np.concatenate(audio_g[c]), np.concatenate(audio_w[c])
        assert ga.size == wa.size == nblocks
        if MODES[c % 4] != capi.WR_FM:
            assert np.abs(ga - wa).max() <= 4e-6, c
        assert t.state(chans[c])[0] == rxs[c].s.phase
    t.destroy()


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
@pytest.mark.parametrize("rates", [(48_000, 48_000, 48_000), (48_000, 24_000, 24_000), (96_000, 48_000, 8_000),
                                   (192_000, 64_000, 64_000)])
def test_channel_decimations_one_two_three(dev, oracle, nco, rates):
    """D1 = 1, 2, 3: every output frame's window reaches into the previous one's, and for the
    first 63 / 32 / 21 frames of a block into the previous block."""
    fs, crate, arate = rates
    t = Tuner(dev, fs, 3, 5000, nco)
    rxs, chs = [], []
    for c, f in enumerate((1000, -7000, 0)):
        m = [oracle.AM, oracle.USB, oracle.LSB][c]
        rxs.append(oracle.Receiver(fs, f, fs // 4, crate, m, crate // 4, arate))
        chs.append(t.add_receiver(f, fs // 4, crate, m, crate // 4, arate))
    n = 5000 - 5000 % (fs // arate)
    for b in range(3):
        iq = synth.fm_stream(n, fs, [1000, -7000], start_frame=b * n, amp=0.3)
        t.submit_host(iq)
        for rx, ch in zip(rxs, chs):
            wa = rx.run(iq)[0]
            ga = t.fetch(ch, capi.WR_STAGE_AUDIO, wa.size + 4)
            assert ga.size == wa.size
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32))
            else:
                assert np.abs(ga - wa).max() <= 4e-6
    t.destroy()


@pytest.mark.parametrize("nco", [capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_one_odd_receiver_costs_its_own_lane_group(dev, oracle, nco):
    """130 receivers, one of them with a different channel filter: lane groups 0 and 1 stay on the
    uniform-taps kernel (with the post stage riding along), group 2 -- which holds the odd one --
    takes the per-lane-taps kernel in a launch of its own.  Then the odd one is retuned to the
    common filter (one launch again) and a receiver of group 0 goes odd instead."""
    cfg = _mini_c2(130)
    fs = cfg["fs"]
    t = Tuner(dev, fs, 130, 40_000, nco)
    rxs, chans = [], []
    for c, f in enumerate(cfg["ifs"]):
        pb = 200_000 if c == 129 else cfg["chan_pb"]
        rxs.append(oracle.Receiver(fs, f, pb, cfg["chan_rate"], oracle.USB, cfg["audio_pb"], cfg["audio_rate"]))
        chans.append(t.add_receiver(f, pb, cfg["chan_rate"], capi.WR_USB, cfg["audio_pb"], cfg["audio_rate"]))
    t.audio_ring(8)
    want = []
    for b in range(6):
        if b == 3:                                    # setPassband while running recomputes the taps (lowpass.cxx:55-61)
            for c, pb in ((129, cfg["chan_pb"]), (7, 300_000)):
                rxs[c].s.chan_fir.coeff[:64] = list(oracle.lowpass_design(pb, fs))
                t.set_filter(chans[c], 0, pb, cfg["chan_rate"])
        iq = synth.fm_stream(40_000, fs, cfg["ifs"][::16], start_frame=b * 40_000, amp=0.1)
        t.submit_host(iq)
        want.append([rx.run(iq)[0] for rx in rxs])
    t.flush()
    for b in range(6):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b
        for c in range(130):
            assert np.abs(audio[c] - want[b][c]).max() <= 4e-6, (b, c)
    t.destroy()


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE, capi.WR_NCO_SPLIT])
def test_more_than_sixteen_lane_groups(dev, oracle, nco):
    """1100 receivers = 18 lane groups: a DDC launch maps 16 of them, the rest go out in launches
    of their own (and one receiver with its own passband sits in group 17).  Receivers of the first,
    the 16th, the 17th and the 18th group against the oracle over two blocks."""
    fs, n, nch = 2_000_000, 20_000, 1100
    ifs = [(-nch // 2 + c) * 900 + 77 for c in range(nch)]
    t = Tuner(dev, fs, nch, n, nco)
    odd = 1090
    chans = [t.add_receiver(f, 128_000 if c != odd else 300_000, 5_000, capi.WR_USB, 160, 1_000)
             for c, f in enumerate(ifs)]
    probe = [0, 63, 960, 1023, 1024, 1025, 1087, 1088, odd, 1099]
    rxs = {c: oracle.Receiver(fs, ifs[c], 128_000 if c != odd else 300_000, 5_000, oracle.USB, 160, 1_000)
           for c in probe}
    start = 0
    for _ in range(2):
        iq = synth.fm_stream(n, fs, [ifs[c] for c in probe[::2]], start_frame=start, seed=3, fm_base=30.0, beta=2.0)
        start += n
        t.submit_host(iq)
        for c in probe:
            wa, wc, _ = rxs[c].run(iq)
            gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)
            ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, n)
            assert gc.size == wc.size and ga.size == wa.size
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), c
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), c
            else:
                assert np.abs(gc - wc).max() <= IQ_ATOL, c
                assert np.abs(ga - wa).max() <= AUDIO_ATOL, c
    t.destroy()
    # the library refuses what it cannot seat
    import ctypes as C
    h = C.c_void_p()
    assert dev.lib.wr_tuner_create(C.byref(h), dev.h, fs, 4097, n, nco) == capi.WR_ERR_ARG


@pytest.mark.parametrize("nco", [capi.WR_NCO_ROTATE, capi.WR_NCO_SPLIT, capi.WR_NCO_EXACT])
def test_few_distinct_channel_filters_share_the_fast_kernel(dev, oracle, nco):
    """receiverhandler.cxx:130-137 gives every receiver its own passband control.  Up to four
    distinct channel filters per lane group stay on the fast (window-folded) ROTATE kernel, one
    copy of the sample window per filter; the fifth sends the group to the per-lane-taps kernel.
    150 receivers: group 0 has 4 filters, group 1 two, group 2 one, then filters change while
    running (one group goes to five and comes back).  Every receiver against the oracle."""
    cfg = _mini_c2(150)
    fs = cfg["fs"]
    pbs = [128_000, 200_000, 64_000, 300_000, 31_250]
    def pb_of(c, phase):
        if c < 64:
            k = c % 4 if phase != 1 else c % 5             # phase 1: five filters in group 0
            return pbs[k]
        if c < 128:
            return pbs[c % 2] if phase < 2 else pbs[0]     # phase 2: group 1 becomes uniform
        return pbs[0]
    t = Tuner(dev, fs, 150, 40_000, nco)
    rxs, chans = [], []
    for c, f in enumerate(cfg["ifs"]):
        rxs.append(oracle.Receiver(fs, f, pb_of(c, 0), cfg["chan_rate"], oracle.LSB, cfg["audio_pb"], cfg["audio_rate"]))
        chans.append(t.add_receiver(f, pb_of(c, 0), cfg["chan_rate"], capi.WR_LSB, cfg["audio_pb"], cfg["audio_rate"]))
    for b in range(6):
        phase = b // 2
        if b in (2, 4):
            for c in range(150):
                if pb_of(c, phase) != pb_of(c, phase - 1):
                    rxs[c].s.chan_fir.coeff[:64] = list(oracle.lowpass_design(pb_of(c, phase), fs))
                    t.set_filter(chans[c], 0, pb_of(c, phase), cfg["chan_rate"])
        iq = synth.fm_stream(40_000, fs, cfg["ifs"][::16], start_frame=b * 40_000, amp=0.1)
        t.submit_host(iq)
        audio = t.fetch_audio_all()
        for c in range(150):
            wa, wc, _ = rxs[c].run(iq)
            ga = audio[t.slot(chans[c])]
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (b, c)
            else:
                assert np.abs(ga - wa).max() <= 4e-6, (b, c)
            if c % 37 == 0:
                gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * 40_000)
                assert np.abs(gc - wc).max() <= (0.0 if nco == capi.WR_NCO_EXACT else IQ_ATOL), (b, c)
    t.destroy()


def test_the_largest_tuner(dev, oracle):
    """WR_MAX_CHANNELS = 4096 receivers = 64 lane groups, four DDC launches of 16 groups a block (the
    post stage of all 64 rides in the first).  Receivers of the first, a middle and the last lane group
    against the oracle over three blocks, the default NCO mode, FM and USB."""
    fs, n, nch = 2_000_000, 24_000, 4096
    ifs = [(-nch // 2 + c) * 240 + 31 for c in range(nch)]
    t = Tuner(dev, fs, nch, n, capi.WR_NCO_ROTATE)
    mode = lambda c: capi.WR_FM if c % 2 else capi.WR_USB
    chans = [t.add_receiver(f, 128_000, 5_000, mode(c), 160, 1_000) for c, f in enumerate(ifs)]
    probe = [0, 1, 63, 1024, 2049, 4032, 4095]
    rxs = {c: oracle.Receiver(fs, ifs[c], 128_000, 5_000, oracle.FM if c % 2 else oracle.USB, 160, 1_000) for c in probe}
    t.audio_ring(4)
    start, want = 0, []
    for b in range(3):
        iq = synth.fm_stream(n, fs, [ifs[c] for c in probe], start_frame=start, seed=5, amp=0.1, fm_base=30.0, beta=2.0)
        start += n
        t.submit_host(iq)
        want.append({c: rxs[c].run(iq) for c in probe})
    t.flush()
    for b in range(3):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b and audio.shape == (4096, n // 400 // 5)
        for c in probe:
            assert np.abs(audio[t.slot(chans[c])] - want[b][c][0]).max() <= AUDIO_ATOL, (b, c)
    for c in probe:                                           # and the last block's channel IQ
        gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)
        assert np.abs(gc - want[2][c][1]).max() <= IQ_ATOL, c
    t.destroy()


@pytest.mark.parametrize("d2,blocks,mode", [(5, [38_400, 39_100, 38_400], capi.WR_USB),
                                            (2, [61_440, 61_440], capi.WR_USB),
                                            (5, [38_400, 38_400], capi.WR_FM)])
def test_post_stage_runs_of_tiles(dev, oracle, d2, blocks, mode):
    """The post stage in runs: with enough tiles a workgroup takes 2 or 4 consecutive 16-frame tiles
    and keeps the staged demodulator rows they share in LDS (post_role).  1024 receivers = 16 lane
    groups; (D2 5, 384 audio frames) = 24 tiles a group: runs of 2, a ragged last tile in the second
    block; (D2 2, 1536 frames) = 96 tiles: runs of 4 and rows carried over that overlap the fresh
    ones; lane group 3 has two audio filters (taps per lane from memory, the others' come through
    the scalar cache).  EXACT mode, USB: the same bits as the oracle; FM: within tolerance."""
    fs, nch = 2_000_000, 1024
    chan_rate, audio_rate = 100_000, 100_000 // d2
    ifs = [(-nch // 2 + c) * 900 + 77 for c in range(nch)]
    t = Tuner(dev, fs, nch, max(blocks), capi.WR_NCO_EXACT)
    apb = lambda c: 3_000 if (192 <= c < 256 and c % 3 == 0) else 4_000        # group 3: mixed audio filters
    chans = [t.add_receiver(f, 40_000, chan_rate, mode, apb(c), audio_rate) for c, f in enumerate(ifs)]
    probe = [0, 63, 192, 193, 195, 255, 600, 960, 1023]
    rxs = {c: oracle.Receiver(fs, ifs[c], 40_000, chan_rate, {capi.WR_USB: oracle.USB, capi.WR_FM: oracle.FM}[mode],
                              apb(c), audio_rate) for c in probe}
    # through the audio ring, blocks submitted back to back: the post stage of every block but the
    # last rides in the next block's launch (taps from memory or the scalar cache), the last one's
    # runs as a kernel of its own (taps in registers)
    t.audio_ring(len(blocks))
    start, want = 0, []
    for n in blocks:
        iq = synth.fm_stream(n, fs, [ifs[c] for c in probe], start_frame=start, seed=11, amp=0.1, fm_base=300.0, beta=2.0)
        start += n
        t.submit_host(iq)
        want.append({c: rxs[c].run(iq)[0] for c in probe})
    t.flush()
    for b, n in enumerate(blocks):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b and audio.shape[1] == n // 20 // d2
        for c in probe:
            ga, wa = audio[t.slot(chans[c])], want[b][c]
            if mode == capi.WR_USB:
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (c, b)
            else:
                assert np.abs(ga - wa).max() <= AUDIO_ATOL, (c, b)
    t.destroy()


def test_every_lane_group_mixed_and_more_than_sixteen_of_them(dev, oracle):
    """1100 receivers, seven channel passbands dealt round the receivers: every one of the 18 lane
    groups holds more than WR_TAPSETS filters, so all DDC launches are of the per-lane-taps variant (16
    groups, then 2), and the previous block's post stage -- all 18 groups -- rides in the first of
    them.  Probes in the first, the 16th, the 17th and the last group against the oracle, three
    blocks through the audio ring."""
    fs, n, nch = 2_000_000, 20_000, 1100
    ifs = [(-nch // 2 + c) * 900 + 77 for c in range(nch)]
    pb = lambda c: 70_000 + 31_250 * (c % 7)
    t = Tuner(dev, fs, nch, n, capi.WR_NCO_ROTATE)
    chans = [t.add_receiver(f, pb(c), 5_000, capi.WR_USB, 160, 1_000) for c, f in enumerate(ifs)]
    probe = [0, 5, 63, 1000, 1023, 1024, 1030, 1087, 1088, 1099]
    rxs = {c: oracle.Receiver(fs, ifs[c], pb(c), 5_000, oracle.USB, 160, 1_000) for c in probe}
    t.audio_ring(3)
    start, want = 0, []
    for _ in range(3):
        iq = synth.fm_stream(n, fs, [ifs[c] for c in probe[::2]], start_frame=start, seed=3, fm_base=30.0, beta=2.0)
        start += n
        t.submit_host(iq)
        want.append({c: rxs[c].run(iq)[0] for c in probe})
    t.flush()
    for b in range(3):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b
        for c in probe:
            assert np.abs(audio[t.slot(chans[c])] - want[b][c]).max() <= AUDIO_ATOL, (b, c)
    t.destroy()
