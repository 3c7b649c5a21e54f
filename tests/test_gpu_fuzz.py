"""-m gpu: seeded random configurations of the fused path against the oracle -- channel counts,
IFs over the whole band (negative, zero, near Nyquist), passbands incl. the degenerate
maxbin = 0, all four detectors mixed in one tuner, decimations below and above the FIR length,
ragged block sizes, retunes / mode / passband changes between blocks."""
import os

import numpy as np
import pytest

from webradio_amd import capi, synth
from webradio_amd.device import Tuner

pytestmark = pytest.mark.gpu

FM_ATOL = 2.4e-7
RATES = [  # (fs, chan_rate, audio_rate): integer related
    (2_000_000, 5_000, 1_000), (2_000_000, 250_000, 50_000), (2_048_000, 256_000, 32_000),
    (240_000, 24_000, 8_000), (1_000_000, 100_000, 100_000), (960_000, 48_000, 48_000),
]


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "12"))))
def test_random_configuration(dev, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    fs, crate, arate = RATES[seed % len(RATES)]
    d1, d2 = fs // crate, crate // arate
    nchan = int(rng.choice([1, 2, 5, 33, 64, 70]))
    nco = (capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE)[seed % 3]
    base = d1 * d2
    block = int(rng.choice([base * 3, base * 7 + int(rng.integers(0, base)), 4096, 10_000]))
    block = max(block, 1)
    if block > 60_000:
        block = base * 2 + 5
    ifs = [int(v) for v in rng.integers(-fs // 2 + 1, fs // 2, nchan)]
    ifs[0] = 0
    cpbs = [int(rng.choice([fs // 40, fs // 16, fs // 8, fs // 3, 0])) for _ in range(nchan)]
    apbs = [int(rng.choice([crate // 30, crate // 8, crate // 4])) for _ in range(nchan)]
    modes = [int(rng.integers(0, 4)) for _ in range(nchan)]
    t = Tuner(dev, fs, nchan, block, nco)
    rxs, chans = [], []
    for c in range(nchan):
        rxs.append(oracle.Receiver(fs, ifs[c], cpbs[c], crate, modes[c], apbs[c], arate))
        chans.append(t.add_receiver(ifs[c], cpbs[c], crate, modes[c], apbs[c], arate))
    # the audio filter is linear: an error of e in its input gives at most e * sum|taps| out
    again = [max(1.0, float(np.abs(oracle.lowpass_design(apbs[c], crate)).sum())) for c in range(nchan)]
    carriers = [ifs[c] for c in range(0, nchan, max(1, nchan // 4))][:4]
    # a channel that has ever run the FM detector carries device-atan2f values in its audio
    # filter history: within tolerance, no longer bit-exact
    was_fm = [m == capi.WR_FM for m in modes]
    pos = 0
    for b in range(4):
        if b == 2:                                   # control-plane changes at a block boundary
            for c in range(0, nchan, 3):
                ifs[c] = int(rng.integers(-fs // 2 + 1, fs // 2))
                modes[c] = int(rng.integers(0, 4))
                was_fm[c] = was_fm[c] or modes[c] == capi.WR_FM
                rxs[c].set_if(ifs[c]); rxs[c].set_mode(modes[c])
                t.set_if(chans[c], ifs[c]); t.set_mode(chans[c], modes[c])
        iq = synth.fm_stream(block, fs, carriers, start_frame=pos, amp=0.5 / max(len(carriers), 1),
                             fm_base=fs / 70_000.0, fm_step=fs / 300_000.0, beta=2.0, seed=seed)
        pos += block
        t.submit_host(iq)
        k1 = block // d1
        for c in range(nchan):
            wa, wc, wd = rxs[c].run(iq)
            gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * k1 + 2)
            ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, k1 + 2)
            assert gc.size == wc.size and ga.size == wa.size
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (seed, b, c)
                if not was_fm[c]:
                    assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (seed, b, c)
                elif modes[c] != capi.WR_FM or (cpbs[c] and ifs[c] in carriers):
                    assert np.abs(ga - wa).max() <= 4 * FM_ATOL, (seed, b, c)
            else:
                assert np.abs(gc - wc).max() <= 1e-6, (seed, b, c)
                if modes[c] != capi.WR_FM and wa.size:
                    # detector input within 1e-6 (USB/LSB add two components: 2e-6)
                    assert np.abs(ga - wa).max() <= 2e-6 * again[c], (seed, b, c)
    for c in range(nchan):
        ph, _ = t.state(chans[c])
        assert ph == rxs[c].s.phase
    t.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "8"))))
def test_random_spectrum_streams(dev, oracle, seed):
    """SpectrumSink: random size, hop and push chunking against the oracle (which is fed, for
    hops below the frame size, the overlapped frames explicitly)."""
    from webradio_amd.device import Spectrum
    rng = np.random.default_rng(500 + seed)
    n = int(2 ** rng.integers(3, 15))
    hop = int(rng.choice([0, n, n // 2, max(1, n // 4), max(1, (3 * n) // 4)]))
    total = int(n * 3 + rng.integers(0, 2 * n))
    iq = synth.fm_stream(total, 2_400_000, [100_000, -450_000], amp=0.3, noise_dbfs=-45, seed=seed)
    s = Spectrum(dev, n, hop)
    eff = hop if hop else n
    pos, frames = 0, 0
    while pos < total:
        chunk = int(min(total - pos, rng.integers(1, 2 * n)))
        s.push_host(iq[2 * pos: 2 * (pos + chunk)])
        pos += chunk
        frames = 0 if pos < n else (pos - n) // eff + 1
        assert s.frames_done() == frames
        if frames:
            start = (frames - 1) * eff
            o = oracle.Spectrum(n)
            o.process(iq[2 * start: 2 * (start + n)])
            wb, wd = o.bins(), o.get()
            peak = np.abs(wb[0::2] + 1j * wb[1::2]).max()
            assert np.abs(s.get_bins() - wb).max() <= 2e-6 * peak
            strong = wd >= wd.max() - 60.0
            assert np.abs(s.get_db() - wd)[strong].max() <= 0.02
    s.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "6"))))
def test_random_standalone_blocks(dev, oracle, seed):
    """mix / fir / demod kernels with random geometry: bit-exact (FM within atan2f ulps)."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(1, 30_000))
    iq = rng.uniform(-1, 1, 2 * n).astype(np.float32)
    step = oracle.phase_step(int(rng.integers(-1_199_999, 1_199_999)), 2_400_000)
    ph = int(rng.integers(0, 1 << 31))
    want, pw = oracle.mix(oracle.sin_table(), ph, step, iq)
    got, pg = dev.mix(iq, ph, step)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and pg == pw
    ch, dec = int(rng.integers(1, 4)), int(rng.integers(1, 130))
    coeff = oracle.lowpass_design(int(rng.integers(0, 1_000_000)), 2_400_000)
    fir = oracle.Fir(ch, dec, coeff)
    hist = dev.malloc(63 * ch * 4)
    frames = int(rng.integers(1, 5000))
    for _ in range(3):
        x = rng.uniform(-1, 1, frames * ch).astype(np.float32)
        assert np.array_equal(dev.fir_decimate(x, ch, dec, coeff, hist).view(np.uint32), fir.process(x).view(np.uint32))
    dev.free(hist)
    for mode in range(4):
        w, _ = oracle.demod(mode, (0.25, -0.5), iq)
        g, _ = dev.demod(mode, iq, (0.25, -0.5))
        if mode == capi.WR_FM:
            assert np.abs(g - w).max() <= FM_ATOL
        else:
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "9"))))
def test_random_streams_through_the_ring(dev, oracle, seed):
    """The same kind of random configuration, consumed the way a streaming sink would: no fetch
    between submits, every block's audio taken from the pinned ring afterwards.  With one
    channel filter and one audio filter per lane group (what radio.cxx sets up) the demod + audio
    filter of a block then run inside the NEXT block's launch (wr_tuner_flush in the header)."""
    rng = np.random.default_rng(5000 + seed)
    # every audio decimation the riding post stage is instantiated for (1..6), and one it is not (8)
    rates = RATES + [(2_000_000, 5_000, 2_500), (2_000_000, 5_000, 1_250), (1_200_000, 6_000, 1_000)]
    fs, crate, arate = rates[seed % len(rates)]
    d1, d2 = fs // crate, crate // arate
    nchan = int(rng.choice([1, 3, 64, 65, 130]))
    base = d1 * d2
    block = int(rng.choice([base * 2, base * 9, base * 5 + int(rng.integers(0, base)), 3 * 4096]))
    if block > 60_000:
        block = base * 3 + 1
    cpb, apb = int(rng.choice([fs // 16, fs // 8, fs // 3])), int(rng.choice([crate // 8, crate // 4]))
    ifs = [int(v) for v in rng.integers(-fs // 2 + 1, fs // 2, nchan)]
    modes = [int(rng.integers(0, 4)) for _ in range(nchan)]
    t = Tuner(dev, fs, nchan, block)
    rxs, chans = [], []
    for c in range(nchan):
        rxs.append(oracle.Receiver(fs, ifs[c], cpb, crate, modes[c], apb, arate))
        chans.append(t.add_receiver(ifs[c], cpb, crate, modes[c], apb, arate))
    gain = max(1.0, float(np.abs(oracle.lowpass_design(apb, crate)).sum()))
    nb = 6
    t.audio_ring(nb)
    carriers = ifs[:: max(1, nchan // 3)][:3]
    want, pos = [], 0
    was_fm = [m == capi.WR_FM for m in modes]
    for b in range(nb):
        if b == 3:
            c = int(rng.integers(nchan))
            ifs[c] = int(rng.integers(-fs // 2 + 1, fs // 2))
            modes[c] = int(rng.integers(0, 4))
            was_fm[c] = was_fm[c] or modes[c] == capi.WR_FM
            rxs[c].set_if(ifs[c]); rxs[c].set_mode(modes[c])
            t.set_if(chans[c], ifs[c]); t.set_mode(chans[c], modes[c])
        iq = synth.fm_stream(block, fs, carriers, start_frame=pos, amp=0.5 / len(carriers), fm_base=fs / 70_000.0,
                             fm_step=fs / 300_000.0, beta=2.0, seed=seed)
        pos += block
        t.submit_host(iq)
        want.append([rx.run(iq)[0] for rx in rxs])
    t.flush()
    assert t.ring_stats() == (nb, 0)
    for b in range(nb):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b and audio.shape[1] == want[b][0].size
        for c in range(nchan):
            if not was_fm[c] and want[b][c].size:
                assert np.abs(audio[c] - want[b][c]).max() <= 2e-6 * gain, (seed, b, c)
    t.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "12"))))
def test_random_streams_with_blocks_per_launch(dev, seed):
    """wr_tuner_set_blocks_per_launch under a random stream of calls: the same sequence of submits
    (blocks that follow on in device memory and blocks that do not, whole and ragged sizes, host
    blocks), setters (IF, mode, passband, af_gain), flushes and state reads goes through a tuner that
    launches every block on its own and one that holds up to 2-5 blocks.  Everything either of them
    hands out through the audio ring, laid end to end, is the same bits; so are the NCO phases read
    on the way."""
    import torch
    rng = np.random.default_rng(7000 + seed)
    fs, crate, arate = RATES[seed % len(RATES)]
    d1, d2 = fs // crate, crate // arate
    q = d1 * d2                                             # frames per audio frame
    nchan = int(rng.choice([3, 40, 64, 130]))
    nco = (capi.WR_NCO_ROTATE, capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT)[seed % 3]
    whole = q * int(rng.integers(2, 9))                     # the usual block
    if whole > 40_000:
        whole = q * 2
    hold = int(rng.integers(2, 6))
    total = whole * 40
    ifs = [int(v) for v in rng.integers(-fs // 2 + 1, fs // 2, nchan)]
    iq = synth.fm_stream(total, fs, ifs[:3], amp=0.15, fm_base=fs / 70_000.0, beta=2.0, seed=seed)
    x = torch.from_numpy(iq).cuda()

    # the script: a list of operations both tuners replay
    ops, pos = [], 0
    while pos + 2 * whole < total and len(ops) < 70:
        r = rng.random()
        if r < 0.60:
            ops.append(("dev", pos, whole)); pos += whole
        elif r < 0.68:                                      # ragged: not a whole number of audio frames
            n = whole + int(rng.integers(1, q))
            ops.append(("dev", pos, n)); pos += n
        elif r < 0.74:                                      # a block from elsewhere: does not follow on
            ops.append(("dev_copy", pos, whole)); pos += whole
        elif r < 0.80:
            ops.append(("host", pos, whole)); pos += whole
        elif r < 0.86:
            ops.append(("set_if", int(rng.integers(0, nchan)), int(rng.integers(-fs // 2 + 1, fs // 2))))
        elif r < 0.90:
            ops.append(("set_mode", int(rng.integers(0, nchan)), int(rng.integers(0, 4))))
        elif r < 0.93:
            ops.append(("set_filter", int(rng.integers(0, nchan)), int(rng.choice([fs // 40, fs // 16, fs // 8]))))
        elif r < 0.95:
            ops.append(("gain", int(rng.integers(0, nchan)), float(rng.choice([-6.0, 0.0, 3.5]))))
        elif r < 0.98:
            ops.append(("flush",))
        else:
            ops.append(("state", int(rng.integers(0, nchan))))

    def play(join):
        t = Tuner(dev, fs, nchan, whole * hold + q, nco)
        chans = [t.add_receiver(f, fs // 16, crate, int(m), crate // 8, arate)
                 for f, m in zip(ifs, rng2.integers(0, 4, nchan))]
        t.audio_ring(128)
        if join:
            t.blocks_per_launch(hold)
        rows, states = [], []

        def drain():
            while t.ring_stats()[0]:
                a, _ = t.ring_acquire()
                rows.append(a.copy())
                t.ring_release()
        for op in ops:
            if op[0] == "dev":
                t.submit_device(x[2 * op[1]: 2 * (op[1] + op[2])], op[2])
            elif op[0] == "dev_copy":
                t.submit_device(x[2 * op[1]: 2 * (op[1] + op[2])].clone(), op[2])
                torch.cuda.synchronize()                    # (the clone must outlive the launch that reads it)
                t.flush()
                dev.sync()
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
        t.destroy()
        return np.concatenate([r[slots] for r in rows], axis=1) if rows else np.zeros((nchan, 0), np.float32), states

    rng2 = np.random.default_rng(seed)
    one, st1 = play(False)
    rng2 = np.random.default_rng(seed)
    many, st2 = play(True)
    assert one.shape == many.shape and one.shape[1] > 0
    assert st1 == st2
    assert np.array_equal(one.view(np.uint32), many.view(np.uint32))


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "12"))))
def test_random_post_stage_runs(dev, oracle, seed):
    """The riding post stage where it takes runs of tiles (16 lane groups, enough audio frames for runs
    of two or four): random audio decimation 1-6, detector per receiver, mixed and shared audio
    filters per lane group, ragged block sizes.  EXACT NCO mode: AM/USB/LSB receivers are the
    oracle's bits, FM ones within tolerance; blocks through the audio ring, so every block but the last
    went through the riding variant and the last through the kernel of its own."""
    rng = np.random.default_rng(9000 + seed)
    fs, crate = 2_400_000, 120_000
    d2 = int(rng.integers(1, 7))
    arate = crate // d2
    nch = 1024
    tiles = int(rng.choice([24, 30, 96, 100]))                 # x 16 groups: 384.. -> runs of 2, 1536.. -> runs of 4
    k2 = tiles * 16 - int(rng.integers(0, 16))                # a ragged last tile, mostly
    base = k2 * d2 * 20
    blocks = [base, base + int(rng.integers(0, 20 * d2)), max(20 * d2, base - 20 * d2 * int(rng.integers(0, 40)))]
    ifs = [(-nch // 2 + c) * 1100 + 13 for c in range(nch)]
    modes = rng.integers(0, 4, nch)
    mixed_group = int(rng.integers(0, 16))                    # one lane group with two audio filters
    apb = lambda c: arate // 5 if (c // 64 == mixed_group and c % 2) else arate // 4
    t = Tuner(dev, fs, nch, max(blocks), capi.WR_NCO_EXACT)
    chans = [t.add_receiver(f, 50_000, crate, int(modes[c]), apb(c), arate) for c, f in enumerate(ifs)]
    probe = [int(v) for v in rng.choice(nch, 6, replace=False)] + [mixed_group * 64, mixed_group * 64 + 1]
    rxs = {c: oracle.Receiver(fs, ifs[c], 50_000, crate, int(modes[c]), apb(c), arate) for c in probe}
    t.audio_ring(len(blocks))
    start, want = 0, []
    for n in blocks:
        iq = synth.fm_stream(n, fs, [ifs[c] for c in probe[:4]], start_frame=start, seed=seed, amp=0.1, fm_base=300.0, beta=2.0)
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
            if modes[c] == capi.WR_FM:
                gain = max(1.0, float(np.abs(oracle.lowpass_design(apb(c), crate)).sum()))
                assert np.abs(ga - wa).max() <= FM_ATOL * gain * 4, (c, b)
            else:
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (c, b, int(modes[c]))
    t.destroy()
