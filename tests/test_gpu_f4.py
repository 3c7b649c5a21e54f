"""-m gpu: SURVEY 8f-4 inside the fused per-tuner path -- LowPass::_firLength as a run-time value
(dsp/lowpass.cxx:38-39 FIXME), a second channel-filter stage (H4: a 12.5 kHz channel off a fast
stream needs more than one 64-tap stage), and the two receiver controls the reference only stubs
(af_gain, squelch: web/receiverhandler.cxx:112,118-119,127) -- against the oracle."""
import os

import numpy as np
import pytest

from webradio_amd import capi, synth
from webradio_amd.device import Tuner

pytestmark = pytest.mark.gpu
MODE_NAMES = {capi.WR_AM: "AM", capi.WR_FM: "FM", capi.WR_USB: "USB", capi.WR_LSB: "LSB"}


class OracleChain:
    """A Receiver chain assembled from the oracle's blocks with explicit filter lengths:
    mixer -> LowPass(L1, D1) [-> LowPass(L1b, D1b)] -> demodulator -> LowPass(L2, D2)."""

    def __init__(self, oracle, fs, if_hz, l1, pb1, d1, mode, l2, pb2, d2, stage2=None):
        self.o, self.mode = oracle, mode
        self.table = oracle.sin_table()
        self.step = oracle.phase_step(if_hz, fs)
        self.phase = 0
        self.prev = (0.0, 0.0)
        self.f1 = oracle.Fir(2, d1, oracle.lowpass_design(pb1, fs, l1))
        r = fs // d1
        self.f1b = None
        if stage2:
            l1b, pb1b, d1b = stage2
            self.f1b = oracle.Fir(2, d1b, oracle.lowpass_design(pb1b, r, l1b))
            r //= d1b
        self.f2 = oracle.Fir(1, d2, oracle.lowpass_design(pb2, r, l2))

    def run(self, iq):
        mixed, self.phase = self.o.mix(self.table, self.phase, self.step, iq)
        c = self.f1.process(mixed)
        if self.f1b is not None:
            c = self.f1b.process(c)
        d, self.prev = self.o.demod(self.mode, self.prev, c)
        return self.f2.process(d), c, d


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
@pytest.mark.parametrize("lengths", [(32, 64), (16, 16), (64, 8), (32, 32)])
def test_fir_length_inside_the_fused_path(dev, oracle, nco, lengths):
    """Filters of 8..32 taps ride the fused kernels as 64-tap filters whose oldest taps are zero:
    bit-identical to the oracle's `_n` filters in EXACT mode (channel IQ, AM audio), within the
    usual tolerance under ROTATE.  Ragged blocks, 70 receivers (two lane groups)."""
    l1, l2 = lengths
    fs, d1, d2 = 2_000_000, 400, 5
    ifs = [(-35 + c) * 6250 + 321 for c in range(70)]
    probe = [0, 1, 33, 63, 64, 69]
    # (a stream keeps one block size: the reference's LowPass loses its history when the size changes,
    # lowpass.cxx:138-141, quirk Q7 -- the oracle's Fir does too, the product deliberately does not)
    for n, blocks in ((40_000, 3), (2_000, 5), (48_000, 2)):
        t = Tuner(dev, fs, 70, n, nco)
        chans = [t.add_receiver(f, 128_000, fs // d1, capi.WR_AM, 800, fs // d1 // d2, fir_lengths=lengths) for f in ifs]
        rxs = {c: OracleChain(oracle, fs, ifs[c], l1, 128_000, d1, oracle.AM, l2, 800, d2) for c in probe}
        pos = 0
        for _ in range(blocks):
            iq = synth.fm_stream(n, fs, [ifs[c] for c in probe[::2]], start_frame=pos, amp=0.15, fm_base=30.0, beta=2.0)
            pos += n
            t.submit_host(iq)
            for c in probe:
                wa, wc, _ = rxs[c].run(iq)
                gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)
                ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, n)
                assert gc.size == wc.size and ga.size == wa.size
                if nco == capi.WR_NCO_EXACT:
                    assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (n, c)
                    assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (n, c)
                else:
                    assert np.abs(gc - wc).max() <= 1e-6 and np.abs(ga - wa).max() <= 2e-6, (n, c)
        t.destroy()
    # what the fused path does not take is refused, not silently shortened
    import ctypes as C
    t = Tuner(dev, fs, 1, 1000, nco)
    c = C.c_int()
    capi.check(t.lib.wr_chan_add(t.h, C.byref(c)))
    assert t.lib.wr_chan_set_filter_n(t.h, 0, 0, 512, 128_000, 5_000) == capi.WR_ERR_ARG     # every stage: up to 256
    assert t.lib.wr_chan_set_filter_n(t.h, 0, 0, 48, 128_000, 5_000) == capi.WR_ERR_ARG
    capi.check(t.lib.wr_chan_set_filter_n(t.h, 0, 0, 128, 128_000, 5_000))
    assert t.lib.wr_chan_set_filter_n(t.h, 0, 1, 512, 800, 1_000) == capi.WR_ERR_ARG
    assert t.lib.wr_chan_set_filter_n(t.h, 0, 2, 512, 800, 1_000) == capi.WR_ERR_ARG
    capi.check(t.lib.wr_chan_set_filter_n(t.h, 0, 1, 128, 800, 1_000))                        # r05: the audio filter too
    t.destroy()


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
@pytest.mark.parametrize("l1,d1,sizes", [(128, 400, ((40_000, 3), (2_000, 5))), (256, 400, ((48_000, 2), (2_000, 4))),
                                         (256, 40, ((200, 12), (4_000, 3))), (128, 20, ((100, 9),))])
def test_channel_filter_of_128_and_256_taps_inside_the_tuner(dev, oracle, nco, l1, d1, sizes):
    """r03, SURVEY 8f-4: LowPass::_firLength 128 and 256 for the channel filter INSIDE the tuner's launch sequence.
    EXACT: k_tuner_ddc_long, the reference's own arithmetic with the last L - 1 mixed frames kept per channel as
    LowPass::block does -- bit-identical to the oracle's mixer -> `_n` filter -> AM -> audio filter cascade.
    ROTATE: the output frames whose window lies inside the block take k_tuner_ddc_long_rot (L / 64 segments of the
    ROTATE recurrence), the ones that reach into the previous block still the reference's arithmetic -- within the
    ROTATE tolerance on every channel.
    70 receivers (two lane groups), blocks shorter than the filter's history (200 frames against 255), a retune
    in mid-stream (the history keeps the frames as they were mixed, downconverter.cxx:59-67), and a receiver with
    the usual 64 taps beside them in the same tuner (another rate group, the fast kernels)."""
    fs, d2 = 2_000_000, 5
    ifs = [(-35 + c) * 6250 + 321 for c in range(70)]
    probe = [0, 1, 33, 63, 64, 69]
    pb1 = 128_000
    assert oracle.lowpass_maxbin_n(l1, pb1, fs) >= 1
    for n, blocks in sizes:
        t = Tuner(dev, fs, 71, n, nco)
        chans = [t.add_receiver(f, pb1, fs // d1, capi.WR_AM, fs // d1 // 8, fs // d1 // d2, fir_lengths=(l1, 64)) for f in ifs]
        plain = t.add_receiver(4321, pb1, fs // d1, capi.WR_USB, fs // d1 // 8, fs // d1 // d2)
        rxs = {c: OracleChain(oracle, fs, ifs[c], l1, pb1, d1, oracle.AM, 64, fs // d1 // 8, d2) for c in probe}
        rxp = OracleChain(oracle, fs, 4321, 64, pb1, d1, oracle.USB, 64, fs // d1 // 8, d2)
        again = max(1.0, float(np.abs(oracle.lowpass_design(fs // d1 // 8, fs // d1)).sum()))
        t.profile(True)                                    # both rate groups' launches stamp their events
        pos = 0
        for b in range(blocks):
            if b == 1:                                     # retune one of them between two blocks
                t.set_if(chans[33], ifs[33] + 7777)
                rxs[33].step = oracle.phase_step(ifs[33] + 7777, fs)
            iq = synth.fm_stream(n, fs, [ifs[c] for c in probe[::2]] + [4321], start_frame=pos, amp=0.12, fm_base=30.0, beta=2.0)
            pos += n
            t.submit_host(iq)
            for c in probe:
                wa, wc, _ = rxs[c].run(iq)
                gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)
                ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, n)
                assert gc.size == wc.size and ga.size == wa.size
                if nco == capi.WR_NCO_EXACT:
                    assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (n, b, c)
                    assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (n, b, c)
                else:
                    assert np.abs(gc - wc).max() <= 1e-6, (n, b, c, float(np.abs(gc - wc).max()))
                    if wa.size:                            # AM: |.| is 1-Lipschitz, then the linear audio filter
                        assert np.abs(ga - wa).max() <= 2e-6 * again, (n, b, c)
            wa, wc, _ = rxp.run(iq)
            gc = t.fetch(plain, capi.WR_STAGE_CHAN_IQ, 2 * n)
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (n, b)
            else:
                assert np.abs(gc - wc).max() <= 1e-6, (n, b)
        assert float(np.abs(t.fetch(chans[0], capi.WR_STAGE_AUDIO, n)).max()) > 1e-3 or n < d1 * d2     # a live channel
        launches, ms = t.profile_read()
        assert launches == 2 * blocks and 0.0 < ms < 50.0, (launches, ms)
        t.destroy()


@pytest.mark.parametrize("fs,d1,pb1,sizes", [(5_000_000, 20, 160_000, ((200_000, 2), (1_000, 8), (5_000, 4))),
                                             (100_000_000, 400, 6_400_000, ((400_000, 2), (20_000, 5)))])
@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
def test_two_stage_channel_filter_through_the_tuner(dev, oracle, nco, fs, d1, pb1, sizes):
    """a 12.5 kHz NFM-width channel off a fast stream.  One 64-tap LowPass cannot do it
    (64 * 12 500 / fs / 2 = bin 0 for 5 Msps and for 100 Msps, lowpass.cxx:167: all taps zero); the
    reference's own means is a second LowPass in a row.  Here: DDC + first stage fs -> 250 k (off 5 Msps:
    D1 = 20; off 100 Msps, as H4 words it: D1 = 400, passband 6.4 MHz as in BASELINE config 2), second channel
    stage 250 k -> 25 k (D = 10, passband 12.5 kHz -> bin 1), AM, audio filter 25 k -> 5 k -- all through
    wr_tuner_submit, 66 receivers, against the oracle's cascade."""
    assert oracle.lowpass_maxbin(12_500, fs) == 0 and oracle.lowpass_maxbin(12_500, 250_000) == 1
    assert fs // d1 == 250_000 and oracle.lowpass_maxbin(pb1, fs) >= 1
    step = 30_000 if fs == 5_000_000 else 700_000
    ifs = [(-33 + c) * step + 4321 for c in range(66)]
    probe = [0, 31, 63, 64, 65]
    live = 0.0
    # one block size per stream (Q7, see above).  Off 5 Msps 1 000 frames: 50 first-stage, 5 second-stage, 1 audio frame
    for n, blocks in sizes:
        t = Tuner(dev, fs, 66, n, nco)
        chans = [t.add_receiver(f, pb1, 250_000, capi.WR_AM, 4_000, 5_000, stage2=(64, 12_500, 25_000)) for f in ifs]
        rxs = {c: OracleChain(oracle, fs, ifs[c], 64, pb1, d1, oracle.AM, 64, 4_000, 5, stage2=(64, 12_500, 10))
               for c in probe}
        pos = 0
        for _ in range(blocks):
            t0 = (np.arange(n) + pos) / fs
            iq = np.zeros(2 * n, np.float32)
            for c in probe[::2]:                           # AM carriers with a 1 kHz tone
                env = 0.05 * (1.0 + 0.5 * np.cos(2 * np.pi * 1000.0 * t0))
                ph = 2 * np.pi * ((ifs[c] * t0) % 1.0)
                iq[0::2] += (env * np.cos(ph)).astype(np.float32)
                iq[1::2] += (env * np.sin(ph)).astype(np.float32)
            pos += n
            t.submit_host(iq)
            for c in probe:
                wa, wc, _ = rxs[c].run(iq)
                gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)      # the demodulator's input: after stage 2
                ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, n)
                assert gc.size == wc.size and ga.size == wa.size, (n, c)
                if nco == capi.WR_NCO_EXACT:
                    assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (n, c)
                    assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (n, c)
                else:
                    assert np.abs(gc - wc).max() <= 1e-6 and np.abs(ga - wa).max() <= 2e-6, (n, c)
            live = max(live, float(np.abs(t.fetch(chans[0], capi.WR_STAGE_AUDIO, n)).max()))
        t.destroy()
    assert live > 1e-3                                     # a live channel, not zeros


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
@pytest.mark.parametrize("stage,length", [("audio", 128), ("audio", 256), ("second", 128), ("second", 256),
                                          ("both", 128), ("both", 256), ("all", 256)])
def test_audio_filter_and_second_stage_of_128_and_256_taps_inside_the_tuner(dev, oracle, nco, stage, length, d2=5):
    """r05, SURVEY 8f-4: LowPass::_firLength 128 / 256 (lowpass.cxx:38-39 applies to every LowPass;
    radio.cxx:69,71 builds two per receiver) for the AUDIO filter and for a SECOND channel stage, inside the tuner's
    own launch sequence: the rate group is keyed by the lengths, keeps L - 1 rows of history (demodulator output /
    first-stage IQ) as LowPass::block does (lowpass.cxx:138-142), and k_tuner_iq2 / k_tuner_demod + k_tuner_audio add
    the products oldest sample first as lowpass.cxx:150-158 does.
    EXACT: channel IQ and AM/USB/LSB audio bit-identical to the oracle's mixer -> `_n` filter [-> `_n` filter] ->
    detector -> `_n` filter cascade; ROTATE: within the usual tolerance scaled by the filters' absolute gain.
    "all": the channel filter has `length` taps as well (k_tuner_ddc_long).  70 receivers (two lane groups), blocks
    shorter than the histories (the audio filter's 255 rows against 25 demodulator frames per block), a ragged block
    size, a retune in mid-stream, all three linear detectors and FM, a 64-tap receiver beside them in the same tuner;
    no stand-alone block kernel runs (wr_block_kernel_calls).  D2 = 5: the fused demodulator + audio filter
    k_tuner_post<5, L / 64>; the next test: D2 = 7, which takes the two kernels."""
    fs, d1, d1b = (2_000_000 if d2 == 5 else 2_100_000), 40, 5         # (rates stay integer related, dspblock.cxx:119-121)
    two = stage in ("second", "both", "all")
    l1 = length if stage == "all" else 64
    l1b = length if two else None
    l2 = length if stage in ("audio", "both", "all") else 64
    pb1, r1 = 200_000, fs // d1                              # 50 k after the first stage
    r_dem = r1 // d1b if two else r1                         # the demodulator's input rate: 10 k or 50 k
    pb1b, pb2 = r1 // 8, r_dem // 8
    assert oracle.lowpass_maxbin_n(l2, pb2, r_dem) >= 1 and (not two or oracle.lowpass_maxbin_n(l1b, pb1b, r1) >= 1)
    ifs = [(-35 + c) * 6250 + 321 for c in range(70)]
    modes_c = [capi.WR_AM, capi.WR_USB, capi.WR_LSB, capi.WR_FM]
    modes_o = [oracle.AM, oracle.USB, oracle.LSB, oracle.FM]
    probe = [0, 1, 2, 3, 33, 63, 64, 69]
    calls0 = dev.lib.wr_block_kernel_calls()
    unit = d1 * (d1b if two else 1) * d2
    for n, blocks in ((unit * 40, 3), (unit * 5, 8 if two else 14), (unit * 13 + d1 * 3, 4)):
        t = Tuner(dev, fs, 71, n, nco)
        st2 = (l1b, pb1b, r1 // d1b) if two else None
        chans = [t.add_receiver(f, pb1, r1, modes_c[c % 4], pb2, r_dem // d2, fir_lengths=(l1, l2), stage2=st2)
                 for c, f in enumerate(ifs)]
        plain = t.add_receiver(4321, pb1, r1, capi.WR_USB, r1 // 8, r1 // d2)
        rxs = {c: OracleChain(oracle, fs, ifs[c], l1, pb1, d1, modes_o[c % 4], l2, pb2, d2,
                              stage2=(l1b, pb1b, d1b) if two else None) for c in probe}
        rxp = OracleChain(oracle, fs, 4321, 64, pb1, d1, oracle.USB, 64, r1 // 8, d2)
        gain2 = max(1.0, float(np.abs(oracle.lowpass_design(pb2, r_dem, l2)).sum()))
        gain1b = max(1.0, float(np.abs(oracle.lowpass_design(pb1b, r1, l1b)).sum())) if two else 1.0
        pos = 0
        live = 0.0
        for b in range(blocks):
            if b == 1:                                     # retune one of them between two blocks
                t.set_if(chans[33], ifs[33] + 7777)
                rxs[33].step = oracle.phase_step(ifs[33] + 7777, fs)
            iq = synth.fm_stream(n, fs, [ifs[c] for c in probe[::2]] + [4321], start_frame=pos, amp=0.1, fm_base=30.0, beta=2.0)
            pos += n
            t.submit_host(iq)
            for c in probe:
                wa, wc, _ = rxs[c].run(iq)
                gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)      # the demodulator's input
                ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, n)
                assert gc.size == wc.size and ga.size == wa.size, (n, b, c)
                fm = modes_c[c % 4] == capi.WR_FM
                if nco == capi.WR_NCO_EXACT:
                    assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (n, b, c)
                    if not fm:
                        assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (n, b, c)
                    elif wa.size:                          # atan2f: ulps per demodulated frame, then the linear filter
                        assert np.abs(ga - wa).max() <= 4.8e-7 * gain2, (n, b, c)
                else:
                    assert np.abs(gc - wc).max() <= 1e-6 * gain1b, (n, b, c, float(np.abs(gc - wc).max()))
                    if wa.size and not fm:                 # |.| and Re/Im sums are 2-Lipschitz, then the linear filter
                        assert np.abs(ga - wa).max() <= 2e-6 * gain1b * gain2, (n, b, c)
                if wa.size and c == 0:
                    live = max(live, float(np.abs(ga).max()))
            wa, wc, _ = rxp.run(iq)
            gc = t.fetch(plain, capi.WR_STAGE_CHAN_IQ, 2 * n)
            ga = t.fetch(plain, capi.WR_STAGE_AUDIO, n)
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (n, b)
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (n, b)
            else:
                assert np.abs(gc - wc).max() <= 1e-6 and np.abs(ga - wa).max() <= 4e-6, (n, b)
        assert live > 1e-3, (n, live)                      # a live channel, not zeros
        t.destroy()
    assert dev.lib.wr_block_kernel_calls() == calls0       # nothing left the tuner's own launches


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
@pytest.mark.parametrize("stage,length", [("audio", 128), ("both", 256)])
def test_long_audio_filter_with_an_audio_decimation_of_7(dev, oracle, nco, stage, length):
    """... and an audio decimation the fused kernel is not instantiated for: k_tuner_demod + k_tuner_audio with
    L - 1 history rows, inside the tuner's launch sequence all the same."""
    test_audio_filter_and_second_stage_of_128_and_256_taps_inside_the_tuner(dev, oracle, nco, stage, length, d2=7)


@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("length", [128, 256])
def test_long_audio_filter_keeps_its_history_over_a_seek_and_a_kept_demodulator(dev, oracle, length, keep):
    """The same group state under the calls that touch it from outside: wr_tuner_seek (a time-sharded stream starts
    in the middle: empty histories of L - 1 rows, phase in closed form -- keep: made real by a launch, k_tuner_demod +
    k_tuner_audio behind it; not keep: the lazy seek, k_tuner_post<D2, L / 64> reading the all-zero history set),
    wr_tuner_keep_stages(DEMOD) (the demodulator rows sit behind L - 1 history rows now) and a receiver that joins a
    running group (its own history rows zeroed, lowpass.cxx:138-139) -- EXACT mode, bit for bit."""
    fs, d1, d2, n = 2_000_000, 40, 5, 8_000
    ifs = [50_000, -75_000, 4321]
    t = Tuner(dev, fs, 4, n, capi.WR_NCO_EXACT)
    if keep:
        t.keep_stages(capi.WR_STAGE_DEMOD)
    chans = [t.add_receiver(f, 200_000, fs // d1, capi.WR_AM, 6_250, fs // d1 // d2, fir_lengths=(64, length)) for f in ifs]
    start = 123_456 * d1 * d2                                # the shard begins here
    t.seek(start)
    rxs = [OracleChain(oracle, fs, f, 64, 200_000, d1, oracle.AM, length, 6_250, d2) for f in ifs]
    for rx in rxs:
        rx.phase = (rx.step * start) & 0x7FFFFFFF
    late = None
    for b in range(4):
        if b == 2:                                         # a fourth receiver joins: fresh blocks, the phase of a fresh mixer
            chans.append(t.add_receiver(-4444, 200_000, fs // d1, capi.WR_USB, 6_250, fs // d1 // d2, fir_lengths=(64, length)))
            late = OracleChain(oracle, fs, -4444, 64, 200_000, d1, oracle.USB, length, 6_250, d2)
            rxs.append(late)
        iq = synth.fm_stream(n, fs, ifs[:2], start_frame=start + b * n, amp=0.2, fm_base=30.0, beta=2.0)
        t.submit_host(iq)
        for ch, rx in zip(chans, rxs):
            wa, wc, wd = rx.run(iq)
            assert np.array_equal(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n).view(np.uint32), wc.view(np.uint32)), b
            if keep:
                assert np.array_equal(t.fetch(ch, capi.WR_STAGE_DEMOD, n).view(np.uint32), wd.view(np.uint32)), b
            assert np.array_equal(t.fetch(ch, capi.WR_STAGE_AUDIO, n).view(np.uint32), wa.view(np.uint32)), b
    t.destroy()


def test_long_filters_at_c2_full_size(dev, oracle):
    """BASELINE config 2's size (256 receivers, one 4 M-frame block off 100 Msps, input made on the device) with a second
    channel stage of 128 taps (250 k -> 50 k) and an audio filter of 256 taps (50 k -> 10 k), AM:
      - EXACT: three receivers against the oracle's cascade on the first 400 000 frames (causal filters: a prefix of the
        input gives a prefix of every stage), bit for bit -- second-stage IQ and audio;
      - ROTATE (the shipped mode: k_tuner_iq2, its post stage k_tuner_post<5, 4> riding/launched as the tuner decides)
        against EXACT on ALL 256 receivers within the ROTATE tolerance through the two filters' absolute gains."""
    import torch
    c2 = synth.C2
    fs, n = c2["input_rate"], c2["block_frames"]
    ifs = synth.c2_ifs()
    d1 = fs // c2["chan_rate"]
    r1 = c2["chan_rate"]
    st2 = (128, r1 // 16, r1 // 5)
    pb2, arate = r1 // 5 // 8, r1 // 5 // 5
    x = synth.fm_stream_torch(n, fs, ifs[::4], "cuda")
    torch.cuda.synchronize()
    outs = {}
    for mode in (capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE):
        t = Tuner(dev, fs, 256, n, mode)
        chans = [t.add_receiver(f, c2["chan_passband"], r1, capi.WR_AM, pb2, arate, fir_lengths=(64, 256), stage2=st2) for f in ifs]
        t.submit_device(x, n)
        dev.sync()
        k1b, k2 = n // d1 // 5, n // d1 // 5 // 5
        outs[mode] = [(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * k1b), t.fetch(ch, capi.WR_STAGE_AUDIO, k2)) for ch in chans]
        assert outs[mode][0][0].size == 2 * k1b and outs[mode][0][1].size == k2
        t.destroy()
    g1b = max(1.0, float(np.abs(oracle.lowpass_design(st2[1], r1, st2[0])).sum()))
    g2 = max(1.0, float(np.abs(oracle.lowpass_design(pb2, r1 // 5, 256)).sum()))
    worst_iq = max(float(np.abs(a[0] - b[0]).max()) for a, b in zip(outs[capi.WR_NCO_EXACT], outs[capi.WR_NCO_ROTATE]))
    worst_au = max(float(np.abs(a[1] - b[1]).max()) for a, b in zip(outs[capi.WR_NCO_EXACT], outs[capi.WR_NCO_ROTATE]))
    assert worst_iq <= 1e-6 * g1b and worst_au <= 2e-6 * g1b * g2, (worst_iq, worst_au)
    assert max(float(np.abs(a[1]).max()) for a in outs[capi.WR_NCO_EXACT][::4]) > 1e-3       # the carrier channels are live
    m = 400_000
    xh = x[: 2 * m].cpu().numpy()
    for c in (0, 128, 252):
        rx = OracleChain(oracle, fs, ifs[c], 64, c2["chan_passband"], d1, oracle.AM, 256, pb2, 5, stage2=(128, st2[1], 5))
        wa, wc, _ = rx.run(xh)
        got_iq, got_au = outs[capi.WR_NCO_EXACT][c]
        assert wc.size and wa.size
        assert np.array_equal(got_iq[: wc.size].view(np.uint32), wc.view(np.uint32)), c
        assert np.array_equal(got_au[: wa.size].view(np.uint32), wa.view(np.uint32)), c


@pytest.mark.parametrize("keep_demod", [False, True])
def test_af_gain_and_squelch(dev, oracle, keep_demod):
    """af_gain / squelch (named and left as FIXMEs by the reference, receiverhandler.cxx:112-127): the
    build's definition (include/webradio_amd.h) against its scalar restatement, bit for bit, in the
    fused post stage and in the two-kernel path; 0 dB and an open squelch change nothing."""
    fs, d1, d2 = 2_000_000, 400, 5
    ifs = [50_000, -75_000, 4321, 99_999]
    t = Tuner(dev, fs, 4, 40_000, capi.WR_NCO_EXACT)
    if keep_demod:
        t.keep_stages(capi.WR_STAGE_DEMOD)
    chans = [t.add_receiver(f, 128_000, 5_000, capi.WR_AM, 160, 1_000) for f in ifs]
    rxs = [oracle.Receiver(fs, f, 128_000, 5_000, oracle.AM, 160, 1_000) for f in ifs]
    settings = [(0.0, None), (6.0, None), (-3.5, -46.0), (0.0, -46.0)]
    pos = 0
    for b in range(3):
        if b == 1:                                         # staged, applied from this block on
            for ch, (g, sq) in zip(chans, settings):
                t.set_af_gain(ch, g)
                t.set_squelch(ch, sq if sq is not None else 0.0, sq is not None)
        # the carrier of receivers 2 and 3 fades in and out: the squelch opens and closes within a block
        n = 40_000
        tt = (np.arange(n) + pos) / fs
        iq = (0.002 * np.random.default_rng(b).standard_normal(2 * n)).astype(np.float32)
        for f in ifs:
            env = 0.02 * (1.0 + np.sign(np.sin(2 * np.pi * 150.0 * tt))) * 0.5
            ph = 2 * np.pi * ((f * tt) % 1.0)
            iq[0::2] += (env * np.cos(ph)).astype(np.float32)
            iq[1::2] += (env * np.sin(ph)).astype(np.float32)
        pos += n
        t.submit_host(iq)
        for c, (ch, rx) in enumerate(zip(chans, rxs)):
            wa, wc, _ = rx.run(iq)
            g, sq = settings[c] if b >= 1 else (0.0, None)
            want = oracle.af_gain_squelch(wa, wc, d2, g, sq)
            got = t.fetch(ch, capi.WR_STAGE_AUDIO, n)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (b, c)
            if sq is not None:
                muted = np.count_nonzero(want == 0.0)
                assert 0 < muted < want.size                # the gate really opens and closes
    t.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("WR_FUZZ_SEEDS", "8"))))
def test_random_f4_configurations(dev, oracle, seed):
    """Seeded mixtures of everything above in one tuner: filter lengths per receiver, receivers with
    and without a second channel stage (two rate groups), af_gain and squelch on some, detectors
    mixed, passbands mixed (tap sets), one block size per stream -- every receiver against the
    oracle's cascade; AM/USB/LSB bit for bit in EXACT mode."""
    rng = np.random.default_rng(4000 + seed)
    nco = (capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE)[seed % 2]
    fs, d1, d1b, d2 = 2_000_000, 40, 10, 5                   # 2 M -> 50 k [-> 5 k] -> /5
    n = int(rng.choice([d1 * d1b * d2 * 4, d1 * d1b * d2 * 9, 40_000]))
    nchan = int(rng.choice([3, 20, 70]))
    modes = [capi.WR_AM, capi.WR_USB, capi.WR_LSB]
    t = Tuner(dev, fs, nchan, n, nco)
    specs, chans, rxs = [], [], []
    for c in range(nchan):
        f = int(rng.integers(-fs // 2 + 1, fs // 2))
        l1, l2 = int(rng.choice([8, 16, 32, 64, 64, 128, 256])), int(rng.choice([16, 64, 64, 128, 256]))   # 128 / 256: k_tuner_ddc_long (r03); r05: the audio filter and the second stage too
        pb1 = int(rng.choice([fs // 16, fs // 8, fs // 5]))
        two = bool(rng.integers(0, 2))
        mode = modes[int(rng.integers(0, 3))]
        r_in = fs // d1 // (d1b if two else 1)                # the demodulator's input rate
        pb2 = r_in // 8
        st2 = (int(rng.choice([32, 64, 128, 256])), (fs // d1) // 8, fs // d1 // d1b) if two else None
        gain = float(rng.choice([0.0, 0.0, 6.0, -12.5]))
        sq = float(rng.choice([-60.0, -35.0])) if rng.integers(0, 3) == 0 else None
        specs.append((f, two, gain, sq))
        chans.append(t.add_receiver(f, pb1, fs // d1, mode, pb2, r_in // d2, fir_lengths=(l1, l2), stage2=st2))
        rxs.append(OracleChain(oracle, fs, f, l1, pb1, d1, mode, l2, pb2, d2,
                               stage2=(st2[0], st2[1], d1b) if two else None))
        t.set_af_gain(chans[c], gain)
        t.set_squelch(chans[c], sq if sq is not None else 0.0, sq is not None)
    carriers = [specs[c][0] for c in range(0, nchan, max(1, nchan // 3))][:3]
    pos = 0
    for b in range(3):
        iq = synth.fm_stream(n, fs, carriers, start_frame=pos, amp=0.2, fm_base=40.0, beta=1.5, seed=seed)
        pos += n
        t.submit_host(iq)
        for c in range(nchan):
            wa, wc, _ = rxs[c].run(iq)
            want = oracle.af_gain_squelch(wa, wc, d2, specs[c][2], specs[c][3])
            gc = t.fetch(chans[c], capi.WR_STAGE_CHAN_IQ, 2 * n)
            ga = t.fetch(chans[c], capi.WR_STAGE_AUDIO, n)
            assert gc.size == wc.size and ga.size == want.size, (seed, b, c)
            if nco == capi.WR_NCO_EXACT:
                assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32)), (seed, b, c)
                assert np.array_equal(ga.view(np.uint32), want.view(np.uint32)), (seed, b, c)
            else:
                assert np.abs(gc - wc).max() <= 1e-6, (seed, b, c)
                g = 10.0 ** (specs[c][2] / 20.0)
                # a squelch decision may differ where the power sits within 1e-6 of the threshold
                differ = np.abs(ga - want) > 4e-6 * max(1.0, g)
                if specs[c][3] is None:
                    assert not differ.any(), (seed, b, c)
                else:
                    assert differ.sum() <= 2, (seed, b, c)
    t.destroy()


@pytest.mark.parametrize("d2", [8, 10])
def test_fused_post_stage_for_audio_decimations_8_and_10(dev, oracle, d2):
    """r03: the fused demodulator + audio filter (k_tuner_post / the post role riding in the next block's DDC
    launch) also for D2 = 8 and 10 -- 256 k -> 32 k is BASELINE config 1, 240 k -> 24 k and 480 k -> 48 k the
    reference's own defaults (radio.cxx:79-81).  70 receivers (two lane groups, all four detectors) off 2 Msps,
    D1 = 20, four blocks through the audio ring in the host runtime's mode (ROTATE):
      - the post stage riding in the next launch = the post stage launched on its own (WR_DEFER_POST=0), bit for bit;
      - against the oracle: AM/USB/LSB within the ROTATE tolerance through the audio filter, FM on carriers."""
    import os
    fs, d1, nch = 2_000_000, 20, 70
    chan_rate, audio_rate = fs // d1, fs // d1 // d2
    ifs = [(-nch // 2 + c) * 9_000 + 311 for c in range(nch)]
    modes_c = [capi.WR_FM, capi.WR_AM, capi.WR_USB, capi.WR_LSB]
    modes_o = [oracle.FM, oracle.AM, oracle.USB, oracle.LSB]
    n = d1 * d2 * 16 * 5 + d1 * d2 * 3                        # five tiles of 16 audio frames and a ragged one
    blocks = [synth.fm_stream(n, fs, ifs[::4], start_frame=b * n, seed=5, amp=0.2, fm_base=200.0, beta=2.0) for b in range(4)]
    rxs = [oracle.Receiver(fs, f, 250_000, chan_rate, modes_o[c % 4], audio_rate // 2, audio_rate) for c, f in enumerate(ifs)]
    want = [np.stack([rx.run(iq)[0] for rx in rxs]) for iq in blocks]
    outs = []
    for env in ("1", "0"):
        os.environ["WR_DEFER_POST"] = env
        try:
            t = Tuner(dev, fs, nch, n, capi.WR_NCO_ROTATE)
        finally:
            del os.environ["WR_DEFER_POST"]
        chans = [t.add_receiver(f, 250_000, chan_rate, modes_c[c % 4], audio_rate // 2, audio_rate) for c, f in enumerate(ifs)]
        t.audio_ring(len(blocks))
        for iq in blocks:
            t.submit_host(iq)
        t.flush()
        got = []
        for b in range(len(blocks)):
            a, seq = t.ring_acquire()
            assert seq == b and a.shape[1] == n // d1 // d2
            got.append(np.stack([a[t.slot(ch)] for ch in chans]))
            t.ring_release()
        outs.append(got)
        t.destroy()
    taps2 = oracle.lowpass_design(audio_rate // 2, chan_rate)
    tol = 2e-6 * max(1.0, float(np.abs(taps2).sum()))
    for b in range(len(blocks)):
        assert np.array_equal(outs[0][b].view(np.uint32), outs[1][b].view(np.uint32)), b
        for c in range(nch):
            if c % 4 == 0:                                     # the FM receivers: they sit on the carriers (ifs[::4])
                assert np.abs(outs[0][b][c] - want[b][c]).max() <= 1e-5, (b, c)
            else:
                assert np.abs(outs[0][b][c] - want[b][c]).max() <= tol, (b, c)
