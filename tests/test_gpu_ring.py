"""-m gpu: the pinned audio ring (include/webradio_amd.h wr_tuner_audio_ring*): what the
AudioStreamManager sinks of a tuner would consume (web/audiostream.cxx:65-73) without one
device-to-host call per channel and without stalling the producer."""
import threading

import numpy as np
import pytest

from webradio_amd import capi, synth
from webradio_amd.device import Tuner

pytestmark = pytest.mark.gpu

FS, N = 2_000_000, 40_000
IFS = [(-4 + c) * 6250 + 1234 for c in range(70)]       # 70 channels: two lane groups, one ragged


def _tuner(dev):
    t = Tuner(dev, FS, len(IFS), N)
    for c, f in enumerate(IFS):
        t.add_receiver(f, 128_000, 5_000, [capi.WR_FM, capi.WR_AM][c % 2], 160, 1_000)
    return t


def _blocks(n):
    return [synth.fm_stream(N, FS, IFS[::4], start_frame=b * N, amp=0.3) for b in range(n)]


def _expected(dev, blocks):
    t = _tuner(dev)
    out = []
    for iq in blocks:
        t.submit_host(iq)
        out.append(np.stack([t.fetch(c, capi.WR_STAGE_AUDIO, 100) for c in range(len(IFS))]))
    t.destroy()
    return out


def test_ring_delivers_every_block_in_order(dev):
    blocks = _blocks(3)
    want = _expected(dev, blocks)
    t = _tuner(dev)
    t.audio_ring(3)
    assert t.submit_count() == 0
    for b, iq in enumerate(blocks):
        assert t.submit_count() == b            # the number this submit's ring entry will carry (what TunerBatch asks for)
        t.submit_host(iq)
    assert t.submit_count() == 3
    assert t.ring_stats()[0] in (2, 3)          # the last block's post stage may wait for the next launch
    t.flush()
    assert t.ring_stats() == (3, 0)
    for b in range(3):
        audio, seq = t.ring_acquire()
        assert seq == b and audio.shape == (128, 20)              # slots used (2 lane groups), k2
        assert np.array_equal(audio[: len(IFS)], want[b])
        t.ring_release()
    with pytest.raises(capi.WrError):
        t.ring_acquire()                                            # nothing queued
    t.destroy()


def test_ring_overrun_drops_the_new_block(dev):
    blocks = _blocks(5)
    want = _expected(dev, blocks)
    t = _tuner(dev)
    t.audio_ring(2)
    for iq in blocks[:4]:
        t.submit_host(iq)
    t.flush()
    assert t.ring_stats() == (2, 2)                                 # blocks 2 and 3 were dropped
    a0, s0 = t.ring_acquire()
    with pytest.raises(capi.WrError):
        t.ring_acquire()                                            # one slot at a time
    t.ring_release()
    a1, s1 = t.ring_acquire()
    t.ring_release()
    assert (s0, s1) == (0, 1) and np.array_equal(a0[: len(IFS)], want[0]) and np.array_equal(a1[: len(IFS)], want[1])
    t.submit_host(blocks[4])                                        # the stream itself never stopped
    t.flush()
    a4, s4 = t.ring_acquire()
    t.ring_release()
    assert s4 == 4 and np.array_equal(a4[: len(IFS)], want[4])
    with pytest.raises(capi.WrError):
        t.ring_release()
    t.audio_ring(0)
    with pytest.raises(capi.WrError):
        t.ring_acquire()
    t.destroy()


def test_ring_consumer_thread(dev):
    """producer = the pipeline thread submitting blocks, consumer = another thread, as the
    reference's HTTP connection threads are (web/audiostream.cxx)"""
    nb = 12
    blocks = _blocks(4)
    want = _expected(dev, [blocks[b % 4] for b in range(nb)])
    t = _tuner(dev)
    t.audio_ring(nb)
    got, err = {}, []

    def consumer():
        try:
            while len(got) < nb:
                try:
                    audio, seq = t.ring_acquire()
                except capi.WrError:
                    continue                                        # nothing queued yet
                got[seq] = audio[: len(IFS)].copy()
                t.ring_release()
        except Exception as e:                                      # pragma: no cover
            err.append(e)

    th = threading.Thread(target=consumer)
    th.start()
    for b in range(nb):
        t.submit_host(blocks[b % 4])
    t.flush()                                                       # end of the stream
    th.join(60)
    assert not th.is_alive() and not err
    assert sorted(got) == list(range(nb)) and t.ring_stats() == (0, 0)
    for b in range(nb):
        assert np.array_equal(got[b], want[b]), b
    t.destroy()


def test_deferred_post_stage_keeps_block_order(dev, oracle):
    """The demod + audio filter of block b run inside the launch of block b+1 (wr_tuner_flush in the
    header).  Nothing may be observed out of order: mode / IF / audio passband changes made
    between two submits (no fetch in between, so a post stage IS pending) apply to the right
    block, a keep-stages request switches paths mid-stream, a flush in the middle is harmless.
    Every block's audio, taken from the ring afterwards, against the oracle."""
    nb = 10
    blocks = _blocks(nb)
    rxs = [oracle.Receiver(FS, f, 128_000, 5_000, [oracle.FM, oracle.AM][c % 2], 160, 1_000) for c, f in enumerate(IFS)]
    t = _tuner(dev)
    t.audio_ring(nb)
    want = []
    for b, iq in enumerate(blocks):
        if b == 2:                                    # detector change while block 1's post stage is pending
            rxs[3].set_mode(oracle.USB); t.set_mode(3, capi.WR_USB)
        if b == 4:
            rxs[5].set_if(4321); t.set_if(5, 4321)
        if b == 5:
            t.flush(); t.flush()
        if b == 6:
            t.keep_stages(capi.WR_STAGE_DEMOD)       # two-kernel path from here on ...
        if b == 8:
            t.keep_stages()                           # ... and back
        t.submit_host(iq)
        want.append(np.stack([rx.run(iq)[0] for rx in rxs]))
    t.flush()
    assert t.ring_stats() == (nb, 0)
    for b in range(nb):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b
        for c in range(len(IFS)):
            if rxs[c].s.mode == oracle.FM or (c == 3 and b < 2):
                continue                               # FM: device atan2 (checked elsewhere within tolerance)
            assert np.abs(audio[c] - want[b][c]).max() <= 4e-6, (b, c)
    t.destroy()


def test_deferral_off_gives_the_same_bits(dev):
    import os
    blocks = _blocks(4)
    # ragged stream: blocks too short for a channel-rate frame (k1 = 0: no post stage at all), too
    # short for an audio frame (k2 = 0: only the end-of-block state), and ordinary ones
    long = np.concatenate(_blocks(3))
    cuts = [0, 40_000, 40_100, 40_500, 42_500, 42_501, 82_501, 83_301, 120_000]
    blocks = blocks + [long[2 * a: 2 * b] for a, b in zip(cuts[:-1], cuts[1:])]
    outs = []
    for env in ("1", "0"):
        os.environ["WR_DEFER_POST"] = env
        try:
            t = _tuner(dev)
        finally:
            del os.environ["WR_DEFER_POST"]
        t.audio_ring(len(blocks))
        for iq in blocks:
            t.submit_host(iq)
        t.flush()
        got = []
        for b in range(len(blocks)):
            a, seq = t.ring_acquire()
            assert seq == b
            got.append(a.copy())
            t.ring_release()
        outs.append(got)
        t.destroy()
    assert [g.shape for g in outs[0]] == [g.shape for g in outs[1]]
    assert sum(g.shape[1] for g in outs[0]) > 0
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_receivers_come_and_go_while_a_post_stage_is_pending(dev, oracle):
    """Receivers added and removed between two submits with nothing fetched in between
    (radio.cxx:151-163 does this from the REST thread): the pending post stage of the block
    before is launched before the channel tables change, a new receiver starts from empty
    filters, the others are undisturbed."""
    blocks = _blocks(6)
    t = Tuner(dev, FS, 200, N)
    rx, ch = {}, {}
    def add(key, f, mode_o, mode_c):
        rx[key] = oracle.Receiver(FS, f, 128_000, 5_000, mode_o, 160, 1_000)
        ch[key] = t.add_receiver(f, 128_000, 5_000, mode_c, 160, 1_000)
    for c in range(66):                              # two lane groups from the start
        add(c, IFS[c % len(IFS)] + 17 * c, oracle.AM, capi.WR_AM)
    t.audio_ring(6)
    want = []
    for b, iq in enumerate(blocks):
        if b == 2:
            add("late", 7777, oracle.USB, capi.WR_USB)
        if b == 3:
            t.remove_receiver(ch.pop(5)); rx.pop(5)
        if b == 4:
            for c in range(100, 170):                 # a third lane group appears
                add(c, -50_000 + 700 * c, oracle.LSB, capi.WR_LSB)
        t.submit_host(iq)
        want.append({k: r.run(iq)[0] for k, r in rx.items()})
        slots = {k: t.lib and _slot(t, v) for k, v in ch.items()}
        want[-1]["_slots"] = slots
    t.flush()
    for b in range(6):
        audio, seq = t.ring_acquire()
        t.ring_release()
        assert seq == b
        slots = want[b].pop("_slots")
        for k, w in want[b].items():
            assert np.abs(audio[slots[k]] - w).max() <= 4e-6, (b, k)       # default NCO mode: within tolerance
    t.destroy()


def _slot(t, chan):
    import ctypes as C
    s = C.c_int()
    capi.check(t.lib.wr_chan_slot(t.h, chan, C.byref(s)))
    return s.value


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
def test_blocks_per_launch_gives_the_same_bits(dev, oracle, nco):
    """wr_tuner_set_blocks_per_launch: back-to-back device blocks held and launched as one.  Same
    stream through a tuner that launches every block and one that joins up to three: the audio is
    the same bits -- whole groups, a group cut short by a fetch, a block that does not follow on in
    memory, a retune staged between two held blocks (it must take effect at ITS block boundary)."""
    import torch
    fs, n, nch = 2_000_000, 40_000, 70
    ifs = [(-35 + c) * 6250 + 99 for c in range(nch)]
    nblk = 9
    iq = synth.fm_stream(nblk * n, fs, ifs[::9], amp=0.1, fm_base=30.0, beta=2.0)
    x = torch.from_numpy(iq).cuda()
    other = x[: 2 * n].clone()                                  # block 5 comes from somewhere else in memory
    other.copy_(x[2 * 5 * n: 2 * 6 * n])

    def run(join, fetch_at):
        t = Tuner(dev, fs, nch, 3 * n, nco)
        chans = [t.add_receiver(f, 128_000, 5_000, capi.WR_USB, 160, 1_000) for f in ifs]
        if join:
            t.blocks_per_launch(3)
        out = []
        for b in range(nblk):
            if b == 7:
                t.set_if(chans[3], ifs[3] + 1234)               # staged: applies from block 7 on
            src = other if b == 5 else x[2 * b * n: 2 * (b + 1) * n]
            t.submit_device(src, n)
            if b in fetch_at:                                   # reading results launches what is held
                out.append(t.fetch_audio_all().copy())
        t.destroy()
        return np.concatenate(out, axis=1)

    k2 = n // 2000
    every = run(False, set(range(nblk)))
    assert every.shape[1] == nblk * k2
    # launches the joining tuner forms: [0,1,2]  [3,4] (block 5 does not follow on in memory)  [5] (cut
    # by the fetch)  [6] (cut by the staged retune)  [7,8] (cut by the final fetch)
    grouped = run(True, {2, 4, 5, 6, 8})
    assert grouped.shape == every.shape
    assert np.array_equal(every.view(np.uint32), grouped.view(np.uint32))
    # a fetch in the middle of a group launches the part that is there: [0,1] [2,3,4] [5] ([6] not
    # fetched) [7,8]
    cut = run(True, {1, 4, 5, 8})
    keep = np.r_[0:6 * k2, 7 * k2:9 * k2]
    assert cut.shape[1] == 8 * k2
    assert np.array_equal(every[:, keep].view(np.uint32), cut.view(np.uint32))


def test_blocks_per_launch_at_c2_size(dev):
    """bench.py's configuration: 256 receivers, 4 M-frame blocks off 100 Msps, four blocks per launch
    (ROTATE, the post stage in runs of four tiles riding in the next launch).  Eight consecutive
    resident blocks through a tuner that launches each on its own and through one that joins four:
    every audio sample of every receiver is the same bits."""
    import torch
    c2 = synth.C2
    fs, n = c2["input_rate"], c2["block_frames"]
    ifs = synth.c2_ifs()
    nblk, B = 8, 4
    x = synth.fm_stream_torch(n * nblk, fs, ifs[::4], "cuda")
    torch.cuda.synchronize()

    def run(join):
        t = Tuner(dev, fs, 256, n * B, capi.WR_NCO_ROTATE)
        for f in ifs:
            t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"], c2["audio_rate"])
        t.blocks_per_launch(B if join else 1)
        out = []
        for b in range(nblk):
            t.submit_device(x[2 * n * b: 2 * n * (b + 1)], n)
            if not join or b % B == B - 1:
                out.append(t.fetch_audio_all().copy())
        t.destroy()
        return np.concatenate(out, axis=1)

    one, four = run(False), run(True)
    assert one.shape == four.shape == (256, nblk * n // 400 // 5)
    assert np.array_equal(one.view(np.uint32), four.view(np.uint32))
    assert float(np.abs(one[::4]).max()) > 0.0                  # the carrier channels carry audio
