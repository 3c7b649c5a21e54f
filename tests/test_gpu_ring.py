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
    for iq in blocks:
        t.submit_host(iq)
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
