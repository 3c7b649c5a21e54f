"""-m gpu: the config-5 time-sharding driver with the product's TunerShard on one GPU
(single rank and, over gloo, two ranks sharing the GPU): identical to one sequential pass."""
import os
import sys

import numpy as np
import pytest

from webradio_amd import capi, synth, timeshard
from webradio_amd.device import Tuner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS, D1, D2 = 2_000_000, 400, 5
IFS = [50_000, -75_000, 1234, 99_999]
T, N = 60_000, 4


def _stream():
    return synth.fm_stream(T * N, FS, IFS[:2], amp=0.3, fm_base=30.0, beta=2.0)


def _sequential(dev, nco):
    t = Tuner(dev, FS, len(IFS), T * N, nco)
    chans = [t.add_receiver(f, 128_000, 5_000, capi.WR_FM, 160, 1_000) for f in IFS]
    t.submit_host(_stream())
    a = np.stack([t.fetch(ch, capi.WR_STAGE_AUDIO, T * N) for ch in chans])
    t.destroy()
    return a


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_SPLIT, capi.WR_NCO_ROTATE])
def test_time_shard_single_rank_bit_identical(dev, nco):
    iq = _stream()
    shard = timeshard.TunerShard(dev, FS, IFS, 128_000, 5_000, capi.WR_FM, 160, 1_000, T + timeshard.halo_frames(D1, D2), nco)

    class Solo:
        rank, world = 0, 1
        def exchange(self, tail):
            return None
    out = timeshard.run_time_sharded(Solo(), lambda c: iq[2 * c * T: 2 * (c + 1) * T], N, T, D1, D2, shard,
                                     lambda a: a, lambda a: a)
    shard.close()
    got = np.concatenate([out[c] for c in range(N)], axis=1)
    want = _sequential(dev, nco)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("between", ["nothing", "getter", "setter", "seek_twice", "kept_demod", "no_fetch"])
def test_lazy_seek_equals_the_sequential_pass(dev, between):
    """r03: wr_tuner_seek launches nothing where the next submit's DDC can take the phase in closed form and read
    all-zero state sets (one launch per chunk: the post stage of the chunk before rides in it).  Whatever comes between
    the seek and the submit -- a getter, a setter (its group upload), a second seek, the demodulator rows kept (two-kernel
    path: the seek is made real first) -- and also when NOTHING is fetched between the chunks (so that the chunk before
    really rides), every chunk is the sequential pass's bits."""
    nco = capi.WR_NCO_ROTATE if between == "no_fetch" else capi.WR_NCO_EXACT     # (ROTATE: the mode whose post stage rides)
    iq = _stream()
    H = timeshard.halo_frames(D1, D2)
    t = Tuner(dev, FS, len(IFS), T + H, nco)
    chans = [t.add_receiver(f, 128_000, 5_000, capi.WR_FM, 160, 1_000) for f in IFS]
    if between == "kept_demod":
        t.keep_stages(capi.WR_STAGE_DEMOD)
    if between == "no_fetch":
        t.audio_ring(N)
    drop = H // (D1 * D2)
    got = []
    for c in range(N):
        first = max(0, c * T - H)
        block = iq[2 * first: 2 * (c + 1) * T]
        if between == "seek_twice":
            t.seek(12345)
        t.seek(first)
        if between == "getter":
            ph, prev = t.state(chans[1])
            assert not prev.any()
        if between == "setter":
            t.set_if(chans[2], IFS[2])                       # the same value: marks the group for an upload
        t.submit_host(block)
        if between != "no_fetch":
            a = np.stack([t.fetch(ch, capi.WR_STAGE_AUDIO, T + H) for ch in chans])
            got.append(a[:, (drop if c else 0):])
    if between == "no_fetch":
        t.flush()
        for c in range(N):
            a, seq = t.ring_acquire()
            t.ring_release()
            assert seq == c
            slots = [t.slot(ch) for ch in chans]
            got.append(a[slots][:, (drop if c else 0):])
    t.destroy()
    got = np.concatenate(got, axis=1)
    want = _sequential(dev, nco)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_native_ring_world_one_bit_identical(dev):
    """The C ABI's halo ring (wr_ring_*: ncclSend / ncclRecv on RCCL, its own stream, events to the device's
    stream) at world size 1, where the rank is its own neighbour: every chunk's halo really travels through
    ncclSend/ncclRecv, the chunks live in HBM, and the audio is the sequential pass's, bit for bit."""
    import torch
    from webradio_amd.device import Ring
    assert Ring.rccl_version() > 20000                                   # RCCL loaded and answers (2.x.y -> 2xxyy)
    iq = torch.from_numpy(_stream()).cuda()
    shard = timeshard.TunerShard(dev, FS, IFS, 128_000, 5_000, capi.WR_FM, 160, 1_000, T + timeshard.halo_frames(D1, D2),
                                 capi.WR_NCO_SPLIT)
    ring = timeshard.RingHalo(None, 0, 1, dev=dev)
    out = timeshard.run_time_sharded(ring, lambda c: iq[2 * c * T: 2 * (c + 1) * T], N, T, D1, D2, shard,
                                     lambda a: a.contiguous(), lambda t: t)
    assert ring.native.exchanges() == N
    ring.close()
    shard.close()
    got = np.concatenate([out[c] for c in range(N)], axis=1)
    want = _sequential(dev, capi.WR_NCO_SPLIT)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _receivers(which):
    if which == "uniform-and-odd-groups":
        # 64 receivers on one channel filter (a lane group for the uniform-taps kernel) + 6 with six different filters
        # (more than WR_TAPSETS: the per-lane-taps kernel): TWO DDC launches per submit
        rx = [((c - 32) * 20_000 + 777, 128_000) for c in range(64)]
        return rx + [(900_000 - 30_000 * i, pb) for i, pb in enumerate((64_000, 128_000, 190_000, 250_000, 320_000, 380_000))]
    return [(f, 128_000) for f in IFS]


@pytest.mark.parametrize("which", ["plain", "profiled-every-launch", "uniform-and-odd-groups", "marking-switched-on-late"])
def test_ring_exchange_after_the_tuners_launches(dev, which):
    """wr_ring_exchange_after + wr_tuner_mark_launches, used the way bench.py --workload c5 uses them: resident chunks in a
    rotation of three [halo | chunk] buffers, every halo through ncclSend / ncclRecv (world 1: to itself), posted a round
    ahead and ordered behind the TUNER's launches (their own completion events) instead of an event record on the
    device's stream; nothing fetched between the chunks (one launch per chunk: lazy seek, riding post stage).
    The audio is the sequential pass's, bit for bit.

    r04: also with every launch profiled (the launch's stop event is the profiler's then: the mark is an
    ordinary record), with two DDC launches per submit (the LAST one carries the mark), and with marking switched on
    when a block is already in flight."""
    import torch
    nco = capi.WR_NCO_ROTATE
    H = timeshard.halo_frames(D1, D2)
    rxs = _receivers(which)
    iq = torch.from_numpy(_stream()).cuda()
    nb = 3
    bufs = [torch.zeros(2 * (H + T), device="cuda") for _ in range(nb)]
    t = Tuner(dev, FS, len(rxs), T + H, nco)
    chans = [t.add_receiver(f, pb, 5_000, capi.WR_FM, 160, 1_000) for f, pb in rxs]
    t.audio_ring(N)
    late = which == "marking-switched-on-late"
    if not late:
        t.mark_launches(True)
    if which == "profiled-every-launch":
        t.profile(1)
    ring = timeshard.RingHalo(None, 0, 1, dev=dev)

    def load(c):                                              # chunk c into its buffer (behind the halo)
        bufs[c % nb][2 * H:].copy_(iq[2 * c * T: 2 * (c + 1) * T])

    load(0)
    if late:
        # a tuner that does not mark its launches cannot order an exchange: an error, not a silent race
        assert dev.lib.wr_ring_exchange_after(ring.native.h, t.h, capi.ptr(bufs[0][2 * T:]), capi.ptr(bufs[1][: 2 * H]),
                                              2 * H) == capi.WR_ERR_STATE
        assert b"wr_tuner_mark_launches" in dev.lib.wr_last_error()
    for c in range(N):
        if c + 1 < N and not (late and c == 0):
            load(c + 1)
            # chunk c + 1's halo = the last H frames of chunk c: to the ring neighbour (ourselves), a round ahead
            ring.post(bufs[c % nb][2 * T:], bufs[(c + 1) % nb][: 2 * H], t)
        if c == 0:
            t.seek(0)
            t.submit_device(bufs[0][2 * H:], T)
        else:
            t.seek(c * T - H)
            t.submit_device(bufs[c % nb], H + T)
        if late and c == 0:
            t.mark_launches(True)                             # chunk 0 is in flight, unmarked: one record stands for it
            load(1)
            ring.post(bufs[0][2 * T:], bufs[1][: 2 * H], t)
        if c + 1 < N:
            ring.wait()                                       # before the next chunk's submit reads its halo
    t.flush()
    if which == "profiled-every-launch":
        launches, ms = t.profile_read()
        assert launches == N and ms > 0
        t.profile(False)
    got = []
    drop = H // (D1 * D2)
    slots = [t.slot(ch) for ch in chans]
    for c in range(N):
        a, seq = t.ring_acquire()
        t.ring_release()
        assert seq == c
        got.append(a[slots][:, (drop if c else 0):])
    assert ring.native.exchanges() == N - 1
    ring.close()
    t.destroy()
    got = np.concatenate(got, axis=1)
    t = Tuner(dev, FS, len(rxs), T * N, nco)                  # the sequential pass of the same receivers
    chans = [t.add_receiver(f, pb, 5_000, capi.WR_FM, 160, 1_000) for f, pb in rxs]
    t.submit_host(_stream())
    want = np.stack([t.fetch(ch, capi.WR_STAGE_AUDIO, T * N) for ch in chans])
    t.destroy()
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the argument checks of the new entry points
    assert dev.lib.wr_tuner_mark_launches(None, 1) == capi.WR_ERR_ARG
    assert dev.lib.wr_ring_exchange_after(None, None, None, None, 0) == capi.WR_ERR_ARG


def test_ring_entry_points_argument_checks(dev):
    """wr_ring_* fail with WR_ERR_ARG and a message on bad arguments, before RCCL is touched."""
    import ctypes as C
    lib = capi.load()
    assert lib.wr_ring_id_bytes() == 128
    h = C.c_void_p()
    ident = (C.c_ubyte * 128)()
    assert lib.wr_ring_make_id(ident, 64) == capi.WR_ERR_ARG
    assert lib.wr_ring_create(C.byref(h), dev.h, ident, 128, 2, 2) == capi.WR_ERR_ARG      # rank out of range
    assert lib.wr_ring_create(C.byref(h), None, ident, 128, 0, 1) == capi.WR_ERR_ARG
    assert lib.wr_ring_exchange(None, None, None, 0) == capi.WR_ERR_ARG
    assert lib.wr_ring_wait(None) == capi.WR_ERR_ARG and b"wr_ring_wait" in lib.wr_last_error()
    assert lib.wr_ring_destroy(None) == capi.WR_OK


def _worker(rank, world, outdir):
    sys.path.insert(0, ROOT)
    import torch
    from webradio_amd.device import Device
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _proc
    dist = _proc.init_gloo(rank, world, os.path.join(outdir, "rdzv"))      # (one GPU on the test box: a gloo ring)
    dev = Device(0)
    iq = _stream()
    shard = timeshard.TunerShard(dev, FS, IFS, 128_000, 5_000, capi.WR_FM, 160, 1_000, T + timeshard.halo_frames(D1, D2))
    out = timeshard.run_time_sharded(timeshard.RingHalo(dist, rank, world), lambda c: iq[2 * c * T: 2 * (c + 1) * T],
                                     N, T, D1, D2, shard, lambda a: torch.from_numpy(np.ascontiguousarray(a)),
                                     lambda t: t.numpy())
    shard.close()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{str(c): a for c, a in out.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_time_shard_two_ranks(dev, tmp_path):
    import _proc
    _proc.spawn_ranks(_worker, 2, (2, str(tmp_path)), timeout=150)
    parts = {}
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        for k in d.files:
            parts[int(k)] = d[k]
    got = np.concatenate([parts[c] for c in range(N)], axis=1)
    want = _sequential(dev, capi.WR_NCO_SPLIT)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("nco", [capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE])
def test_c5_parameters_one_rank_against_oracle(dev, oracle, nco):
    """BASELINE config 5 on its own parameters: fs = 1 Gsps, 256 channels on the 3.125 MHz raster,
    D1 = 4000 with the 64 MHz passband whose product 64 * passband just fits 32 bits
    (lowpass.cxx:167), D2 = 5, halo 260 000 frames.  Three chunks time-sharded on one rank
    (wr_tuner_seek + one [halo | chunk] block each) against the ORACLE's sequential pass over the
    same stream, on a sample of the channels."""
    c5 = synth.C5
    fs, d1, d2 = c5["input_rate"], 4000, 5
    H = timeshard.halo_frames(d1, d2)
    assert H == 260_000 and timeshard.discarded_audio_frames(d1, d2) == 13
    assert oracle.lowpass_maxbin(c5["chan_passband"], fs) == 2
    ifs = synth.c2_ifs(256, c5)
    T, n = 280_000, 3
    probe = [0, 1, 63, 64, 128, 200, 255]
    iq = synth.fm_stream(T * n, fs, [ifs[c] for c in probe[::2]], amp=0.1, fm_base=3000.0, fm_step=500.0, beta=2.0)
    shard = timeshard.TunerShard(dev, fs, ifs, c5["chan_passband"], c5["chan_rate"], capi.WR_FM, c5["audio_passband"],
                                 c5["audio_rate"], T + H, nco)

    class Solo:
        rank, world = 0, 1
        def exchange(self, tail):
            return None
    out = timeshard.run_time_sharded(Solo(), lambda c: iq[2 * c * T: 2 * (c + 1) * T], n, T, d1, d2, shard,
                                     lambda a: a, lambda a: a)
    shard.close()
    got = np.concatenate([out[c] for c in range(n)], axis=1)
    assert got.shape == (256, T * n // (d1 * d2))
    for c in probe:
        rx = oracle.Receiver(fs, ifs[c], c5["chan_passband"], c5["chan_rate"], oracle.FM, c5["audio_passband"],
                             c5["audio_rate"])
        want = rx.run(iq)[0]
        tol = 4.8e-7 if nco == capi.WR_NCO_EXACT else 1e-5
        if c in probe[::2] or nco == capi.WR_NCO_EXACT:      # FM on a noise-only channel is ill-conditioned
            assert np.abs(got[c] - want).max() <= tol, c
