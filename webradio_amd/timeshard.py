"""Multi-GPU sharding of the hot path.

Config 4 -- independent tuners: one tuner (FrontEnd + its receivers) per GPU, no data-path
collective at all (`tuners_for_rank`).  bench.py uses this.

Config 5 -- ONE wideband stream cut in time: consecutive chunks of T frames go round-robin to
the ranks.  Every dependency of the path is finite (63-frame FIR histories, one
previous channel-rate frame for FM, closed-form NCO phase), so a rank can compute its chunk
from scratch if it is also given the last

    H = ceil(((63 + 1) * D1 + 63) / (D1 * D2)) * D1 * D2      frames

of the previous chunk: it runs [halo | chunk] as one block from an empty state with the NCO
phase set to its closed-form value at the first halo frame, and throws away the first
H / (D1*D2) audio frames (exactly the ones an empty history contaminates).  The result is
bit-identical to processing the stream sequentially.  The halo travels from the ring
neighbour with one send/recv pair per chunk (RCCL over one xGMI link when the tensors are in
HBM and the backend is "nccl"; the same code runs over gloo on CPU for the tests).
"""
import numpy as np

FIR = 64


def tuners_for_rank(n_tuners, rank, world):
    """Config 4: tuner t runs on rank t mod world."""
    return [t for t in range(n_tuners) if t % world == rank]


def halo_frames(d1, d2):
    need = (FIR - 1 + 1) * d1 + (FIR - 1)
    q = d1 * d2
    return -(-need // q) * q


def discarded_audio_frames(d1, d2):
    return halo_frames(d1, d2) // (d1 * d2)


def phase_at(step, frame):
    """DownConverter::phase after `frame` input frames from phase 0 (downconverter.cxx:103)."""
    return (int(step) * int(frame)) % (1 << 31)


class RingHalo:
    """Tail-of-chunk exchange between ring neighbours: rank r sends to r+1, receives from r-1.

    A thin caller.  With a wr_dev (`dev`) the pair goes through the C ABI's wr_ring_* -- RCCL's
    ncclSend / ncclRecv directly, on the ring's own stream, so `post` can be issued a round ahead and
    `wait` costs the device's stream one event (bench.py --workload c5).  Without one (the CPU
    tests' gloo ranks, host tensors) it is torch.distributed's batch_isend_irecv, synchronous."""

    def __init__(self, dist, rank, world, dev=None):
        self.dist, self.rank, self.world = dist, rank, world
        self.native = None
        if dev is not None:
            from .device import Ring
            ident = [Ring.make_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ident, src=0)        # 128 bytes, once
            self.native = Ring(dev, ident[0], rank, world)

    def post(self, tail, recv, tuner=None):
        """native only: enqueue the pair (tail -> rank + 1, recv <- rank - 1) behind what the device's
        stream holds now -- or, with a `tuner` that marks its launches, behind that tuner's launches only
        (the halo is input and `recv` a buffer nobody else reads: nothing goes on the device's stream);
        returns at once"""
        if tuner is not None:
            self.native.exchange_after(tuner, tail, recv, tail.numel())
        else:
            self.native.exchange(tail, recv, tail.numel())

    def wait(self):
        self.native.wait()

    def exchange(self, tail):
        """tail: 1-D float32 torch tensor (2*H floats) on the backend's device.  Returns the
        tail sent by the previous rank (None for world == 1 without a native ring)."""
        import torch
        if self.native is not None:
            recv = torch.empty_like(tail)
            self.post(tail, recv)
            self.wait()
            return recv
        if self.world == 1:
            return None
        recv = torch.empty_like(tail)
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        ops = [self.dist.P2POp(self.dist.isend, tail, nxt), self.dist.P2POp(self.dist.irecv, recv, prv)]
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        return recv

    def close(self):
        if self.native is not None:
            self.native.destroy()
            self.native = None


def run_time_sharded(ring, get_chunk, nchunks, chunk_frames, d1, d2, process, to_tensor, from_tensor):
    """Round-robin time sharding of one stream.

    get_chunk(c)          -> interleaved float32 IQ of chunk c (2*chunk_frames floats), as whatever
                             `process` consumes (numpy on CPU, torch cuda tensor on GPU)
    process(block, start_frame, nframes) -> audio array [channels][nframes // (d1*d2)] computed
                             from an EMPTY state with the NCO phase of `start_frame`
    to_tensor / from_tensor  convert a chunk tail to/from the tensor type of the backend
    Returns {chunk index: audio [channels][chunk_frames // (d1*d2)]} for this rank's chunks.
    """
    rank, world = ring.rank, ring.world
    H = halo_frames(d1, d2)
    drop = discarded_audio_frames(d1, d2)
    assert chunk_frames % (d1 * d2) == 0 and chunk_frames >= H
    out = {}
    carried = None                      # rank 0: the tail that arrived during the previous round
    rounds = -(-nchunks // world)
    for i in range(rounds):
        c = i * world + rank
        have = c < nchunks
        chunk = get_chunk(c) if have else get_chunk(nchunks - 1)     # idle ranks still take part in the ring
        tail = to_tensor(chunk[2 * (chunk_frames - H):])
        received = ring.exchange(tail)
        if world == 1 and received is None:
            halo, carried = carried, chunk[2 * (chunk_frames - H):]
        elif rank == 0:
            halo, carried = carried, from_tensor(received)           # needed next round
        else:
            halo = from_tensor(received)
        if not have:
            continue
        if c == 0:
            out[c] = process(chunk, 0, chunk_frames)
        else:
            block = _concat(halo, chunk)
            audio = process(block, c * chunk_frames - H, chunk_frames + H)
            out[c] = audio[:, drop:]
    return out


def _concat(a, b):
    if isinstance(a, np.ndarray):
        return np.concatenate([a, b])
    import torch
    return torch.cat([a, b])


class TunerShard:
    """`process` callback on a GPU: a wr_tuner that starts every block from the state of a stream
    beginning at the block's first frame (wr_tuner_seek: one call for all channels) and hands the
    audio of all channels back with one transfer (wr_tuner_fetch_audio_all)."""

    def __init__(self, dev, input_rate, ifs, chan_passband, chan_rate, mode, audio_passband, audio_rate,
                 max_frames, nco=0):
        from .device import Tuner
        from . import capi
        self.capi = capi
        self.t = Tuner(dev, input_rate, len(ifs), max_frames, nco)
        self.dev = dev
        self.ch = [self.t.add_receiver(f, chan_passband, chan_rate, mode, audio_passband, audio_rate) for f in ifs]
        self.slots = None
        self.d = (input_rate // chan_rate) * (chan_rate // audio_rate)

    def submit(self, block, start_frame, nframes):
        """seek + submit, nothing fetched: the audio stays in HBM (wr_tuner_audio_dev)"""
        self.t.seek(start_frame)
        if isinstance(block, np.ndarray):
            self.t.submit_host(block)
        else:
            self.t.submit_device(block, nframes)

    def __call__(self, block, start_frame, nframes):
        self.submit(block, start_frame, nframes)
        audio = self.t.fetch_audio_all()                     # [slots][frames], one device-to-host copy
        if self.slots is None:
            self.slots = [self.t.slot(ch) for ch in self.ch]
        return audio[self.slots, : nframes // self.d]

    def close(self):
        self.t.destroy()
