"""Multi-GPU sharding logic on CPU with the gloo backend, world_size 2.

The DSP of each chunk is done by the ORACLE here (tests may use it); what is under test is
the sharding itself: chunk dealing, the ring halo exchange, the closed-form phase at a chunk
start, the number of discarded audio frames -- the time-sharded result must be bit-identical
to the sequential one.  The same driver runs on GPUs with the product's TunerShard
(tests/test_gpu_timeshard.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = dict(fs=2_000_000, d1=400, d2=5, cpb=128_000, crate=5_000, apb=160, arate=1_000,
           ifs=[50_000, -75_000, 1234], T=60_000, nchunks=5)


def _stream(total):
    from webradio_amd import synth
    return synth.fm_stream(total, CFG["fs"], CFG["ifs"][:2], amp=0.3, fm_base=30.0, beta=2.0)


def _oracle_process_factory(oracle, mode):
    def process(block, start_frame, nframes):
        from webradio_amd import timeshard
        rows = []
        for f in CFG["ifs"]:
            rx = oracle.Receiver(CFG["fs"], f, CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
            rx.s.phase = timeshard.phase_at(oracle.phase_step(f, CFG["fs"]), start_frame)
            rows.append(rx.run(np.asarray(block))[0])
        return np.stack(rows)
    return process


def _worker(rank, world, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import wr_oracle as oracle
    from webradio_amd import timeshard
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _proc
    dist = _proc.init_gloo(rank, world, os.path.join(outdir, "rdzv"))      # (one GPU on the test box: a gloo ring)
    T, n = CFG["T"], CFG["nchunks"]
    iq = _stream(T * n)
    ring = timeshard.RingHalo(dist, rank, world)
    out = timeshard.run_time_sharded(
        ring, lambda c: iq[2 * c * T: 2 * (c + 1) * T], n, T, CFG["d1"], CFG["d2"],
        _oracle_process_factory(oracle, oracle.FM),
        to_tensor=lambda a: torch.from_numpy(np.ascontiguousarray(a)),
        from_tensor=lambda t: t.numpy())
    assert sorted(out) == timeshard_chunks(n, rank, world)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{str(c): a for c, a in out.items()})
    # the bench's max-over-ranks reduction
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()


def timeshard_chunks(n, rank, world):
    return [c for c in range(n) if c % world == rank]


def test_halo_size_and_discard():
    from webradio_amd import timeshard
    # (63+1)*D1 + 63 frames, rounded up to whole audio frames
    assert timeshard.halo_frames(400, 5) == 26_000 and timeshard.discarded_audio_frames(400, 5) == 13
    assert timeshard.halo_frames(4000, 5) == 260_000        # C5
    assert timeshard.halo_frames(8, 8) == 576 and timeshard.discarded_audio_frames(8, 8) == 9
    assert timeshard.tuners_for_rank(8, 3, 8) == [3] and timeshard.tuners_for_rank(8, 1, 2) == [1, 3, 5, 7]
    assert timeshard.phase_at(-1, 3) == (1 << 31) - 3


def test_single_rank_time_shard_equals_sequential(oracle):
    from webradio_amd import timeshard
    T, n = CFG["T"], CFG["nchunks"]
    iq = _stream(T * n)

    class Solo:
        rank, world = 0, 1
        def exchange(self, tail):
            return None
    out = timeshard.run_time_sharded(Solo(), lambda c: iq[2 * c * T: 2 * (c + 1) * T], n, T, CFG["d1"], CFG["d2"],
                                     _oracle_process_factory(oracle, oracle.FM), lambda a: a, lambda a: a)
    got = np.concatenate([out[c] for c in range(n)], axis=1)
    want = _oracle_process_factory(oracle, oracle.FM)(iq, 0, T * n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("world", [2, 3, 8], ids=["world2", "world3-ragged", "world8-a-node"])
def test_ranks_gloo_ring_halo(tmp_path, oracle, world):
    """The N > 1 path of BASELINE config 5 on CPU ranks over gloo: chunks dealt round-robin, every halo from the ring
    neighbour, the max-over-ranks reduction -- at 2 ranks, at 3 (the chunks do not divide among the ranks: idle ranks still
    take part in the ring) and at 8, the node's rank count (more ranks than chunks but one: most of them idle in round 2)."""
    import _proc
    _proc.spawn_ranks(_worker, world, (world, str(tmp_path)), timeout=170)
    T, n = CFG["T"], CFG["nchunks"]
    parts = {}
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        for k in d.files:
            parts[int(k)] = d[k]
    assert sorted(parts) == list(range(n))
    got = np.concatenate([parts[c] for c in range(n)], axis=1)
    want = _oracle_process_factory(oracle, oracle.FM)(_stream(T * n), 0, T * n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
