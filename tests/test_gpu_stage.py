"""-m gpu: sparse staging (wr_stage_windows_from_host, r04): of a host block only the frames the tuner's taps reach --
[k * period - (length - 1), k * period] for every output frame k -- and the block's tail cross PCIe; they land at their own
positions of the staged float block, bit for bit what the whole-block paths put there, and a tuner fed from such a block
gives the very audio it gives from the whole block."""
import ctypes as C

import numpy as np
import pytest

from webradio_amd import capi, synth
from webradio_amd.device import Tuner

pytestmark = pytest.mark.gpu


def _needed(nframes, period, length, tail):
    need = np.zeros(nframes, bool)
    for k in range((nframes - 1) // period + 1):          # the output frames of THIS block: their windows end inside it
        lo, hi = max(0, k * period - (length - 1)), min(nframes - 1, k * period)
        if lo <= hi:
            need[lo:hi + 1] = True
    need[max(0, nframes - tail):] = True
    return need


@pytest.mark.parametrize("u8", [True, False], ids=["u8", "f32"])
@pytest.mark.parametrize("nframes,period,length,tail", [(40_000, 400, 64, 63), (100_000, 4000, 64, 1200), (8_192, 130, 64, 64),
                                                       (20_000, 400, 32, 700), (30_001, 333, 64, 100), (1_000, 2_000, 64, 63),
                                                       (640, 64, 64, 640), (60_000, 1_000, 256, 255), (50_000, 900, 128, 300)])
def test_windows_and_tail_land_where_the_whole_block_puts_them(dev, page_locked, u8, nframes, period, length, tail):
    import torch
    rng = np.random.default_rng(nframes + period)
    if u8:
        host = page_locked(2 * nframes, np.uint8)
        host[:] = rng.integers(0, 256, 2 * nframes, dtype=np.uint8)
        want = ((host.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)).astype(np.float32)
    else:
        host = page_locked(2 * nframes, np.float32)
        host[:] = rng.standard_normal(2 * nframes).astype(np.float32)
        want = host
    assert host.ctypes.data % 16 == 0
    out = torch.full((2 * nframes,), float("nan"), device="cuda")
    assert dev.lib.wr_stage_windows_from_host(dev.h, host.ctypes.data_as(C.c_void_p), int(u8), capi.ptr(out), nframes, period,
                                              length, tail) == 0, dev.lib.wr_last_error()
    assert dev.lib.wr_dev_wait_uploads(dev.h) == 0
    got = out.cpu().numpy().reshape(-1, 2)
    need = _needed(nframes, period, length, tail)
    w = want.reshape(-1, 2)
    assert np.array_equal(got[need].view(np.uint32), w[need].view(np.uint32))            # everything the taps reach: exact
    staged = ~np.isnan(got[:, 0])
    assert np.array_equal(got[staged].view(np.uint32), w[staged].view(np.uint32))        # and nothing staged is wrong
    if period >= 4 * length and nframes >= 8 * period:
        assert staged.mean() < 0.5                                                       # it IS sparse


def test_argument_checks(dev):
    import torch
    out = torch.zeros(1024, device="cuda")
    host = np.zeros(1024, np.float32)                          # not page-locked
    lib = dev.lib
    assert lib.wr_stage_windows_from_host(dev.h, host.ctypes.data_as(C.c_void_p), 0, capi.ptr(out), 512, 100, 64, 63) == capi.WR_ERR_ARG
    assert b"page-locked" in lib.wr_last_error()
    assert lib.wr_stage_windows_from_host(None, None, 0, None, 0, 100, 64, 0) == capi.WR_ERR_ARG
    assert lib.wr_stage_windows_from_host(dev.h, host.ctypes.data_as(C.c_void_p), 0, capi.ptr(out), 512, 0, 64, 63) == capi.WR_ERR_ARG
    assert lib.wr_stage_windows_from_host(dev.h, host.ctypes.data_as(C.c_void_p), 0, capi.ptr(out), 512, 100, 5000, 63) == capi.WR_ERR_ARG


@pytest.mark.parametrize("u8", [True, False], ids=["u8", "f32"])
def test_tuner_fed_from_a_sparsely_staged_block(dev, page_locked, u8):
    """Three consecutive blocks at C2's ratios (D1 = 400, 64 taps): the audio of a tuner whose blocks were staged sparsely is
    the audio of the same tuner fed the whole blocks -- the kernel reads nothing the sparse stage left out."""
    import torch
    fs, n = 2_000_000, 80_000
    ifs = [(-4 + c) * 6250 + 1234 for c in range(8)]
    iq = synth.fm_stream(3 * n, fs, ifs[::2], fm_base=30.0, beta=2.0)
    if u8:
        raw = np.clip(np.round(127.5 + 127.0 * iq), 0, 255).astype(np.uint8)
        full = ((raw.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)).astype(np.float32)
    else:
        raw, full = iq, iq
    src = page_locked(raw.size, raw.dtype)
    src[:] = raw
    raw = src
    want, got = [], []
    for sparse in (False, True):
        t = Tuner(dev, fs, 8, n, capi.WR_NCO_ROTATE)
        chans = [t.add_receiver(f, 128_000, 5_000, capi.WR_FM, 160, 1_000) for f in ifs]
        rows = []
        stage = torch.full((2 * n,), float("nan"), device="cuda")
        for b in range(3):
            if sparse:
                piece = raw[2 * n * b: 2 * n * (b + 1)]
                assert dev.lib.wr_stage_windows_from_host(dev.h, piece.ctypes.data_as(C.c_void_p), int(u8), capi.ptr(stage), n,
                                                          400, 64, 63) == 0, dev.lib.wr_last_error()
                t.submit_device(stage, n)
            else:
                t.submit_host(full[2 * n * b: 2 * n * (b + 1)])
            rows.append(np.stack([t.fetch(ch, capi.WR_STAGE_AUDIO, n) for ch in chans]))
            stage.fill_(float("nan"))
        t.destroy()
        (got if sparse else want).append(np.concatenate(rows, axis=1))
    dev.lib.wr_dev_wait_uploads(dev.h)
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert np.isfinite(got[0]).all() and np.abs(got[0]).max() > 1e-3


@pytest.mark.parametrize("u8", [True, False], ids=["u8", "f32"])
def test_host_submit_out_of_page_locked_memory_is_staged_sparsely(dev, page_locked, u8):
    """wr_tuner_submit / wr_tuner_submit_u8 with WR_HOST: a block in page-locked memory goes through the sparse staging kernel
    (D1 = 400, 64 taps), one in pageable memory through the copy -- the same audio bit for bit, block after block."""
    fs, n = 2_000_000, 80_000
    ifs = [(-4 + c) * 6250 + 1234 for c in range(8)]
    iq = synth.fm_stream(3 * n, fs, ifs[::2], fm_base=30.0, beta=2.0)
    if u8:
        data = np.clip(np.round(127.5 + 127.0 * iq), 0, 255).astype(np.uint8)
    else:
        data = iq
    locked = page_locked(data.size, data.dtype)
    locked[:] = data
    outs = []
    for src in (data, locked):
        t = Tuner(dev, fs, 8, n, capi.WR_NCO_ROTATE)
        chans = [t.add_receiver(f, 128_000, 5_000, capi.WR_FM, 160, 1_000) for f in ifs]
        rows = []
        for b in range(3):
            blk = src[2 * n * b: 2 * n * (b + 1)]
            if u8:
                capi.check(dev.lib.wr_tuner_submit_u8(t.h, blk.ctypes.data_as(C.c_void_p), n, capi.WR_HOST))
            else:
                capi.check(dev.lib.wr_tuner_submit(t.h, blk.ctypes.data_as(C.c_void_p), n, capi.WR_HOST))
            dev.sync()
            how = C.c_int()
            capi.check(dev.lib.wr_tuner_last_staging(t.h, C.byref(how)))
            assert how.value == (2 if src is locked else 1)          # sparsely staged / copied whole
            rows.append(np.stack([t.fetch(ch, capi.WR_STAGE_AUDIO, n) for ch in chans]))
        t.destroy()
        outs.append(np.concatenate(rows, axis=1))
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.isfinite(outs[1]).all() and np.abs(outs[1]).max() > 1e-3


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("WR_FUZZ_SEEDS", "16"))))
def test_random_window_shapes(dev, page_locked, seed):
    """Seeded fuzz of the staging kernel: random block lengths (ragged against the period), periods, window lengths (longer
    than a wave's 64 chunks too), tails (longer than the block too) and both source formats -- every frame the taps reach
    and the tail are there bit for bit, nothing staged is wrong."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    u8 = bool(rng.integers(0, 2))
    length = int(rng.choice([2, 16, 63, 64, 65, 128, 256, 500, 1024]))
    period = int(rng.integers(1, 6)) * length + int(rng.integers(0, 700))
    nframes = int(rng.integers(1, 40)) * period + int(rng.integers(0, period)) + 8
    tail = int(rng.choice([0, 1, length - 1, length + 17, nframes, nframes + 5]))
    if u8:
        host = page_locked(2 * nframes, np.uint8)
        host[:] = rng.integers(0, 256, 2 * nframes, dtype=np.uint8)
        want = ((host.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)).astype(np.float32)
    else:
        host = page_locked(2 * nframes, np.float32)
        host[:] = rng.standard_normal(2 * nframes).astype(np.float32)
        want = host
    out = torch.full((2 * nframes,), float("nan"), device="cuda")
    assert dev.lib.wr_stage_windows_from_host(dev.h, host.ctypes.data_as(C.c_void_p), int(u8), capi.ptr(out), nframes, period,
                                              length, tail) == 0, dev.lib.wr_last_error()
    assert dev.lib.wr_dev_wait_uploads(dev.h) == 0
    got = out.cpu().numpy().reshape(-1, 2)
    need = _needed(nframes, period, length, min(tail, nframes))
    w = want.reshape(-1, 2)
    what = (u8, nframes, period, length, tail)
    assert np.array_equal(got[need].view(np.uint32), w[need].view(np.uint32)), what
    staged = ~np.isnan(got[:, 0])
    assert np.array_equal(got[staged].view(np.uint32), w[staged].view(np.uint32)), what


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("WR_FUZZ_SEEDS", "8"))))
def test_random_host_submits_sparse_against_copied(dev, page_locked, seed):
    """Seeded fuzz of the C ABI's own sparse staging: random rates (sparse and dense windows), channel-filter lengths (64, 128),
    receiver counts, ragged block lengths and both source formats; every block submitted once from pageable memory (copied
    whole) and once from page-locked memory (staged sparsely where the windows allow) -- the same audio bit for bit, and
    wr_tuner_last_staging says which way each block went."""
    rng = np.random.default_rng(9000 + seed)
    u8 = bool(rng.integers(0, 2))
    fs = 2_000_000
    d1 = int(rng.choice([8, 100, 125, 200, 400, 1000]))
    l1 = int(rng.choice([64, 64, 128]))
    d2 = int(rng.choice([4, 5]))
    nrx = int(rng.choice([1, 7, 64, 70]))
    ifs = [int(rng.integers(-900_000, 900_000)) for _ in range(nrx)]
    crate = fs // d1
    if fs % d1 or crate % d2:
        pytest.skip("rates not integer related for this draw")
    blocks = [int(rng.integers(3, 12)) * d1 * d2 + int(rng.choice([0, 0, 1, d1 - 1, d1 * d2 - 1])) for _ in range(3)]
    nmax = max(blocks)
    iq = synth.fm_stream(sum(blocks), fs, ifs[:2], fm_base=30.0, beta=2.0, seed=seed)
    data = np.clip(np.round(127.5 + 127.0 * iq), 0, 255).astype(np.uint8) if u8 else iq
    locked = page_locked(data.size, data.dtype)
    locked[:] = data
    cpb = max(oracle_passband(fs), 64_000)
    outs, ways = [], []
    for src in (data, locked):
        t = Tuner(dev, fs, nrx, nmax, capi.WR_NCO_ROTATE)
        chans = [t.add_receiver(f, cpb, crate, capi.WR_FM, max(crate // 32, 1), crate // d2, fir_lengths=(l1, 64)) for f in ifs]
        rows, pos, way, expect = [], 0, [], []
        for n in blocks:
            blk = src[2 * pos: 2 * (pos + n)]
            # sparsely: page-locked, a window at most every second filter length, a few of them in the block, 16-byte aligned
            expect.append(2 if (src is locked and d1 >= 2 * l1 and n >= 4 * d1 and blk.ctypes.data % 16 == 0) else 1)
            fn = dev.lib.wr_tuner_submit_u8 if u8 else dev.lib.wr_tuner_submit
            capi.check(fn(t.h, blk.ctypes.data_as(C.c_void_p), n, capi.WR_HOST))
            dev.sync()
            how = C.c_int()
            capi.check(dev.lib.wr_tuner_last_staging(t.h, C.byref(how)))
            way.append(how.value)
            rows.append(np.stack([t.fetch(ch, capi.WR_STAGE_AUDIO, n) for ch in chans]))
            pos += n
        t.destroy()
        outs.append(np.concatenate(rows, axis=1))
        ways.append(way)
        assert way == expect, (way, expect, d1, l1, blocks)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (u8, d1, l1, d2, nrx, blocks, ways)


def oracle_passband(fs):
    return fs // 16
