"""-m gpu: the C++ host runtime (webradio_amd/host) end to end, driven like main.cxx drives
the reference: FrontEnd + Receivers + Radio::run().  Checked against the oracle for
  - the fused path (every Receiver of the tuner in one launch sequence),
  - the one-kernel-per-block path (WEBRADIO_NO_FUSION=1), which is bit-exact,
  - the REFERENCE's radio.cxx linked against our classes (oracle/_ref/libwr_boundary.so)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import _proc

from webradio_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "webradio_amd", "host")
CXXT = os.path.join(ROOT, "tests", "cxx")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)

RUNNER = r'''
import ctypes as C, sys, numpy as np, json
lib, npz = sys.argv[1], sys.argv[2]
import torch  # same HIP runtime as the rest of the suite
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz)
iq = np.ascontiguousarray(d["iq"], np.float32); ifs = np.ascontiguousarray(d["ifs"], np.int32)
modes = np.ascontiguousarray(d["modes"], np.int32); p = d["params"]
nrx = ifs.size; cap = int(p[8]); fft = int(p[9])
audio = np.zeros((nrx, cap), np.float32); n = C.c_size_t(); spec = np.zeros(max(fft, 1), np.float32)
fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
L.wr_host_run.argtypes = [fp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, ip, ip, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                          C.c_int, C.c_int, fp, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint, fp]
rc = L.wr_host_run(iq.ctypes.data_as(fp), iq.size // 2, int(p[0]), int(p[1]), nrx, ifs.ctypes.data_as(ip),
                   modes.ctypes.data_as(ip), int(p[2]), int(p[3]), int(p[4]), int(p[5]), int(p[6]), int(p[7]),
                   audio.ctypes.data_as(fp), cap, C.byref(n), fft, spec.ctypes.data_as(fp))
L.wr_block_kernel_calls.restype = C.c_ulonglong     # (libwebradio_amd, a dependency of the harness)
sl = sb = 0
if hasattr(L, "wr_host_stream_blocks"):
    L.wr_host_stream_blocks.restype = L.wr_host_stream_launches.restype = C.c_ulonglong
    sl, sb = int(L.wr_host_stream_launches()), int(L.wr_host_stream_blocks())
np.savez(sys.argv[3], rc=rc, audio=audio[:, :n.value], spec=spec, left=L.wr_host_registry_sizes(),
         traced=L.wr_host_trace_count(), block_calls=int(L.wr_block_kernel_calls()), stream_launches=sl, stream_blocks=sb)
'''


def _run(libname, tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate, retune=(-1, 0), fft=0, env=None):
    lib = os.path.join(CXXT, libname) if not os.path.isabs(libname) else libname
    if not os.path.exists(lib):
        _proc.run(["make", "-s", "-C", CXXT, "all"], timeout=600)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    nblocks = (iq.size // 2) // block
    cap = nblocks * (block // (rate // crate) // (crate // arate)) + 16
    np.savez(inp, iq=iq, ifs=np.array(ifs, np.int32), modes=np.array(modes, np.int32),
             params=np.array([rate, block, cpb, crate, apb, arate, retune[0], retune[1], cap, fft], np.int64))
    # (the tests' blocks are small; the runtime page-locks a source block only from 1 MB on: here from the first byte, so that
    # the staging paths a 100 Msps tuner takes -- DMA from page-locked memory, sparse staging -- are the ones run)
    e = dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0")
    e.update(env or {})
    # a fresh process per run: the host runtime keeps per-process device contexts and env switches
    _proc.run([sys.executable, "-c", RUNNER, lib, inp, out], env=e)
    r = np.load(out)
    assert int(r["rc"]) == 0
    assert int(r["left"]) == 0                    # registries empty again (radio.cxx:98,143)
    _run.last_stream = (int(r["stream_launches"]), int(r["stream_blocks"]))
    return r["audio"], r["spec"]


def _oracle(oracle, iq, rate, block, ifs, modes, cpb, crate, apb, arate, retune=(-1, 0)):
    outs = []
    for c, (f, m) in enumerate(zip(ifs, modes)):
        rx = oracle.Receiver(rate, f, cpb, crate, m, apb, arate)
        a = []
        for b in range((iq.size // 2) // block):
            if c == 0 and b == retune[0]:
                rx.set_if(retune[1])
            a.append(rx.run(iq[2 * b * block: 2 * (b + 1) * block])[0])
        outs.append(np.concatenate(a))
    return np.stack(outs)


CFG = dict(rate=2_000_000, block=40_000, cpb=128_000, crate=5_000, apb=160, arate=1_000)


def test_fused_receivers_match_oracle(tmp_path, oracle):
    ifs = [(-5 + c) * 6250 + 1234 for c in range(10)]
    modes = [1, 0, 2, 3, 1, 1, 0, 1, 2, 1]            # FM AM USB LSB ...
    iq = synth.fm_stream(4 * CFG["block"], CFG["rate"], ifs[::2], fm_base=30.0, beta=2.0)
    got, _ = _run("libwr_host_pipeline.so", tmp_path, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"],
                  CFG["crate"], CFG["apb"], CFG["arate"])
    want = _oracle(oracle, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"])
    assert got.shape == want.shape
    for c in range(0, len(ifs), 2):                   # carrier channels
        assert np.abs(got[c] - want[c]).max() <= 1e-5, c
    for c, m in enumerate(modes):                     # linear detectors: every channel
        if m != 1:
            assert np.abs(got[c] - want[c]).max() <= 2e-6, c


@pytest.mark.parametrize("block", [40_000, 41_000, 40_300], ids=["whole-audio-frames", "ragged-1000", "ragged-300"])
def test_sparse_staging_and_parts_give_the_same_bits(tmp_path, block):
    """r04: how a source block gets to the GPU and how many parts it goes through in changes no bit of what the sinks
    receive.  WEBRADIO_SPARSE=1 (default; only the frames under the taps cross PCIe: D1 = 400 here, 64 taps) against the
    whole block staged; WEBRADIO_PIECES parts (the audio of each put straight into the audio filters' output vectors)
    against one; the SpectrumSink beside them, whose frame then comes from a block of which only the tail was staged."""
    ifs = [(-5 + c) * 6250 + 1234 for c in range(10)]
    modes = [1, 0, 2, 3, 1, 1, 0, 1, 2, 1]
    # (a block that is not a whole number of channel or audio frames: the reference truncates per block, dspblock.cxx:177-178;
    # such a block goes through in one part, and its windows and tail are staged all the same)
    iq = synth.fm_stream(4 * block, CFG["rate"], ifs[::2], fm_base=30.0, beta=2.0)
    runs = {}
    for name, env in (("whole-1", {"WEBRADIO_SPARSE": "0", "WEBRADIO_PIECES": "1"}),
                      ("sparse-1", {"WEBRADIO_SPARSE": "1", "WEBRADIO_PIECES": "1"}),
                      ("sparse-4", {"WEBRADIO_SPARSE": "1", "WEBRADIO_PIECES": "4", "WEBRADIO_PIECE_MIN_FRAMES": "1000"}),
                      ("whole-4", {"WEBRADIO_SPARSE": "0", "WEBRADIO_PIECES": "4", "WEBRADIO_PIECE_MIN_FRAMES": "1000"}),
                      ("sparse-default", {})):
        runs[name] = _run("libwr_host_pipeline.so", tmp_path, iq, CFG["rate"], block, ifs, modes, CFG["cpb"],
                          CFG["crate"], CFG["apb"], CFG["arate"], fft=512, env=env)
    base_audio, base_spec = runs["whole-1"]
    assert np.abs(base_audio).max() > 1e-3 and np.isfinite(base_spec).all()
    for name, (audio, spec) in runs.items():
        assert np.array_equal(audio.view(np.uint32), base_audio.view(np.uint32)), name
        assert np.array_equal(spec.view(np.uint32), base_spec.view(np.uint32)), name


def test_runtime_fir_length_through_the_host_classes(tmp_path, oracle):
    """LowPass::setFirLength(512) on both filters of every Receiver (the reference's
    FIXME at lowpass.cxx:38-39): longer than the tuner takes (WR_FIR_FUSED_MAX = 256), such chains run
    block by block, bit-identical for the linear detectors to the reference's algorithm with
    _firLength = 512 (r05: 128 and 256 stay in the batch, the next test)."""
    L = 512
    ifs, modes = [50_000, -75_000, 10], [0, 2, 3]
    rate, block, cpb, crate, apb, arate = CFG["rate"], CFG["block"], CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"]
    iq = synth.fm_stream(3 * block, rate, ifs[:2], amp=0.3)
    got, _ = _run("libwr_host_pipeline.so", tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate,
                  env={"WR_TEST_FIR_LENGTH": str(L), "WEBRADIO_TRACE": "1"})
    assert int(np.load(str(tmp_path / "out.npz"))["traced"]) == 0          # nothing fused
    table = oracle.sin_table()
    for c, (f, m) in enumerate(zip(ifs, modes)):
        f1 = oracle.Fir(2, rate // crate, oracle.lowpass_design(cpb, rate, L))
        f2 = oracle.Fir(1, crate // arate, oracle.lowpass_design(apb, crate, L))
        phase, prev, want = 0, (0.0, 0.0), []
        for b in range(3):
            mixed, phase = oracle.mix(table, phase, oracle.phase_step(f, rate), iq[2 * b * block: 2 * (b + 1) * block])
            d, prev = oracle.demod(m, prev, f1.process(mixed))
            want.append(f2.process(d))
        want = np.concatenate(want)
        assert got[c].size == want.size
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), c
    # and it is not the 64-tap result
    base, _ = _run("libwr_host_pipeline.so", tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate,
                   env={"WEBRADIO_NO_FUSION": "1"})
    assert np.abs(base - got).max() > 1e-4


@pytest.mark.parametrize("which,L", [("audio", 128), ("audio", 256), ("both", 128), ("both", 256)])
def test_long_audio_filter_stays_in_the_tuner_batch(tmp_path, oracle, which, L):
    """r05: audioFilter()->setFirLength(128 | 256), alone or together with the channel filter:
    the Receiver stays in the source's tuner batch (WEBRADIO_TRACE shows it submitting, and the library ran no
    stand-alone block kernel: `block_calls` in the harness's output).  WEBRADIO_NCO=exact: the oracle's bits for the
    linear detectors; the default mode within the ROTATE tolerance through the long filter."""
    ifs, modes = [50_000, -75_000, 10, 33_333], [0, 2, 3, 0]
    rate, block, cpb, crate, apb, arate = CFG["rate"], CFG["block"], CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"]
    iq = synth.fm_stream(4 * block, rate, ifs[:2], amp=0.3)
    var = {"audio": {"WR_TEST_FIR_LENGTH_AUDIO": str(L)}, "both": {"WR_TEST_FIR_LENGTH": str(L)}}[which]
    got, _ = _run("libwr_host_pipeline.so", tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate, retune=(2, 61_000),
                  env=dict(var, WEBRADIO_TRACE="1", WEBRADIO_NCO="exact"))
    z = np.load(str(tmp_path / "out.npz"))
    assert int(z["traced"]) > 0 and int(z["block_calls"]) == 0
    fast, _ = _run("libwr_host_pipeline.so", tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate, retune=(2, 61_000),
                   env=dict(var, WEBRADIO_TRACE="1"))
    z = np.load(str(tmp_path / "out.npz"))
    assert int(z["traced"]) > 0 and int(z["block_calls"]) == 0
    table = oracle.sin_table()
    L1 = L if which == "both" else 64
    gain2 = max(1.0, float(np.abs(oracle.lowpass_design(apb, crate, L)).sum()))
    for c, (f, m) in enumerate(zip(ifs, modes)):
        f1 = oracle.Fir(2, rate // crate, oracle.lowpass_design(cpb, rate, L1))
        f2 = oracle.Fir(1, crate // arate, oracle.lowpass_design(apb, crate, L))
        phase, prev, want = 0, (0.0, 0.0), []
        for b in range(4):
            if c == 0 and b == 2:
                f = 61_000
            mixed, phase = oracle.mix(table, phase, oracle.phase_step(f, rate), iq[2 * b * block: 2 * (b + 1) * block])
            d, prev = oracle.demod(m, prev, f1.process(mixed))
            want.append(f2.process(d))
        want = np.concatenate(want)
        assert got[c].size == want.size
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), c
        assert np.abs(fast[c] - want).max() <= 4e-6 * gain2, (c, float(np.abs(fast[c] - want).max()))


@pytest.mark.parametrize("L", [128, 256])
def test_long_channel_filter_stays_in_the_tuner_batch(tmp_path, oracle, L):
    """r03: channelFilter()->setFirLength(128 | 256) with the audio filter at its 64 taps: the Receiver stays in the
    source's tuner batch (WEBRADIO_TRACE shows it submitting).  WEBRADIO_NCO=exact: the reference's own arithmetic
    (k_tuner_ddc_long), the linear detectors are the oracle's bits; the default mode: L / 64 segments of the ROTATE
    recurrence (k_tuner_ddc_long_rot), within its tolerance.  A retune between blocks 1 and 2."""
    ifs, modes = [50_000, -75_000, 10, 33_333], [0, 2, 3, 0]
    rate, block, cpb, crate, apb, arate = CFG["rate"], CFG["block"], CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"]
    iq = synth.fm_stream(4 * block, rate, ifs[:2], amp=0.3)
    got, _ = _run("libwr_host_pipeline.so", tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate, retune=(2, 61_000),
                  env={"WR_TEST_FIR_LENGTH_CHAN": str(L), "WEBRADIO_TRACE": "1", "WEBRADIO_NCO": "exact"})
    assert int(np.load(str(tmp_path / "out.npz"))["traced"]) > 0
    fast, _ = _run("libwr_host_pipeline.so", tmp_path, iq, rate, block, ifs, modes, cpb, crate, apb, arate, retune=(2, 61_000),
                   env={"WR_TEST_FIR_LENGTH_CHAN": str(L), "WEBRADIO_TRACE": "1"})
    assert int(np.load(str(tmp_path / "out.npz"))["traced"]) > 0
    table = oracle.sin_table()
    for c, (f, m) in enumerate(zip(ifs, modes)):
        f1 = oracle.Fir(2, rate // crate, oracle.lowpass_design(cpb, rate, L))
        f2 = oracle.Fir(1, crate // arate, oracle.lowpass_design(apb, crate))
        phase, prev, want = 0, (0.0, 0.0), []
        for b in range(4):
            if c == 0 and b == 2:
                f = 61_000
            mixed, phase = oracle.mix(table, phase, oracle.phase_step(f, rate), iq[2 * b * block: 2 * (b + 1) * block])
            d, prev = oracle.demod(m, prev, f1.process(mixed))
            want.append(f2.process(d))
        want = np.concatenate(want)
        assert got[c].size == want.size
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), c
        assert np.abs(fast[c] - want).max() <= 4e-6, (c, float(np.abs(fast[c] - want).max()))   # USB / LSB add two components


def test_unfused_blocks_are_bit_exact(tmp_path, oracle):
    """WEBRADIO_NO_FUSION=1: each block runs its own kernel on host vectors, exactly the
    reference's dataflow -- AM/USB/LSB chains are bit-identical."""
    ifs, modes = [50_000, -75_000, 10], [0, 2, 3]
    iq = synth.fm_stream(3 * CFG["block"], CFG["rate"], ifs[:2], amp=0.3)
    got, _ = _run("libwr_host_pipeline.so", tmp_path, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"],
                  CFG["crate"], CFG["apb"], CFG["arate"], retune=(1, 12_345), env={"WEBRADIO_NO_FUSION": "1"})
    want = _oracle(oracle, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"], CFG["crate"], CFG["apb"],
                   CFG["arate"], retune=(1, 12_345))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_fused_exact_nco_and_retune(tmp_path, oracle):
    ifs, modes = [50_000, -75_000], [0, 3]
    iq = synth.fm_stream(3 * CFG["block"], CFG["rate"], ifs, amp=0.3)
    got, _ = _run("libwr_host_pipeline.so", tmp_path, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"],
                  CFG["crate"], CFG["apb"], CFG["arate"], retune=(2, -40_000), env={"WEBRADIO_NCO_EXACT": "1"})
    want = _oracle(oracle, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"], CFG["crate"], CFG["apb"],
                   CFG["arate"], retune=(2, -40_000))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_reference_radio_cxx_on_our_blocks(tmp_path, oracle):
    """The reference's own radio.cxx (compiled unchanged) wiring our GPU blocks."""
    lib = os.path.join(ROOT, "oracle", "_ref", "libwr_boundary.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref/libwr_boundary.so not built (needs /root/reference at build time)")
    ifs, modes = [50_000, -75_000, 1234, -9999], [0, 1, 2, 1]
    iq = synth.fm_stream(3 * CFG["block"], CFG["rate"], ifs, amp=0.2, fm_base=30.0, beta=2.0)
    got, spec = _run(lib, tmp_path, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"], CFG["crate"], CFG["apb"],
                     CFG["arate"], fft=1024)
    want = _oracle(oracle, iq, CFG["rate"], CFG["block"], ifs, modes, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"])
    assert np.abs(got - want).max() <= 1e-5
    # FrontEnd::spectrum(): last complete 1024-frame of the stream
    o = oracle.Spectrum(1024)
    o.process(iq)
    w = o.get()
    strong = w >= w.max() - 60
    assert np.abs(spec - w)[strong].max() <= 0.02


RESTART_RUNNER = r'''
import ctypes as C, sys, numpy as np
lib, npz = sys.argv[1], sys.argv[2]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz)
iq = np.ascontiguousarray(d["iq"], np.float32); ifs = np.ascontiguousarray(d["ifs"], np.int32); p = d["params"]
nrx = ifs.size; cap = int(p[8])
audio = np.zeros((2 * nrx, cap), np.float32); n = C.c_size_t()
fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
L.wr_host_run_restart.argtypes = [fp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, ip, C.c_int, C.c_uint, C.c_uint, C.c_uint,
                                  C.c_uint, C.c_uint, fp, C.c_size_t, C.POINTER(C.c_size_t)]
rc = L.wr_host_run_restart(iq.ctypes.data_as(fp), iq.size // 2, int(p[0]), int(p[1]), nrx, ifs.ctypes.data_as(ip), int(p[6]),
                           int(p[2]), int(p[3]), int(p[4]), int(p[5]), int(p[7]), audio.ctypes.data_as(fp), cap, C.byref(n))
np.savez(sys.argv[3], rc=rc, audio=audio[:, :n.value], left=L.wr_host_registry_sizes())
'''


def test_stop_start_keeps_phase_and_two_front_ends(tmp_path, oracle):
    """stop()/start() mid-stream: both LowPass histories restart empty (lowpass.cxx:118-129),
    DownConverter::phase and Demodulator::prev_i/q carry on (quirk Q5); two FrontEnds, each
    with its own TunerBatch, run side by side on different parts of the recording."""
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    ifs, mode = [50_000, -75_000, 4321], 3                        # LSB: linear, tight tolerance
    rate, block = CFG["rate"], CFG["block"]
    nblk, restart_at = 5, 2
    iq = synth.fm_stream((nblk + 1) * block, rate, ifs, amp=0.2)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    cap = nblk * (block // 2000) + 16
    np.savez(inp, iq=iq, ifs=np.array(ifs, np.int32),
             params=np.array([rate, block, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"], mode, restart_at, cap], np.int64))
    _proc.run([sys.executable, "-c", RESTART_RUNNER, lib, inp, out],
                          env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0", WEBRADIO_NCO_EXACT="1"))
    r = np.load(out)
    assert int(r["rc"]) == 0 and int(r["left"]) == 0
    got = r["audio"]
    for t in range(2):
        for c, f in enumerate(ifs):
            rx = oracle.Receiver(rate, f, CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
            a = []
            for b in range(nblk):
                if b == restart_at:                               # fresh filters, same NCO phase and prev_i/q
                    nx = oracle.Receiver(rate, f, CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
                    nx.s.phase, nx.s.prev_i, nx.s.prev_q = rx.s.phase, rx.s.prev_i, rx.s.prev_q
                    rx = nx
                a.append(rx.run(iq[2 * (b + t) * block: 2 * (b + t + 1) * block])[0])
            want = np.concatenate(a)
            assert np.array_equal(got[t * len(ifs) + c].view(np.uint32), want.view(np.uint32)), (t, c)


RERATE_RUNNER = r'''
import ctypes as C, sys, numpy as np
lib, npz = sys.argv[1], sys.argv[2]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz)
iq = np.ascontiguousarray(d["iq"], np.float32); ifs = np.ascontiguousarray(d["ifs"], np.int32); p = [int(v) for v in d["params"]]
nrx = ifs.size; cap = p[11]
audio = np.zeros((2 * nrx, cap), np.float32); n1, n2 = C.c_size_t(), C.c_size_t()
fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
L.wr_host_run_rerate.argtypes = [fp, C.c_size_t] + [C.c_uint] * 7 + [ip, C.c_int] + [C.c_uint] * 4 + [fp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
rc = L.wr_host_run_rerate(iq.ctypes.data_as(fp), iq.size // 2, p[0], p[1], p[2], p[3], p[4], p[5], nrx, ifs.ctypes.data_as(ip), p[6],
                          p[7], p[8], p[9], p[10], audio.ctypes.data_as(fp), cap, C.byref(n1), C.byref(n2))
np.savez(sys.argv[3], rc=rc, a1=audio[:nrx, :n1.value], a2=audio[nrx:, :n2.value], left=L.wr_host_registry_sizes())
'''


def test_stop_set_rate_and_block_size_start(tmp_path, oracle):
    """the source's wr_tuner used to survive stop()/start() with the OLD input rate and
    block size.  stop -> setSampleRate(1 M instead of 2 M) + setBlockSize(60 000 instead of 40 000
    frames) -> start: the NCO steps follow the new rate (downconverter.cxx:80), the larger block is
    accepted, phase and prev_i/q carry over (Q5), the filters restart empty."""
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    ifs, mode = [50_000, -75_000, 4321], 3
    r1, b1, r2, b2, n1, n2 = 2_000_000, 40_000, 1_000_000, 60_000, 2, 2
    iq = synth.fm_stream(n1 * b1 + n2 * b2, r1, ifs, amp=0.2)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    cap = 4096
    np.savez(inp, iq=iq, ifs=np.array(ifs, np.int32),
             params=np.array([r1, b1, r2, b2, n1, n2, mode, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"], cap], np.int64))
    _proc.run([sys.executable, "-c", RERATE_RUNNER, lib, inp, out],
                          env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0", WEBRADIO_NCO_EXACT="1"))
    r = np.load(out)
    assert int(r["rc"]) == 0 and int(r["left"]) == 0
    for c, f in enumerate(ifs):
        rx = oracle.Receiver(r1, f, CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
        a1 = np.concatenate([rx.run(iq[2 * b * b1: 2 * (b + 1) * b1])[0] for b in range(n1)])
        assert np.array_equal(r["a1"][c].view(np.uint32), a1.view(np.uint32)), c
        nx = oracle.Receiver(r2, f, CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
        nx.s.phase, nx.s.prev_i, nx.s.prev_q = rx.s.phase, rx.s.prev_i, rx.s.prev_q
        base = n1 * b1
        a2 = np.concatenate([nx.run(iq[2 * (base + b * b2): 2 * (base + (b + 1) * b2)])[0] for b in range(n2)])
        assert np.array_equal(r["a2"][c].view(np.uint32), a2.view(np.uint32)), c


TRACED_RUNNER = r'''
import ctypes as C, sys, numpy as np
lib, npz = sys.argv[1], sys.argv[2]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz)
iq = np.ascontiguousarray(d["iq"], np.float32); ifs = np.ascontiguousarray(d["ifs"], np.int32); p = [int(v) for v in d["params"]]
nrx = ifs.size; cap = p[8]
audio = np.zeros((2 * nrx, cap), np.float32); n = C.c_size_t(); tr = C.create_string_buffer(4096)
fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
L.wr_host_run_two_traced.argtypes = [fp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, ip, C.c_int] + [C.c_uint] * 5 + [fp, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
rc = L.wr_host_run_two_traced(iq.ctypes.data_as(fp), iq.size // 2, p[0], p[1], nrx, ifs.ctypes.data_as(ip), p[6], p[2], p[3], p[4], p[5], p[7],
                              audio.ctypes.data_as(fp), cap, C.byref(n), tr, 4096)
np.savez(sys.argv[3], rc=rc, audio=audio[:, :n.value], trace=np.array(tr.value.decode()), left=L.wr_host_registry_sizes())
'''


@pytest.mark.parametrize("late", ["0", "1"], ids=["on-time", "late"])
def test_a_source_in_device_memory_is_streamed(tmp_path, oracle, late):
    """r06: a source that produces its blocks in GPU memory (DeviceBlock) goes through
    k_tuner_stream -- the kernel bench.py's headline times -- from the product's own classes: Radio::run(), 70 Receivers in
    two lane groups, a SpectrumSink beside them (whose pushes do not close the launch), audio out of the pinned ring when
    WrStreamCtl::done says so.  The same audio bits as the same blocks out of host memory (a launch per block, WEBRADIO_STREAM=0
    likewise), the oracle's spectrum, and the library's own count of what was streamed."""
    nrx, nblk = 70, 9
    ifs = [(-35 + c) * 6250 + 99 for c in range(nrx)]
    modes = [(1, 3, 0, 2)[c % 4] for c in range(nrx)]
    rate, block = CFG["rate"], CFG["block"]
    iq = synth.fm_stream(nblk * block, rate, ifs[::5], amp=0.1, fm_base=30.0, beta=2.0)
    args = (iq, rate, block, ifs, modes, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"])
    env = {"WEBRADIO_AUDIO_LATE": late}
    host, spec_h = _run("libwr_host_pipeline.so", tmp_path, *args, fft=512, env=env)
    assert _run.last_stream == (0, 0)
    dev, spec_d = _run("libwr_host_pipeline.so", tmp_path, *args, fft=512, env=dict(env, WR_TEST_DEVICE_SOURCE="1"))
    launches, blocks = _run.last_stream
    # (blocks this small go through in one part; the first block's submit also pushes every receiver's parameters and reads
    # their slots back, which closes the launch it opened, and the one-off allocations behind it can outlast the 0.1 s after
    # which the library opens a new launch rather than ring an old one: a few launches -- but not one per block)
    assert launches >= 1 and blocks >= nblk - 1, (launches, blocks)
    assert launches <= 4, "the SpectrumSink's pushes (or anything else per block) closed the launch: %d launches" % launches
    off, _ = _run("libwr_host_pipeline.so", tmp_path, *args, fft=512, env=dict(env, WR_TEST_DEVICE_SOURCE="1", WEBRADIO_STREAM="0"))
    assert _run.last_stream == (0, 0)
    assert host.shape == dev.shape == off.shape and host.shape[1] > 0
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))
    assert np.array_equal(host.view(np.uint32), off.view(np.uint32))
    assert np.array_equal(spec_h.view(np.uint32), spec_d.view(np.uint32))
    if late == "0":
        want = _oracle(oracle, *args)
        assert float(np.abs(host - want).max()) <= 2e-5


def test_late_audio_keeps_both_front_ends_in_flight(tmp_path, oracle):
    """WEBRADIO_AUDIO_LATE=1: Radio::run() (radio.cxx:56-59 pumps the front ends one after the other)
    only ENQUEUES a block per front end and hands out the previous block's audio, which is already in
    the pinned ring: the trace of every run() is submit/audio-was-there for front end 0, then the
    same for front end 1, never a wait -- so both tuners' blocks are in flight together (on a node,
    on two GPUs).  The audio is the on-time audio one block late: a block of silence first."""
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    ifs, mode = [50_000, -75_000, 4321], 3
    rate, block, nblk = CFG["rate"], CFG["block"], 5
    iq = synth.fm_stream((nblk + 1) * block, rate, ifs, amp=0.2)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    cap = nblk * (block // 2000) + 16
    np.savez(inp, iq=iq, ifs=np.array(ifs, np.int32),
             params=np.array([rate, block, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"], mode, nblk, cap], np.int64))
    res = {}
    for late in ("0", "1", "2"):
        _proc.run([sys.executable, "-c", TRACED_RUNNER, lib, inp, out],
                              env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0", WEBRADIO_NCO_EXACT="1", WEBRADIO_TRACE="1",
                                       WEBRADIO_AUDIO_LATE=late))
        r = np.load(out)
        assert int(r["rc"]) == 0 and int(r["left"]) == 0
        res[late] = (r["audio"], str(r["trace"]))
    k2 = block // 2000
    on_time, late = res["0"][0], res["1"][0]
    assert on_time.shape == late.shape == (2 * len(ifs), nblk * k2)
    for t in range(2):                                           # the on-time audio is the oracle's
        for c, f in enumerate(ifs):
            rx = oracle.Receiver(rate, f, CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
            want = np.concatenate([rx.run(iq[2 * (b + t) * block: 2 * (b + t + 1) * block])[0] for b in range(nblk)])
            assert np.array_equal(on_time[t * len(ifs) + c].view(np.uint32), want.view(np.uint32))
    assert not late[:, :k2].any()                                # one block of silence ...
    assert np.array_equal(late[:, k2:].view(np.uint32), on_time[:, :-k2].view(np.uint32))    # ... then the same bits
    runs = res["1"][1].strip("|").split("|")
    assert runs[0] == "S0S1"                                     # nothing to hand out yet
    assert all(r == "S0A0S1A1" for r in runs[1:]), runs           # enqueue, take what is there; never wait
    # ON TIME the unchanged single-threaded Radio::run() (radio.cxx:56-59) SERIALISES the front ends: front end 1's block is
    # submitted only after front end 0's audio has been collected (its 'A'/'W' event) -- every front end's run() waits for its
    # own GPU in turn, so on an 8-GPU node the drop-in path scales across GPUs only with WEBRADIO_AUDIO_LATE (VERDICT r05
    # item 7 i; INTEGRATION.md "C4 through Radio::run()")
    for r in res["0"][1].strip("|").split("|"):
        assert r.startswith("S0") and "S1" in r
        got0 = max(r.find("A0"), r.find("W0"))
        assert 0 < got0 < r.index("S1"), r
    # WEBRADIO_AUDIO_LATE=2: no flush after the submit -- the demodulator + audio filter of a block ride in the
    # NEXT block's launch (one launch per block, as bench.py) -- and the sinks get the block before the previous
    # one: two blocks of silence, then the same bits, and still never a wait
    late2 = res["2"][0]
    assert late2.shape == on_time.shape and not late2[:, :2 * k2].any()
    assert np.array_equal(late2[:, 2 * k2:].view(np.uint32), on_time[:, :-2 * k2].view(np.uint32))
    runs2 = res["2"][1].strip("|").split("|")
    assert runs2[0] == runs2[1] == "S0S1" and all(r == "S0A0S1A1" for r in runs2[2:]), runs2


FILE_RUNNER = r'''
import ctypes as C, sys, numpy as np
lib, path, out = sys.argv[1], sys.argv[2], sys.argv[3]
p = [int(v) for v in sys.argv[4:]]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
fp = C.POINTER(C.c_float)
L.wr_host_run_file.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint,
                               C.c_uint, C.c_int, fp, C.c_size_t, C.POINTER(C.c_size_t)]
audio = np.zeros(1 << 16, np.float32); n = C.c_size_t()
rc = L.wr_host_run_file(path.encode(), *p, audio.ctypes.data_as(fp), audio.size, C.byref(n))
L.wr_host_stream_blocks.restype = C.c_ulonglong
np.savez(out, rc=rc, audio=audio[:n.value], left=L.wr_host_registry_sizes(), stream_blocks=int(L.wr_host_stream_blocks()))
'''


@pytest.mark.parametrize("with_frontend", [1, 0])
def test_c1_recorded_rtlsdr_file(tmp_path, oracle, with_frontend):
    """BASELINE config 1 end to end: FileTuner replays an RTL-SDR format recording (the capture and what the REAL
    reference chain gave for it: tests/golden/reference_c1.npz, see test_gpu_tuner.py::test_c1_single_receiver_u8_file), one
    DownConverter + FM Receiver.  The recording's bytes are staged on the device once per block (2 bytes per
    frame over PCIe) and converted there for SpectrumSink and receiver alike."""
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_c1.npz"))
    c1 = synth.C1
    n = int(g["block_frames"])
    path = str(tmp_path / "capture.bin")
    g["u8"].tofile(path)
    out = str(tmp_path / "out.npz")
    args = [c1["input_rate"], n, 4, c1["if_hz"], 1, c1["chan_passband"], c1["chan_rate"], c1["audio_passband"],
            c1["audio_rate"], with_frontend]
    # (default: the recording's BYTES are staged and converted on the device, the float vector stays unfilled;
    #  WEBRADIO_NO_U8_STAGING=1: FileTuner converts on the host and the float block is staged, as r02 did)
    for env, tol in (({"WEBRADIO_NCO_EXACT": "1"}, 4.8e-7), ({}, 1e-5),
                     ({"WEBRADIO_NCO_EXACT": "1", "WEBRADIO_NO_U8_STAGING": "1"}, 4.8e-7)):
        _proc.run([sys.executable, "-c", FILE_RUNNER, lib, path, out] + [str(a) for a in args],
                              env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0", **env))
        r = np.load(out)
        assert int(r["rc"]) == 0 and int(r["left"]) == 0
        assert r["audio"].size == g["audio"].size
        assert np.abs(r["audio"] - g["audio"]).max() <= tol
    # one block late: FileTuner alternates between two byte buffers and its run() no longer waits for the transfer of the
    # block it has just handed out (RawU8Block::rawU8Buffers, wr_dev_wait_uploads_but) -- the same audio, a block later
    # (silence first, the last block's audio dropped at stop())
    for late in ("1", "2"):
        _proc.run([sys.executable, "-c", FILE_RUNNER, lib, path, out] + [str(a) for a in args],
                              env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0", WEBRADIO_NCO_EXACT="1", WEBRADIO_AUDIO_LATE=late))
        r = np.load(out)
        assert int(r["rc"]) == 0 and int(r["left"]) == 0 and r["audio"].size == g["audio"].size
        per = g["audio"].size // 4 * int(late)
        assert not r["audio"][:per].any()
        assert np.abs(r["audio"][per:] - g["audio"][:-per]).max() <= 4.8e-7


STRESS_RUNNER = r'''
import ctypes as C, sys, numpy as np
lib, npz, out = sys.argv[1], sys.argv[2], sys.argv[3]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz); iq = np.ascontiguousarray(d["iq"], np.float32); p = d["params"]
fp = C.POINTER(C.c_float)
L.wr_host_setter_stress.restype = C.c_long
L.wr_host_setter_stress.argtypes = [fp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, fp, C.c_size_t,
                                    C.POINTER(C.c_size_t)]
nrx, cap = int(p[2]), 4096
audio = np.zeros((nrx, cap), np.float32); n = C.c_size_t()
calls = L.wr_host_setter_stress(iq.ctypes.data_as(fp), iq.size // 2, int(p[0]), int(p[1]), nrx, int(p[3]), int(p[4]),
                                audio.ctypes.data_as(fp), cap, C.byref(n))
np.savez(out, calls=calls, audio=audio[:, :n.value], left=L.wr_host_registry_sizes())
'''


def test_setters_from_another_thread_while_running(tmp_path):
    """H7: the REST handlers call setIF/setPassband/setModeString from other threads while the
    pipeline runs (receiverhandler.cxx:130-137); nothing may crash, hang or produce non-finite
    audio, and the setters must actually have been interleaved with the blocks."""
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    rate, block = CFG["rate"], CFG["block"]
    iq = synth.fm_stream(30 * block, rate, [50_000, -75_000], amp=0.3)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, iq=iq, params=np.array([rate, block, 12, CFG["crate"], CFG["arate"]], np.int64))
    _proc.run([sys.executable, "-c", STRESS_RUNNER, lib, inp, out], env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0"),
                          timeout=240)
    r = np.load(out)
    # (how many setter calls fit beside 30 blocks depends on the box and on how fast the blocks go: hundreds to thousands)
    assert int(r["calls"]) > 100 and int(r["left"]) == 0
    assert r["audio"].shape == (12, block // 2000) and np.isfinite(r["audio"]).all()


MS_RUNNER = r'''
import ctypes as C, sys, numpy as np
lib, npz, out = sys.argv[1], sys.argv[2], sys.argv[3]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz); iq = np.ascontiguousarray(d["iq"], np.float32); p = d["params"]
rates = np.ascontiguousarray(d["rates"], np.uint32); pbs = np.ascontiguousarray(d["pbs"], np.uint32)
fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint)
L.wr_host_run_multistage.restype = C.c_long
L.wr_host_run_multistage.argtypes = [fp, C.c_size_t, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_uint, up, up, C.c_uint, C.c_uint,
                                     fp, C.c_size_t]
audio = np.zeros(1 << 16, np.float32)
n = L.wr_host_run_multistage(iq.ctypes.data_as(fp), iq.size // 2, int(p[0]), int(p[1]), int(p[2]), int(p[3]), rates.size,
                             rates.ctypes.data_as(up), pbs.ctypes.data_as(up), int(p[4]), int(p[5]),
                             audio.ctypes.data_as(fp), audio.size)
np.savez(out, n=n, audio=audio[:max(n, 0)], traced=L.wr_host_trace_count())
'''


def test_multistage_decimation_chain_on_device(tmp_path, oracle):
    """a 12.5 kHz-wide channel off a 20 Msps stream.  One 64-tap stage cannot do it
    (maxbin = 64 * 12500 / 20e6 / 2 = 0: all-zero taps, the reference is silent); three decimating
    LowPass stages in a row (20 M -> 1 M -> 100 k -> 25 k) can, using nothing but the DspBlock API.
    Such a graph is not the fused Receiver shape: every block runs its own kernel, and the
    intermediates are handed over in device memory.  Checked against the oracle's cascade of
    the reference's own blocks, bit for bit (AM detector)."""
    fs, block, f_if = 20_000_000, 400_000, 2_345_000
    rates, pbs = [1_000_000, 100_000, 25_000], [700_000, 40_000, 12_500]     # maxbin 1, 1, 4
    apb, arate, mode = 3_000, 12_500, 0
    assert oracle.lowpass_maxbin(12_500, fs) == 0                    # the single-stage design is degenerate
    n = 3 * block
    t = np.arange(n) / fs
    am = 1.0 + 0.5 * np.sin(2 * np.pi * 1_000 * t)                   # 1 kHz tone, AM on the carrier at f_if
    rng = np.random.default_rng(3)
    sig = 0.3 * am * np.exp(2j * np.pi * f_if * t) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    iq = np.empty(2 * n, np.float32)
    iq[0::2], iq[1::2] = sig.real, sig.imag
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, iq=iq, params=np.array([fs, block, f_if, mode, apb, arate], np.int64), rates=np.array(rates), pbs=np.array(pbs))
    _proc.run([sys.executable, "-c", MS_RUNNER, lib, inp, out], env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0"),
                          timeout=240)
    r = np.load(out)
    got = r["audio"]
    # the oracle's cascade: the reference's blocks chained the same way
    table = oracle.sin_table()
    firs, rate_in = [], fs
    for ro, pb in zip(rates, pbs):
        firs.append(oracle.Fir(2, rate_in // ro, oracle.lowpass_design(pb, rate_in)))
        rate_in = ro
    fa = oracle.Fir(1, rate_in // arate, oracle.lowpass_design(apb, rate_in))
    phase, prev, want = 0, (0.0, 0.0), []
    for b in range(3):
        x, phase = oracle.mix(table, phase, oracle.phase_step(f_if, fs), iq[2 * b * block: 2 * (b + 1) * block])
        for f in firs:
            x = f.process(x)
        d, prev = oracle.demod(mode, prev, x)
        want.append(fa.process(d))
    want = np.concatenate(want)
    assert int(r["n"]) == want.size == 3 * block // (fs // arate)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # and it really receives: the 1 kHz tone dominates the audio once the filters have settled
    tail = got[got.size // 3:] - got[got.size // 3:].mean()
    spec = np.abs(np.fft.rfft(tail * np.hanning(tail.size)))
    spec[:3] = 0.0                                                    # what is left of the carrier's DC term
    assert abs(np.argmax(spec) * arate / tail.size - 1_000) < 2 * arate / tail.size


@pytest.mark.parametrize("fused", [1, 0])
def test_two_lowpass_stages_in_a_row_ride_the_tuner_batch(tmp_path, oracle, fused):
    """DownConverter -> LowPass -> LowPass -> Demodulator -> LowPass (a 12.5 kHz channel off 5 Msps:
    5 M -> 250 k -> 25 k -> AM -> 5 k) built with nothing but connect(): the five blocks enrol as one
    channel of the source's tuner batch, the second LowPass as the batch's second channel-filter
    stage (WEBRADIO_TRACE shows the batch submitting); with WEBRADIO_NO_FUSION=1 every block runs its
    own kernel.  Either way the audio is the oracle cascade's, bit for bit (EXACT NCO)."""
    fs, block, f_if = 5_000_000, 200_000, 1_234_500
    rates, pbs = [250_000, 25_000], [160_000, 12_500]
    apb, arate, mode = 4_000, 5_000, 0
    assert oracle.lowpass_maxbin(12_500, fs) == 0
    n = 3 * block
    t = np.arange(n) / fs
    am = 1.0 + 0.5 * np.sin(2 * np.pi * 400 * t)
    rng = np.random.default_rng(5)
    sig = 0.3 * am * np.exp(2j * np.pi * f_if * t) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    iq = np.empty(2 * n, np.float32)
    iq[0::2], iq[1::2] = sig.real, sig.imag
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, iq=iq, params=np.array([fs, block, f_if, mode, apb, arate], np.int64), rates=np.array(rates), pbs=np.array(pbs))
    env = dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_NCO_EXACT="1", WEBRADIO_TRACE="1")
    if not fused:
        env["WEBRADIO_NO_FUSION"] = "1"
    _proc.run([sys.executable, "-c", MS_RUNNER, lib, inp, out], env=env, timeout=240)
    r = np.load(out)
    assert (int(r["traced"]) > 0) == bool(fused)
    table = oracle.sin_table()
    firs, rate_in = [], fs
    for ro, pb in zip(rates, pbs):
        firs.append(oracle.Fir(2, rate_in // ro, oracle.lowpass_design(pb, rate_in)))
        rate_in = ro
    fa = oracle.Fir(1, rate_in // arate, oracle.lowpass_design(apb, rate_in))
    phase, prev, want = 0, (0.0, 0.0), []
    for b in range(3):
        x, phase = oracle.mix(table, phase, oracle.phase_step(f_if, fs), iq[2 * b * block: 2 * (b + 1) * block])
        for f in firs:
            x = f.process(x)
        d, prev = oracle.demod(mode, prev, x)
        want.append(fa.process(d))
    want = np.concatenate(want)
    assert int(r["n"]) == want.size
    assert np.array_equal(r["audio"].view(np.uint32), want.view(np.uint32))


TAP_RUNNER = r"""
import ctypes as C, sys, numpy as np
lib, npz = sys.argv[1], sys.argv[2]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
d = np.load(npz)
iq = np.ascontiguousarray(d["iq"], np.float32); ifs = np.ascontiguousarray(d["ifs"], np.int32); p = [int(v) for v in d["params"]]
nrx = ifs.size; cap = p[8]; tcap = p[9]
audio = np.zeros((nrx, cap), np.float32); n = C.c_size_t(); tap = np.zeros(tcap, np.float32); tn = C.c_size_t(); tb = C.c_uint()
fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
L.wr_host_run_tap.argtypes = [fp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, ip, C.c_int] + [C.c_uint] * 5 + [fp, C.c_size_t, C.POINTER(C.c_size_t), fp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint)]
rc = L.wr_host_run_tap(iq.ctypes.data_as(fp), iq.size // 2, p[0], p[1], nrx, ifs.ctypes.data_as(ip), p[2], p[3], p[4], p[5], p[6], p[7],
                       audio.ctypes.data_as(fp), cap, C.byref(n), tap.ctypes.data_as(fp), tcap, C.byref(tn), C.byref(tb))
np.savez(sys.argv[3], rc=rc, audio=audio[:, :n.value], tap=tap[:tn.value], tap_blocks=tb.value, left=L.wr_host_registry_sizes())
"""


def test_second_consumer_inside_a_fused_chain(tmp_path, oracle):
    """attaching a consumer to a block inside a fused Receiver chain used to hand it the
    empty, elided vector, and the source's run() failed.  Three receivers (USB), six blocks; before
    block 2 a tap is connected to receiver 0's demodulator.  Receiver 0 leaves the tuner batch there
    and goes on block by block with its NCO phase and its filters restarted empty: the tap gets every
    demodulated block from then on, the audio equals the oracle's for a receiver whose two filters
    are fresh at block 2; receivers 1 and 2 stay fused and are the oracle's bits throughout."""
    lib = os.path.join(CXXT, "libwr_host_pipeline.so")
    ifs, mode, rate, block, nblk, tap_at = [50_000, -75_000, 4321], 2, 2_000_000, 40_000, 6, 2
    iq = synth.fm_stream(nblk * block, rate, ifs, amp=0.2)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    cap, tcap = 4096, 1 << 16
    np.savez(inp, iq=iq, ifs=np.array(ifs, np.int32),
             params=np.array([rate, block, mode, CFG["cpb"], CFG["crate"], CFG["apb"], CFG["arate"], tap_at, cap, tcap], np.int64))
    _proc.run([sys.executable, "-c", TAP_RUNNER, lib, inp, out],
                          env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_PIN_MIN_BYTES="0", WEBRADIO_NCO_EXACT="1"))
    r = np.load(out)
    assert int(r["rc"]) == 0 and int(r["left"]) == 0
    blocks = [iq[2 * b * block: 2 * (b + 1) * block] for b in range(nblk)]
    for c in (1, 2):
        rx = oracle.Receiver(rate, ifs[c], CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
        want = np.concatenate([rx.run(x)[0] for x in blocks])
        assert np.array_equal(r["audio"][c].view(np.uint32), want.view(np.uint32)), c
    # receiver 0: the oracle's chain up to the tap; then the same NCO phase with fresh filters
    rx = oracle.Receiver(rate, ifs[0], CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
    before = np.concatenate([rx.run(x)[0] for x in blocks[:tap_at]])
    nx = oracle.Receiver(rate, ifs[0], CFG["cpb"], CFG["crate"], mode, CFG["apb"], CFG["arate"])
    nx.s.phase, nx.s.prev_i, nx.s.prev_q = rx.s.phase, rx.s.prev_i, rx.s.prev_q
    outs = [nx.run(x) for x in blocks[tap_at:]]
    after = np.concatenate([o[0] for o in outs])
    assert np.array_equal(r["audio"][0].view(np.uint32), np.concatenate([before, after]).view(np.uint32))
    assert int(r["tap_blocks"]) == nblk - tap_at
    demod = np.concatenate([o[2] for o in outs])
    assert np.array_equal(r["tap"].view(np.uint32), demod.view(np.uint32))
