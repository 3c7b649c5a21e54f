"""FileTuner (webradio_amd/host/filetuner.cxx), the RTL-SDR-format recording source BASELINE
config 1 needs: host-only logic, no GPU.  Conversion must be the reference's
(u8 - 128)/128 (io/rtlsdrtuner.cxx:106), i.e. the oracle's wro_u8_to_float."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXXT = os.path.join(ROOT, "tests", "cxx")


@pytest.fixture(scope="module")
def checks():
    lib = os.path.join(CXXT, "libwr_cpu_host_checks.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", CXXT, "libwr_cpu_host_checks.so"])
    L = C.CDLL(lib)
    fp = C.POINTER(C.c_float)
    L.wr_filetuner_play.restype = C.c_long
    L.wr_filetuner_play.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_int, fp, C.c_size_t, C.POINTER(C.c_int),
                                    C.POINTER(C.c_ubyte), C.POINTER(C.c_size_t)]
    return L


def _play(L, path, block, runs, loop):
    out = np.zeros(2 * block * runs, np.float32)
    ok = C.c_int()
    raw = np.zeros(2 * block, np.uint8)
    rf = C.c_size_t()
    n = L.wr_filetuner_play(str(path).encode(), block, runs, loop, out.ctypes.data_as(C.POINTER(C.c_float)), out.size,
                            C.byref(ok), raw.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(rf))
    return n, ok.value, out[:max(n, 0)], raw[: 2 * rf.value]


def test_conversion_and_end_of_file(checks, oracle, tmp_path):
    u8 = np.arange(256, dtype=np.uint8).repeat(2)[: 2 * 200]
    u8 = np.concatenate([u8, u8[::-1]])                      # 400 frames
    path = tmp_path / "cap.bin"
    u8.tofile(path)
    n, ok, out, raw = _play(checks, path, 100, 5, 0)          # 4 full blocks, the 5th run hits EOF
    assert ok == 4 and n == 800
    assert np.array_equal(out, oracle.u8_to_float(u8))
    assert raw.size == 0                                     # no valid raw block after a failed read


def test_loop_and_raw_bytes(checks, oracle, tmp_path):
    u8 = (np.arange(600) * 7 % 256).astype(np.uint8)          # 300 frames
    path = tmp_path / "cap.bin"
    u8.tofile(path)
    n, ok, out, raw = _play(checks, path, 100, 7, 1)          # wraps around twice
    assert ok == 7 and n == 1400
    want = oracle.u8_to_float(np.tile(u8, 3)[:1400])
    assert np.array_equal(out, want)
    assert np.array_equal(raw, np.tile(u8, 3)[1200:1400])    # bytes of the last block


def test_two_byte_buffers_in_turn(checks, tmp_path):
    """RawU8Block::rawU8Buffers() == 2: consecutive blocks come out of different byte buffers and a block's bytes are
    still intact after the next run() (what lets the GPU runtime keep a transfer in flight while the source moves on)."""
    u8 = (np.arange(4096) * 13 % 256).astype(np.uint8)
    path = tmp_path / "cap.bin"
    u8.tofile(path)
    checks.wr_filetuner_two_buffers.argtypes = [C.c_char_p, C.c_uint, C.c_uint]
    assert checks.wr_filetuner_two_buffers(str(path).encode(), 128, 9) == 0


def test_missing_file(checks, tmp_path):
    n, ok, out, raw = _play(checks, tmp_path / "nope.bin", 10, 1, 0)
    assert n == -1                                           # init() fails -> start() false


def test_device_handover_bookkeeping():
    """DspBlock::publishDeviceOutput / upstreamDeviceOutput / acceptsDeviceInput / hostOutputNeeded
    (the hand-over between stand-alone GPU blocks), exercised with stub blocks: no GPU needed."""
    import ctypes as C
    lib = os.path.join(CXXT, "libwr_cpu_host_checks.so")
    L = C.CDLL(lib)
    assert L.wr_handover_checks() == 0
