"""The C-ABI library loads, exports every symbol include/webradio_amd.h declares, and
its host-side design helpers agree with the oracle.  No GPU needed (and none used)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from webradio_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(wr_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_functions():
    names = _declared_functions()
    assert len(names) >= 40
    assert "wr_tuner_submit" in names and "wr_spectrum_get_db" in names


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.wr_abi_version() == capi.WR_ABI_VERSION == 6


def test_python_binding_covers_header():
    assert sorted(capi.SIGNATURES) == _declared_functions()


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "webradio_amd.h")).read()
    for cite in ("dsp/downconverter.cxx:91-114", "dsp/lowpass.cxx:131-162", "dsp/demodulator.cxx:77-115",
                 "io/spectrumsink.cxx:88-123", "io/spectrumsink.cxx:125-142", "dsp/dspblock.h:82-84"):
        assert cite in text, cite


def test_design_helpers_match_oracle(oracle):
    lib = capi.load()
    step = C.c_int()
    for hz, rate in [(100_000, 2_400_000), (-100_000, 2_400_000), (-39_843_750, 100_000_000),
                     (39_843_750, 100_000_000), (1, 3), (-1, 3), (499_999_999, 1_000_000_000)]:
        assert lib.wr_phase_step(hz, rate, C.byref(step)) == 0
        assert step.value == oracle.phase_step(hz, rate)

    table = np.empty(65536, np.float32)
    assert lib.wr_sin_table(capi.ptr(table)) == 0
    assert np.array_equal(table.view(np.uint32), oracle.sin_table().view(np.uint32))

    coeff = np.empty(64, np.float32)
    mb = C.c_uint()
    cases = [(80_000, 2_400_000), (6_400_000, 100_000_000), (200_000, 2_048_000), (12_500, 100_000_000),
             (8_000, 240_000), (8_000, 250_000), (8_000, 256_000), (64_000_000, 1_000_000_000),
             (1_000_000, 2_000_000), (2_000_000, 2_000_000), (3_000_000, 2_000_000), (70_000_000, 1_000_000)]
    for pb, rate in cases:
        assert lib.wr_lowpass_design(pb, rate, capi.ptr(coeff), C.byref(mb)) == 0
        assert mb.value == oracle.lowpass_maxbin(pb, rate)
        ref = oracle.lowpass_design(pb, rate)
        # two independent evaluations of the same inverse DFT (closed form vs direct sum),
        # both accumulated in double: equal up to one float rounding
        assert np.allclose(coeff, ref, rtol=0, atol=4e-9), (pb, rate, np.abs(coeff - ref).max())

    # LowPass::_firLength as a run-time value (the reference's FIXME, lowpass.cxx:38-39): same
    # agreement at every power of two, and fir_length = 64 is the plain entry point bit for bit
    for L in (2, 4, 16, 64, 128, 256, 1024):
        c = np.empty(L, np.float32)
        for pb, rate in cases:
            assert lib.wr_lowpass_design_n(L, pb, rate, capi.ptr(c), C.byref(mb)) == 0
            assert mb.value == oracle.lowpass_maxbin_n(L, pb, rate)
            ref = oracle.lowpass_design(pb, rate, L)
            assert np.allclose(c, ref, rtol=0, atol=1e-8 * max(1, L // 64)), (L, pb, rate, np.abs(c - ref).max())
            if L == 64:
                assert lib.wr_lowpass_design(pb, rate, capi.ptr(coeff), None) == 0
                assert np.array_equal(c.view(np.uint32), coeff.view(np.uint32))
    for bad in (0, 1, 3, 96, 2048):
        assert lib.wr_lowpass_design_n(bad, 1000, 48000, capi.ptr(coeff), None) == capi.WR_ERR_ARG

    for n in (8, 512, 65536):
        w = np.empty(n, np.float32)
        assert lib.wr_spectrum_window(n, capi.ptr(w)) == 0
        assert np.array_equal(w, oracle.spectrum_window(n))


def test_argument_errors():
    lib = capi.load()
    assert lib.wr_phase_step(1, 0, None) == capi.WR_ERR_ARG
    assert lib.wr_lowpass_design(1, 0, None, None) == capi.WR_ERR_ARG
    assert b"bad argument" in lib.wr_last_error()
    assert lib.wr_dev_open(None, 0, None) == capi.WR_ERR_ARG


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must refuse to run, not fall back."""
    lib = capi.load()
    n = C.c_int(-1)
    rc = lib.wr_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.wr_dev_open(C.byref(h), 0, None) == capi.WR_ERR_NODEV
    assert not h.value
    assert b"no CPU path" in lib.wr_last_error() or b"no HIP device" in lib.wr_last_error()


def test_product_never_touches_oracle():
    """Nothing under webradio_amd/ may import, link or name the oracle."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "webradio_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".cxx", "Makefile")):
                text = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"wr_oracle|wro_|oracle/|libwr_ref", text):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
