"""ctypes binding of oracle/libwr_oracle.so (the CPU restatement) and, where it has
been built, oracle/_ref/libwr_ref.so (the real reference sources that compile here).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under webradio_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libwr_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libwr_ref.so")
REFERENCE_ROOT = "/root/reference"

AM, FM, USB, LSB = range(4)
_fp = C.POINTER(C.c_float)


def build(ref=True):
    """(Re)build the oracle, and the reference harness when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class FirState(C.Structure):
    _fields_ = [("channels", C.c_uint), ("decimation", C.c_uint), ("length", C.c_uint),
                ("coeff", C.c_float * 1024), ("block", _fp), ("block_len", C.c_size_t)]


class SpectrumState(C.Structure):
    _fields_ = [("fft_size", C.c_uint), ("inoffset", C.c_uint), ("inbuf", _fp), ("outbuf", _fp),
                ("window", _fp), ("frames_done", C.c_ulong)]


class ReceiverState(C.Structure):
    _fields_ = [("input_rate", C.c_uint), ("chan_rate", C.c_uint), ("audio_rate", C.c_uint),
                ("d1", C.c_uint), ("d2", C.c_uint), ("if_hz", C.c_int), ("phase_step", C.c_int),
                ("mode", C.c_int), ("phase", C.c_uint), ("prev_i", C.c_float), ("prev_q", C.c_float),
                ("chan_fir", FirState), ("audio_fir", FirState),
                ("mixed", _fp), ("chan_iq", _fp), ("demod", _fp),
                ("mixed_len", C.c_size_t), ("chan_len", C.c_size_t), ("demod_len", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        L.wro_phase_step.restype = C.c_int
        L.wro_phase_step.argtypes = [C.c_int, C.c_uint]
        L.wro_lowpass_maxbin.restype = C.c_uint
        L.wro_lowpass_maxbin.argtypes = [C.c_uint, C.c_uint]
        L.wro_lowpass_maxbin_n.restype = C.c_uint
        L.wro_lowpass_maxbin_n.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        L.wro_fir_process.restype = C.c_size_t
        L.wro_receiver_run.restype = C.c_size_t
        L.wro_bench_receivers.restype = C.c_double
        L.wro_bench_receivers_mt.restype = C.c_double
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp)


def sin_table():
    t = np.empty(65536, np.float32)
    lib().wro_sin_table(_p(t))
    return t


def phase_step(if_hz, rate):
    return lib().wro_phase_step(int(if_hz), int(rate))


def mix(table, phase, step, iq):
    iq = _f32(iq)
    out = np.empty_like(iq)
    ph = C.c_uint(phase)
    lib().wro_mix(_p(table), C.byref(ph), C.c_int(step), _p(iq), _p(out), C.c_size_t(iq.size // 2))
    return out, ph.value


def lowpass_window():
    w = np.empty(64, np.float32)
    lib().wro_lowpass_window(_p(w))
    return w


def lowpass_maxbin(passband, rate):
    return lib().wro_lowpass_maxbin(int(passband), int(rate))


def lowpass_design(passband, rate, length=64):
    """LowPass::recalculate with _firLength = length (a power of two; 64 is what the reference compiles in)"""
    c = np.empty(length, np.float32)
    lib().wro_lowpass_design_n(C.c_uint(length), C.c_uint(passband), C.c_uint(rate), _p(c))
    return c


def lowpass_window_n(length):
    w = np.empty(length, np.float32)
    lib().wro_lowpass_window_n(C.c_uint(length), _p(w))
    return w


def lowpass_maxbin_n(length, passband, rate):
    return lib().wro_lowpass_maxbin_n(int(length), int(passband), int(rate))


class Fir:
    """LowPass::process with its history (oracle)."""

    def __init__(self, channels, decimation, coeff):
        self.s = FirState()
        coeff = _f32(coeff)
        lib().wro_fir_init_n(C.byref(self.s), C.c_uint(channels), C.c_uint(decimation), _p(coeff),
                             C.c_uint(coeff.size))
        self.channels, self.decimation = channels, decimation

    def process(self, x):
        x = _f32(x)
        nout = (x.size // self.channels // self.decimation) * self.channels
        out = np.empty(max(nout, 1), np.float32)
        n = lib().wro_fir_process(C.byref(self.s), _p(x), C.c_size_t(x.size), _p(out))
        return out[:n].copy()

    def __del__(self):
        try:
            lib().wro_fir_free(C.byref(self.s))
        except Exception:
            pass


def demod(mode, prev, iq):
    iq = _f32(iq)
    out = np.empty(iq.size // 2, np.float32)
    pi_, pq_ = C.c_float(prev[0]), C.c_float(prev[1])
    ok = lib().wro_demod(C.c_int(mode), C.byref(pi_), C.byref(pq_), _p(iq), _p(out), C.c_size_t(iq.size // 2))
    assert ok
    return out, (pi_.value, pq_.value)


def spectrum_window(n):
    w = np.empty(n, np.float32)
    lib().wro_spectrum_window(C.c_uint(n), _p(w))
    return w


def fft_forward(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().wro_fft_forward(C.c_uint(x.size // 2), _p(x), _p(out))
    return out


def spectrum_db(bins):
    bins = _f32(bins)
    n = bins.size // 2
    out = np.empty(n, np.float32)
    lib().wro_spectrum_db(C.c_uint(n), _p(bins), _p(out))
    return out


def waterfall_row(bins, width, hold):
    bins = _f32(bins)
    n = bins.size // 2
    db = np.empty(width, np.float32)
    pal = np.empty(width, np.uint8)
    lib().wro_waterfall_row(C.c_uint(n), _p(bins), C.c_uint(width), C.c_int(hold), _p(db),
                            pal.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return db, pal


class Spectrum:
    def __init__(self, n):
        self.s = SpectrumState()
        self.n = n
        assert lib().wro_spectrum_init(C.byref(self.s), C.c_uint(n))

    def process(self, iq):
        iq = _f32(iq)
        lib().wro_spectrum_process(C.byref(self.s), _p(iq), C.c_size_t(iq.size // 2))

    def get(self):
        out = np.empty(self.n, np.float32)
        lib().wro_spectrum_get(C.byref(self.s), _p(out))
        return out

    def bins(self):
        return np.ctypeslib.as_array(self.s.outbuf, shape=(2 * self.n,)).copy()

    @property
    def frames_done(self):
        return self.s.frames_done

    def __del__(self):
        try:
            lib().wro_spectrum_free(C.byref(self.s))
        except Exception:
            pass


class Receiver:
    """One reference Receiver chain (radio.cxx:62-90) on the CPU."""

    _table = None

    def __init__(self, input_rate, if_hz, chan_passband, chan_rate, mode, audio_passband, audio_rate):
        self.s = ReceiverState()
        ok = lib().wro_receiver_init(C.byref(self.s), C.c_uint(input_rate), C.c_int(if_hz),
                                     C.c_uint(chan_passband), C.c_uint(chan_rate), C.c_int(mode),
                                     C.c_uint(audio_passband), C.c_uint(audio_rate))
        if not ok:
            raise ValueError("rates not integer related")
        if Receiver._table is None:
            Receiver._table = sin_table()

    @property
    def d1(self):
        return self.s.d1

    @property
    def d2(self):
        return self.s.d2

    def run(self, iq):
        """returns (audio, chan_iq, demod) for one block"""
        iq = _f32(iq)
        n = iq.size // 2
        k1 = n // self.s.d1
        k2 = k1 // self.s.d2
        audio = np.empty(max(k2, 1), np.float32)
        chan = np.empty(max(2 * k1, 1), np.float32)
        dem = np.empty(max(k1, 1), np.float32)
        got = lib().wro_receiver_run(C.byref(self.s), _p(Receiver._table), _p(iq), C.c_size_t(n),
                                     _p(audio), _p(chan), _p(dem))
        return audio[:got].copy(), chan[: 2 * k1].copy(), dem[:k1].copy()

    def set_if(self, if_hz):
        self.s.if_hz = if_hz
        self.s.phase_step = phase_step(if_hz, self.s.input_rate)

    def set_mode(self, mode):
        self.s.mode = mode

    def __del__(self):
        try:
            lib().wro_receiver_free(C.byref(self.s))
        except Exception:
            pass


def bench_receivers(input_rate, ifs, chan_passband, chan_rate, mode, audio_passband, audio_rate,
                    iq, nblocks):
    iq = _f32(iq)
    ifs = np.ascontiguousarray(ifs, dtype=np.int32)
    return lib().wro_bench_receivers(C.c_uint(input_rate), ifs.ctypes.data_as(C.POINTER(C.c_int)),
                                     C.c_uint(ifs.size), C.c_uint(chan_passband), C.c_uint(chan_rate),
                                     C.c_int(mode), C.c_uint(audio_passband), C.c_uint(audio_rate),
                                     _p(iq), C.c_size_t(iq.size // 2), C.c_uint(nblocks), None)


def bench_receivers_mt(input_rate, ifs, chan_passband, chan_rate, mode, audio_passband, audio_rate,
                       iq, nblocks, nthreads):
    """bench_receivers on `nthreads` threads, each with its own subset of the receivers."""
    iq = _f32(iq)
    ifs = np.ascontiguousarray(ifs, dtype=np.int32)
    return lib().wro_bench_receivers_mt(C.c_uint(input_rate), ifs.ctypes.data_as(C.POINTER(C.c_int)),
                                        C.c_uint(ifs.size), C.c_uint(chan_passband), C.c_uint(chan_rate),
                                        C.c_int(mode), C.c_uint(audio_passband), C.c_uint(audio_rate),
                                        _p(iq), C.c_size_t(iq.size // 2), C.c_uint(nblocks), C.c_uint(nthreads))


def af_gain_squelch(audio, demod_in_iq, d2, gain_db=0.0, squelch_dbfs=None):
    """The two receiver controls the reference names and never implements ("FIXME: af_gain, squelch",
    web/receiverhandler.cxx:112,118-119,127) -- so this is the build's OWN definition, restated
    here in scalar float arithmetic for the tests (include/webradio_amd.h: wr_chan_set_af_gain /
    wr_chan_set_squelch): an audio frame is muted when the mean power i*i + q*q of the d2
    demodulator-input frames behind it (summed in order, divided by (float)d2) is below
    10^(dBFS/10); then the sample is multiplied by 10^(dB/20) as a float."""
    audio = np.array(audio, dtype=np.float32)
    z = np.asarray(demod_in_iq, dtype=np.float32).reshape(-1, 2)
    if squelch_dbfs is not None:
        thr = np.float32(10.0 ** (float(squelch_dbfs) / 10.0))
        for k in range(audio.size):
            p = np.float32(0.0)
            for i in range(d2):
                zi, zq = z[k * d2 + i]
                p = np.float32(p + np.float32(np.float32(zi * zi) + np.float32(zq * zq)))
            if np.float32(p / np.float32(d2)) < thr:
                audio[k] = 0.0
    g = np.float32(10.0 ** (float(gain_db) / 20.0))
    if g != np.float32(1.0):
        audio = (audio * g).astype(np.float32)
    return audio


def u8_to_float(b):
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.empty(b.size, np.float32)
    lib().wro_u8_to_float(b.ctypes.data_as(C.POINTER(C.c_ubyte)), _p(out), C.c_size_t(b.size))
    return out


# ---- the real reference, where it could be built (oracle/_ref) -------------------

_ref = None


def ref():
    """libwr_ref.so (reference dspblock.cxx + demodulator.cxx + ref_harness.cxx) or None."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_LIB):
            if os.path.isdir(REFERENCE_ROOT):
                build(ref=True)
            else:
                return None
        R = C.CDLL(REF_LIB)
        R.wr_harness_run.restype = C.c_long
        R.wr_harness_run.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        R.wr_ref_demod.restype = C.c_long
        R.wr_ref_demod.argtypes = [C.c_char_p, _fp, C.c_size_t, C.c_size_t, C.c_int, C.c_char_p, _fp,
                                   C.c_size_t]
        _ref = R
    return _ref


def harness_trace(libhandle, idx):
    buf = C.create_string_buffer(1 << 16)
    n = libhandle.wr_harness_run(idx, buf, len(buf))
    assert 0 <= n < len(buf)
    return buf.value.decode()


def ref_demod(mode_name, iq, block_frames, switch_at=-1, mode2=None):
    iq = _f32(iq)
    n = iq.size // 2
    out = np.empty(n, np.float32)
    got = ref().wr_ref_demod(mode_name.encode(), _p(iq), n, block_frames, switch_at,
                             mode2.encode() if mode2 else None, _p(out), n)
    assert got >= 0, got
    return out[:got].copy()


# ---- r04: the reference's FFTW-calling blocks (oracle/ref_chain.cxx), over the image's hipFFTW --------------------
REF_CHAIN_LIB = os.path.join(_HERE, "_ref", "libwr_ref_chain.so")
_ref_chain = None


def ref_chain():
    """libwr_ref_chain.so -- the REAL dsp/downconverter.cxx, dsp/lowpass.cxx, dsp/demodulator.cxx,
    io/spectrumsink.cxx behind ref_chain.cxx, linked with the image's FFTW3-API library (hipFFTW:
    rocFFT underneath, so it RUNS only where a GPU is) -- or None when it has not been built."""
    global _ref_chain
    if _ref_chain is None:
        if not os.path.exists(REF_CHAIN_LIB):
            if not os.path.isdir(REFERENCE_ROOT):
                return None
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref_chain"])
        R = C.CDLL(REF_CHAIN_LIB)
        R.ref_lowpass_impulse_response.restype = C.c_int
        R.ref_lowpass_impulse_response.argtypes = [C.c_uint, C.c_uint, _fp, C.c_uint]
        R.ref_receiver_chain.restype = C.c_int
        R.ref_receiver_chain.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_uint, C.c_uint, _fp,
                                         C.c_size_t, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t,
                                         C.POINTER(C.c_size_t)]
        R.ref_downconverter.restype = C.c_int
        R.ref_downconverter.argtypes = [C.c_uint, C.c_int, _fp, C.c_size_t, C.c_size_t, _fp]
        R.ref_bench_receivers.restype = C.c_double
        R.ref_bench_receivers.argtypes = [C.c_uint, C.POINTER(C.c_int), C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint, C.c_uint,
                                          _fp, C.c_size_t, C.c_uint, C.c_uint, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
        R.ref_spectrum.restype = C.c_int
        R.ref_spectrum.argtypes = [C.c_uint, C.c_uint, _fp, C.c_size_t, C.c_size_t, _fp]
        _ref_chain = R
    return _ref_chain


def ref_lowpass_taps(passband, rate, ntaps=64):
    """the reference LowPass's coefficients, observed as its impulse response"""
    t = np.empty(ntaps, np.float32)
    got = ref_chain().ref_lowpass_impulse_response(passband, rate, _p(t), ntaps)
    assert got == ntaps, got
    return t


def ref_receiver(fs, if_hz, chan_passband, chan_rate, mode, audio_passband, audio_rate, iq, block_frames):
    """(audio, chan_iq, demod) of the reference's own Receiver chain over the whole of `iq`, block by block"""
    iq = _f32(iq)
    n = iq.size // 2
    d1, d2 = fs // chan_rate, chan_rate // audio_rate
    k1 = n // d1
    chan = np.empty(2 * k1 + 2, np.float32)
    dem = np.empty(k1 + 1, np.float32)
    aud = np.empty(k1 // d2 + 1, np.float32)
    made = (C.c_size_t * 3)()
    rc = ref_chain().ref_receiver_chain(fs, if_hz, chan_passband, chan_rate, mode, audio_passband, audio_rate, _p(iq), n,
                                        block_frames, _p(chan), chan.size, _p(dem), dem.size, _p(aud), aud.size, made)
    assert rc == 0, rc
    return aud[:made[2]].copy(), chan[:made[0]].copy(), dem[:made[1]].copy()


def ref_mix(fs, if_hz, iq, block_frames):
    iq = _f32(iq)
    out = np.empty_like(iq)
    rc = ref_chain().ref_downconverter(fs, if_hz, _p(iq), iq.size // 2, block_frames, _p(out))
    assert rc == 0, rc
    return out


def ref_spectrum_db(fs, fft_size, iq, block_frames):
    iq = _f32(iq)
    db = np.empty(fft_size, np.float32)
    rc = ref_chain().ref_spectrum(fs, fft_size, _p(iq), iq.size // 2, block_frames, _p(db))
    assert rc == 0, rc
    return db


def ref_bench_receivers(input_rate, ifs, chan_passband, chan_rate, mode, audio_passband, audio_rate, iq, nblocks, nthreads):
    """seconds `nthreads` pipeline threads of the REAL reference take for `nblocks` blocks (see oracle/ref_chain.cxx);
    returns (seconds, audio frames produced, absolute sum of the audio)"""
    iq = _f32(iq)
    ifs = np.ascontiguousarray(ifs, dtype=np.int32)
    frames, s = C.c_size_t(), C.c_double()
    dt = ref_chain().ref_bench_receivers(input_rate, ifs.ctypes.data_as(C.POINTER(C.c_int)), ifs.size, chan_passband, chan_rate,
                                         mode, audio_passband, audio_rate, _p(iq), iq.size // 2, nblocks, nthreads,
                                         C.byref(frames), C.byref(s))
    assert dt > 0, dt
    return dt, frames.value, s.value
