"""A SECOND, independent restatement of the reference functions the oracle cannot be pinned for
(no FFTW in this image: DownConverter, LowPass and SpectrumSink do not build), written from the
reference text by a different route than oracle/wr_oracle.c: vectorised numpy float32 array
arithmetic and numpy's FFT instead of scalar C loops and a hand-written DFT.  Test infrastructure
only (tests/test_oracle_second_opinion.py fuzzes the C oracle against it); it pins nothing to the
reference -- it removes the single point of failure of having one reading of the source.

Reference paths are relative to webradio's src/.  numpy float32 `*`, `+`, `-` round once per
operation like the reference's unfused float arithmetic (-O2, no -ffast-math, no FMA contraction
for baseline x86-64), so everything built only from those is compared bit for bit; where libm is
involved (sinf, cosf, atan2f, log10f) or FFTW's own rounding, a tolerance is stated at the call.
"""
import numpy as np

F32 = np.float32
TABLE_BITS = 16
PHASE_BITS = 31


def sin_table():
    """dsp/downconverter.cxx:49-51: sinf((float)n * 2 * M_PI / (float)65536).
    `(float)n * 2` is a float product, `* M_PI` and `/ (float)65536` are double, and sinf takes
    that double narrowed to float.  Evaluated here as the correctly rounded sine of the float
    argument (glibc's sinf is correctly rounded for all but a handful of arguments)."""
    n2 = np.arange(1 << TABLE_BITS, dtype=F32) * F32(2)
    arg = (n2.astype(np.float64) * np.pi / np.float64(F32(1 << TABLE_BITS))).astype(F32)
    return np.sin(arg.astype(np.float64)).astype(F32)


def phase_step(if_hz, rate):
    """dsp/downconverter.cxx:65,80: (int)((int64)hz * (1 << 31) / (int64)rate) -- C integer
    division truncates toward zero."""
    num = int(if_hz) * (1 << PHASE_BITS)
    q = abs(num) // int(rate)
    return -q if num < 0 else q


def mix(table, phase, step, iq):
    """dsp/downconverter.cxx:91-114 for a whole block at once.  Frame n uses the phase before
    its own increment: (phase + n*step) mod 2^31; sine index = phase >> 15, cosine a quarter of
    the table further on.  Returns (mixed, phase after the block)."""
    iq = np.asarray(iq, dtype=F32)
    n = iq.size // 2
    ph = (np.uint64(phase) + np.arange(n, dtype=np.uint64) * np.uint64(step & 0xFFFFFFFF)) & np.uint64((1 << PHASE_BITS) - 1)
    sinidx = (ph >> np.uint64(PHASE_BITS - TABLE_BITS)).astype(np.int64)
    cosidx = (sinidx + (1 << TABLE_BITS) // 4) & ((1 << TABLE_BITS) - 1)
    s, c = table[sinidx], table[cosidx]
    i, q = iq[0::2], iq[1::2]
    out = np.empty_like(iq)
    out[0::2] = i * c + q * s
    out[1::2] = q * c - i * s
    end = (int(phase) + n * (step & 0xFFFFFFFF)) & ((1 << PHASE_BITS) - 1)
    return out, end


def lowpass_window(length=64):
    """dsp/lowpass.cxx:104-110: 0.54 - 0.46 * cosf(2 * M_PI * (float)n / (float)(L - 1)), stored as
    float, then divided by (float)L in float.  cosf: correctly rounded cosine of the float argument."""
    n = np.arange(length, dtype=F32)
    arg = (2.0 * np.pi * n.astype(np.float64) / np.float64(F32(length - 1))).astype(F32)
    c = np.cos(arg.astype(np.float64)).astype(F32)
    w = (0.54 - 0.46 * c.astype(np.float64)).astype(F32)
    return w / F32(length)


def lowpass_maxbin(passband, rate, length=64):
    """dsp/lowpass.cxx:167: unsigned 32-bit arithmetic, left to right."""
    return ((int(length) * int(passband)) % (1 << 32)) // int(rate) // 2


def lowpass_design(passband, rate, length=64):
    """dsp/lowpass.cxx:164-189: a 0/1 spectrum, real and symmetric, through an unnormalised inverse
    DFT (FFTW_BACKWARD); coeff[n] = Re(impulse[(n + L/2) & (L-1)]) * window[n].
    FFTW computes in float; numpy's complex128 inverse FFT narrowed once stands in for it."""
    L = int(length)
    maxbin = lowpass_maxbin(passband, rate, L)
    spec = np.zeros(L, dtype=np.float64)
    for n in range(L // 2 + 1):
        spec[n] = spec[(L - n) & (L - 1)] = 1.0 if n < maxbin else 0.0
    impulse = (np.fft.ifft(spec) * L).real.astype(F32)
    order = (np.arange(L) + L // 2) & (L - 1)
    return impulse[order] * lowpass_window(L)


class Fir:
    """dsp/lowpass.cxx:131-162.  `block` is a member that persists between calls: it is RESIZED to
    input + (L-1) frames first (a shrink truncates it, a growth appends zeros) and only then are its
    last L-1 frames moved to the front as history (:138-141) -- with a constant block size that is
    the true history, after a size change it is not; the new input follows.
    Every output starts from 0.0f and adds coeff[L-1-j] * block[k*D + j] for j = 0, 1, ... in that
    order (the reverse iterator over coeff walks the input forwards)."""

    def __init__(self, channels, decimation, coeff):
        self.ch, self.d = int(channels), int(decimation)
        self.coeff = np.asarray(coeff, dtype=F32)
        self.block = np.zeros(0, dtype=F32)

    def process(self, x):
        x = np.asarray(x, dtype=F32)
        L, ch, d = self.coeff.size, self.ch, self.d
        hist = (L - 1) * ch
        size = x.size + hist
        if self.block.size != size:                  # vector::resize
            nb = np.zeros(size, dtype=F32)
            keep = min(size, self.block.size)
            nb[:keep] = self.block[:keep]
            self.block = nb
        self.block[:hist] = self.block[size - hist:].copy()
        self.block[hist:] = x
        rows = self.block.reshape(-1, ch)
        k = (x.size // ch) // d
        acc = np.zeros((k, ch), dtype=F32)
        if k:
            for j in range(L):
                acc = acc + self.coeff[L - 1 - j] * rows[j: j + (k - 1) * d + 1: d]
        return acc.reshape(-1)


def demod(mode, prev, iq):
    """dsp/demodulator.cxx:77-115; returns (out, new prev).  FM: atan2f(Re, Im) / M_PI / 2.0 with
    the division in double."""
    iq = np.asarray(iq, dtype=F32)
    i, q = iq[0::2], iq[1::2]
    pi_ = np.concatenate([[F32(prev[0])], i[:-1]]).astype(F32)
    pq_ = np.concatenate([[F32(prev[1])], q[:-1]]).astype(F32)
    if mode == "AM":
        out = np.sqrt(i * i + q * q)
    elif mode == "FM":
        ii = i * pi_ + q * pq_
        qq = q * pi_ - i * pq_
        a = np.arctan2(ii.astype(np.float64), qq.astype(np.float64)).astype(F32)    # atan2f, correctly rounded
        out = (a.astype(np.float64) / np.pi / 2.0).astype(F32)
    elif mode == "USB":
        out = i + q
    else:
        out = i - q
    new_prev = (i[-1], q[-1]) if i.size else prev
    return out.astype(F32), new_prev


def spectrum_window(n):
    """io/spectrumsink.cxx:73: 0.54 - 0.46 * cosf(2 * M_PI * (float)n / (float)(N - 1))."""
    k = np.arange(n, dtype=F32)
    arg = (2.0 * np.pi * k.astype(np.float64) / np.float64(F32(n - 1))).astype(F32)
    c = np.cos(arg.astype(np.float64)).astype(F32)
    return (0.54 - 0.46 * c.astype(np.float64)).astype(F32)


class Spectrum:
    """io/spectrumsink.cxx:88-142: frames fill `inbuf` from `inoffset`; a full buffer is windowed in
    place (float), transformed (forward, unnormalised) and the offset returns to 0 (no overlap)."""

    def __init__(self, n):
        self.n = int(n)
        self.window = spectrum_window(n)
        self.inbuf = np.zeros(2 * self.n, dtype=F32)
        self.off = 0
        self.out = None
        self.frames = 0

    def process(self, iq):
        iq = np.asarray(iq, dtype=F32)
        pos, total = 0, iq.size // 2
        while pos < total:
            take = min(self.n - self.off, total - pos)
            self.inbuf[2 * self.off: 2 * (self.off + take)] = iq[2 * pos: 2 * (pos + take)]
            self.off += take
            pos += take
            if self.off == self.n:
                re = self.inbuf[0::2] * self.window
                im = self.inbuf[1::2] * self.window
                self.out = np.fft.fft(re.astype(np.float64) + 1j * im.astype(np.float64))
                self.off = 0
                self.frames += 1

    def get_db(self):
        """getSpectrum: 10 * log10f(re^2 + im^2) - 20 * log10f((float)N), fft-shifted."""
        re, im = self.out.real.astype(F32), self.out.imag.astype(F32)
        p = re * re + im * im
        with np.errstate(divide="ignore"):
            db = (F32(10) * np.log10(p.astype(np.float64)).astype(F32)).astype(F32)
        scaledb = F32(20) * F32(np.log10(np.float64(F32(self.n))))
        return np.fft.fftshift(db - scaledb).astype(F32)
