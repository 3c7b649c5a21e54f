"""The cases the reference's FFTW-calling blocks are run on (oracle/ref_chain.cxx), shared by the golden
generator (tests/golden/make_reference_chain_golden.py, on the GPU box), the CPU test that holds the oracle
against the committed vectors and the GPU test that re-runs the reference live.

Every input is built from integers only -- a seeded numpy Generator (bit-reproducible on any host) and
carriers read from a 4096-entry integer sine table -- and quantised
to a grid float32 represents exactly, so that the two boxes feed the reference and the oracle the very same
bits without the inputs having to be stored."""
import numpy as np

AM, FM, USB, LSB = range(4)

# (passband Hz, input rate Hz): maxbin 0 (all-zero taps), 1, 2, 3, 7, 15, 16, 32 and C5's 32-bit edge (lowpass.cxx:167)
LOWPASS = [(100, 48_000), (8_000, 256_000), (80_000, 2_048_000), (6_400_000, 100_000_000), (64_000_000, 1_000_000_000),
           (200_000, 2_048_000), (500_000, 2_048_000), (1_000_000, 2_048_000), (1_024_000, 2_048_000),
           (2_048_000, 2_048_000), (8_000, 48_000), (80_000, 2_400_000)]

CHAINS = {
    # BASELINE config 1's parameters (C1), four blocks
    "c1_fm": dict(fs=2_048_000, if_hz=100_000, cpb=80_000, crate=256_000, mode=FM, apb=8_000, arate=32_000,
                  block=16_384, blocks=4, carriers=[(100_000, 0.45, 1_000, 5.0)], seed=11),
    # BASELINE config 2's parameters for one of its channels (c = 5), three blocks
    "c2_fm": dict(fs=100_000_000, if_hz=-39_843_750 + 5 * 312_500, cpb=6_400_000, crate=250_000, mode=FM, apb=8_000,
                  arate=50_000, block=16_000, blocks=3, carriers=[(-39_843_750 + 5 * 312_500, 0.3, 700, 3.0),
                                                                  (-39_843_750 + 9 * 312_500, 0.2, 900, 2.0)], seed=12),
    # radio.cxx:78-81's defaults (80 k / 240 k / 8 k / 48 k) off 2.4 Msps, the other three modes, negative IF
    "am": dict(fs=2_400_000, if_hz=-250_000, cpb=80_000, crate=240_000, mode=AM, apb=8_000, arate=48_000,
               block=9_600, blocks=3, carriers=[(-250_000, 0.4, 500, 0.5), (-240_000, 0.1, 0, 0.0)], seed=13),
    "usb": dict(fs=2_400_000, if_hz=-250_000, cpb=80_000, crate=240_000, mode=USB, apb=8_000, arate=48_000,
                block=9_600, blocks=3, carriers=[(-250_000, 0.4, 500, 0.5), (-240_000, 0.1, 0, 0.0)], seed=13),
    "lsb": dict(fs=2_400_000, if_hz=-250_000, cpb=80_000, crate=240_000, mode=LSB, apb=8_000, arate=48_000,
                block=9_600, blocks=3, carriers=[(-250_000, 0.4, 500, 0.5), (-240_000, 0.1, 0, 0.0)], seed=13),
}

MIXES = {
    "if_pos": dict(fs=2_400_000, if_hz=100_000, block=10_000, blocks=3, seed=21),
    "if_neg_c2": dict(fs=100_000_000, if_hz=-39_843_750, block=10_000, blocks=3, seed=22),
    "if_zero": dict(fs=2_048_000, if_hz=0, block=4_096, blocks=2, seed=23),
}

SPECTRA = {
    # SURVEY's probe: 512 points, a tone at +100 kHz off 2.4 Msps -> peak in bin 277
    "n512_tone": dict(fs=2_400_000, n=512, block=512, blocks=2, carriers=[(100_000, 0.2, 0, 0.0)], noise=0.0, seed=31),
    "n4096": dict(fs=2_400_000, n=4_096, block=1_024, blocks=9, carriers=[(300_000, 0.3, 2_000, 4.0), (-700_000, 0.05, 0, 0.0)],
                  noise=0.01, seed=32),
    # BASELINE config 3's size
    "n65536": dict(fs=100_000_000, n=65_536, block=32_768, blocks=2, carriers=[(12_500_000, 0.25, 10_000, 8.0),
                                                                             (-31_000_000, 0.02, 0, 0.0)], noise=0.005, seed=33),
}

_Q = 1 << 15                       # input grid: multiples of 2^-15 (exact in float32)
_TBL_BITS = 12


def _int_sine_table():
    """round(2^30 * sin(2 pi k / 4096)): double-precision libm values rounded to a grid 10^7 times coarser than
    their own error, so every host makes the same integers"""
    import math
    n = 1 << _TBL_BITS
    return np.array([int(round(math.sin(2.0 * math.pi * k / n) * (1 << 30))) for k in range(n)], np.int64)


_TBL = None


def _cis_int(phase32):
    """(cos, sin) * 2^30 of the angle 2 pi phase32 / 2^32, from the integer table (nearest entry)"""
    global _TBL
    if _TBL is None:
        _TBL = _int_sine_table()
    n = 1 << _TBL_BITS
    idx = ((phase32 + (1 << (31 - _TBL_BITS))) >> (32 - _TBL_BITS)) & (n - 1)
    return _TBL[(idx + n // 4) & (n - 1)], _TBL[idx]


def synth_iq(nframes, fs, carriers, noise, seed):
    """interleaved float32 IQ on the 2^-15 grid: carriers (IF Hz, amplitude, modulating Hz, beta) + uniform noise"""
    rng = np.random.default_rng(seed)
    acc_i = np.zeros(nframes, np.int64)
    acc_q = np.zeros(nframes, np.int64)
    n = np.arange(nframes, dtype=np.int64)
    for (f, amp, fm, beta) in carriers:
        step = int(round(f * (1 << 32) / fs)) & 0xFFFFFFFF
        ph = (n * step) & 0xFFFFFFFF
        if fm:
            mstep = int(round(fm * (1 << 32) / fs)) & 0xFFFFFFFF
            _, ms = _cis_int((n * mstep) & 0xFFFFFFFF)
            dev = int(round(beta * (1 << 32) / (2 * 3.141592653589793)))          # beta radians in phase units
            ph = (ph + ((ms * dev) >> 30)) & 0xFFFFFFFF
        c, s = _cis_int(ph)
        a = int(round(amp * _Q))
        acc_i += (c * a) >> 30
        acc_q += (s * a) >> 30
    if noise > 0:
        w = max(1, int(round(noise * _Q)))
        acc_i += rng.integers(-w, w + 1, nframes)
        acc_q += rng.integers(-w, w + 1, nframes)
    iq = np.empty(2 * nframes, np.float32)
    iq[0::2] = acc_i.astype(np.float32) / np.float32(_Q)
    iq[1::2] = acc_q.astype(np.float32) / np.float32(_Q)
    return iq


def chain_input(c):
    return synth_iq(c["block"] * c["blocks"], c["fs"], c["carriers"], 0.004, c["seed"])


def mix_input(c):
    rng = np.random.default_rng(c["seed"])
    q = rng.integers(-_Q + 1, _Q, 2 * c["block"] * c["blocks"])
    return q.astype(np.float32) / np.float32(_Q)


def spectrum_input(c):
    return synth_iq(c["block"] * c["blocks"], c["fs"], c["carriers"], c["noise"], c["seed"])


def sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


# ---- tolerances: what separates the reference run over hipFFTW from the oracle / the HIP path ------------------------
TAPS_TOL = 1e-7          # the 64-point inverse DFT: rocFFT's f32 butterflies vs the oracle's double-accumulated DFT; the
                         # impulse is a sum of up to 63 unit cosines before the window scales it by <= 1/64, so an f32
                         # FFT is good to a few 1e-8 on the widest filters (measured: profiles/r04_reference_pin.txt)
CHAN_TOL = 1e-6          # channel IQ (the oracle's own bar for the fast NCO modes)
AUDIO_TOL = 1e-5
DB_TOL = 0.01            # dB, on the bins within DB_MASK of the frame's peak: SURVEY 8c's own bar (0.01 dB within 80 dB).  Measured
DB_MASK = 80.0           # against the live reference (profiles/r06_reference_pin.txt): <= 3.4e-4 dB within 60 dB, <= 1.2e-3 within 70,
                         # <= 3.8e-3 within 80 (65536 points: f32 butterflies leave ~3e-7 of the peak on every bin)


# ---- full-size cases, LIVE only (the inputs are regenerated on the box; nothing of this size is committed) -------------
# BASELINE config 2 at its full block size and config 5's parameters: a few of the 256 receivers through the reference's
# own chain, the whole 256-receiver tuner through the HIP path.
def _c2_if(c, if0=-39_843_750, step=312_500):
    return if0 + c * step


FULL = {
    "c2_full": dict(fs=100_000_000, cpb=6_400_000, crate=250_000, mode=FM, apb=8_000, arate=50_000, block=4_000_000, blocks=1,
                    channels=256, if0=-39_843_750, if_step=312_500, probe=[0, 5, 129, 254],
                    modes={5: AM, 129: USB, 254: LSB},          # (every other receiver: FM)
                    carriers=[(_c2_if(0), 0.11, 700, 3.0), (_c2_if(5), 0.12, 900, 2.0), (_c2_if(129), 0.1, 1_100, 4.0),
                              (_c2_if(254), 0.09, 500, 2.5), (_c2_if(77), 0.1, 0, 0.0)], seed=41),
    "c5_chunks": dict(fs=1_000_000_000, cpb=64_000_000, crate=250_000, mode=FM, apb=8_000, arate=50_000, block=560_000, blocks=2,
                      channels=256, if0=-398_437_500, if_step=3_125_000, probe=[3, 200],
                      carriers=[(-398_437_500 + 3 * 3_125_000, 0.2, 3_000, 2.0), (-398_437_500 + 200 * 3_125_000, 0.2, 4_000, 2.0)],
                      seed=42),
}


def full_input(c):
    return synth_iq(c["block"] * c["blocks"], c["fs"], c["carriers"], 0.004, c["seed"])


def full_ifs(c):
    return [c["if0"] + i * c["if_step"] for i in range(c["channels"])]


# BASELINE config 3 at its full size: the waterfall of one 4 000 000-frame block (65536 points every 32768 frames, 121 rows);
# these rows of it through the reference's own SpectrumSink, fed the row's 65536 frames
C3_FULL = dict(case="c2_full", n=65_536, hop=32_768, rows=[0, 1, 57, 120])


def full_mode(c, ch):
    return c.get("modes", {}).get(ch, c["mode"])
