"""What a difference in channel IQ may do to FM audio -- the bound the ROTATE / SPLIT tests assert on
EVERY channel, carrier or not (noise-only channels used to be compared on IQ only).

The discriminator (dsp/demodulator.cxx:94-97) is  d[k] = atan2f(Re w, Im w) / (2 pi),
w = z[k] * conj(z[k-1]).  If z and z' differ by dz (per frame), then to first order the angle of w
moves by at most |dz[k]| / |z[k]| + |dz[k-1]| / |z[k-1]| radians: FM is as ill-conditioned as the
channel is empty, and exactly that much.  So per channel-rate frame

    b[k] = K * (|dz[k]| / |z[k]| + |dz[k-1]| / |z[k-1]|) / (2 pi) + FM_ATOL        (cycles)

with K = 1.25 covering the second-order term while the relative differences stay below 0.2 (beyond
that, or where a frame of the reference is exactly zero, the bound is the whole range: one cycle),
FM_ATOL = 2.4e-7 the in-kernel atan2 against glibc's.  The audio LowPass is linear
(dsp/lowpass.cxx:145-159): audio frame k2 differs by at most sum_j |coeff[63-j]| * b[k2*D2 - 63 + j],
plus the float rounding of 64 products and sums of values below half a cycle, taken as random:
8 * 2^-24 * sum|coeff| (2.5e-7 for the reference's 8 kHz audio filter).
A reference angle within its bound of the +-0.5 seam may come out on the other side (the filter sees
the wrap as it is): such a frame counts with the whole range too.
"""
import numpy as np

FM_ATOL = 2.4e-7
K_SECOND_ORDER = 1.25


def demod_bound(want_iq, got_iq):
    """want_iq, got_iq: interleaved float32 channel IQ of one channel over the whole run (all blocks
    concatenated, stream starting from Demodulator's zero prev_i/q).  Returns b[k] in cycles."""
    zw = want_iq[0::2].astype(np.float64) + 1j * want_iq[1::2].astype(np.float64)
    zg = got_iq[0::2].astype(np.float64) + 1j * got_iq[1::2].astype(np.float64)
    dz = np.abs(zg - zw)
    mag = np.abs(zw)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(mag > 0, dz / mag, np.where(dz > 0, np.inf, 0.0))
    prev = np.concatenate([[0.0], rel[:-1]])          # frame -1 is prev_i = prev_q = 0 on both sides: no difference
    s = rel + prev
    b = K_SECOND_ORDER * s / (2 * np.pi) + FM_ATOL
    b[(rel > 0.2) | (prev > 0.2)] = 1.0
    # a frame whose own or previous sample is exactly zero in the reference: atan2f(+-0, +-0) is decided by
    # signs of zeros -- any difference at all moves it by up to the whole range
    zero = (mag == 0)
    zprev = np.concatenate([[False], zero[:-1]])
    b[(zero | zprev) & (s > 0)] = 1.0
    return np.minimum(b, 1.0)


def audio_bound(want_iq, got_iq, taps2, d2, want_demod=None):
    """Bound on |audio_got - audio_want| per audio frame for one channel over the whole run.
    taps2: the audio LowPass's 64 coefficients (coeff[0] meets the newest sample)."""
    b = demod_bound(want_iq, got_iq)
    if want_demod is not None:
        # a reference angle within b of the +-0.5 seam may come out on the other side: whole range
        seam = 0.5 - np.abs(want_demod[: b.size].astype(np.float64))
        b = np.where(seam <= b, 1.0, b)
    a = np.abs(np.asarray(taps2, dtype=np.float64))[::-1]        # a[j] meets block[k*D + j], oldest first
    k2n = b.size // d2
    padded = np.concatenate([np.zeros(63), b])
    out = np.empty(k2n)
    for k2 in range(k2n):
        out[k2] = np.dot(a, padded[k2 * d2: k2 * d2 + 64])
    return out + 8 * 2.0 ** -24 * a.sum()


def assert_fm_audio_within_iq_bound(want_iq, got_iq, want_audio, got_audio, taps2, d2, want_demod=None, what=""):
    """Every audio frame of the channel within what its IQ difference allows.  Returns (worst difference, worst
    ratio difference / bound) for reporting."""
    bound = audio_bound(want_iq, got_iq, taps2, d2, want_demod)
    n = min(bound.size, want_audio.size, got_audio.size)
    assert n == want_audio.size == got_audio.size, (what, n, want_audio.size, got_audio.size)
    diff = np.abs(got_audio[:n].astype(np.float64) - want_audio[:n].astype(np.float64))
    bad = np.nonzero(diff > bound[:n])[0]
    assert bad.size == 0, "%s: audio frame %d differs by %.3e, its IQ difference allows %.3e (%d frames over)" % (
        what, bad[0], diff[bad[0]], bound[bad[0]], bad.size)
    return float(diff.max()), float((diff / bound[:n]).max())
