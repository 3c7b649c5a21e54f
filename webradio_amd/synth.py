"""Synthetic tuner streams: FM carriers on a channel raster plus
seeded white noise, as complex float32 interleaved IQ.  Used by tests (numpy, on the
host, small) and by bench.py (torch, generated directly in HBM).  Data generation
only -- no part of the DSP path.
"""
import numpy as np

# BASELINE config 2 ("C2"): 256 channels on a 312.5 kHz raster off 100 Msps
C2 = dict(
    input_rate=100_000_000,
    channels=256,
    if0=-39_843_750,
    if_step=312_500,
    chan_passband=6_400_000,
    chan_rate=250_000,        # D1 = 400
    audio_passband=8_000,
    audio_rate=50_000,        # D2 = 5
    block_frames=4_000_000,   # 40 ms
)

# BASELINE config 5 ("C5"): ONE 1 Gsps stream, otherwise C2 with D1 = 4000; the passband
# 64 MHz is the largest whose product 64 * passband still fits 32 bits (lowpass.cxx:167, Q6)
C5 = dict(
    input_rate=1_000_000_000,
    channels=256,
    if0=-398_437_500,
    if_step=3_125_000,
    chan_passband=64_000_000,
    chan_rate=250_000,        # D1 = 4000
    audio_passband=8_000,
    audio_rate=50_000,        # D2 = 5
    block_frames=20_000_000,  # chunk T: 20 ms, a multiple of D1 * D2 = 20 000
)

# BASELINE config 1 ("C1"): one receiver off a 2.048 Msps RTL-SDR style stream
C1 = dict(
    input_rate=2_048_000,
    if_hz=100_000,
    chan_passband=80_000,
    chan_rate=256_000,        # D1 = 8
    audio_passband=8_000,
    audio_rate=32_000,        # D2 = 8
    block_frames=131_072,
)


def c2_ifs(n=None, cfg=C2):
    n = cfg["channels"] if n is None else n
    return [cfg["if0"] + c * cfg["if_step"] for c in range(n)]


def fm_stream(nframes, input_rate, carrier_ifs, start_frame=0, amp=None, beta=5.0,
              fm_base=300.0, fm_step=10.0, noise_dbfs=-40.0, seed=12345):
    """Sum of FM carriers A*exp(j(2*pi*IF*t + beta*sin(2*pi*fm*t))) + white noise.

    Returns interleaved float32 IQ of length 2*nframes.  Deterministic in
    (start_frame, seed): consecutive calls with advancing start_frame give one
    continuous stream.
    """
    t = (np.arange(nframes, dtype=np.float64) + start_frame) / float(input_rate)
    sig = np.zeros(nframes, dtype=np.complex128)
    ncar = max(len(carrier_ifs), 1)
    a = (0.5 / ncar) if amp is None else amp
    for idx, f in enumerate(carrier_ifs):
        fm = fm_base + fm_step * idx
        sig += a * np.exp(1j * (2 * np.pi * f * t + beta * np.sin(2 * np.pi * fm * t)))
    rng = np.random.default_rng(seed + start_frame)
    scale = 10.0 ** (noise_dbfs / 20.0)
    noise = (rng.standard_normal(nframes) + 1j * rng.standard_normal(nframes)) * (scale / np.sqrt(2))
    sig += noise
    out = np.empty(2 * nframes, dtype=np.float32)
    out[0::2] = sig.real
    out[1::2] = sig.imag
    return out


def rtl_u8_stream(nframes, input_rate=2_048_000, carrier_if=100_000, tone=1000.0, beta=5.0, seed=7):
    """An RTL-SDR format capture (unsigned 8-bit interleaved IQ) of one FM carrier + noise,
    quantised round(127.5 + 127*x)."""
    t = np.arange(nframes, dtype=np.float64) / float(input_rate)
    z = 0.6 * np.exp(1j * (2 * np.pi * carrier_if * t + beta * np.sin(2 * np.pi * tone * t)))
    rng = np.random.default_rng(seed)
    z += (rng.standard_normal(nframes) + 1j * rng.standard_normal(nframes)) * 0.02
    x = np.empty(2 * nframes, dtype=np.float64)
    x[0::2] = z.real
    x[1::2] = z.imag
    q = np.clip(np.round(127.5 + 127.0 * x), 0, 255).astype(np.uint8)
    return q


def fm_stream_torch(nframes, input_rate, carrier_ifs, device, start_frame=0, beta=5.0,
                    fm_base=300.0, fm_step=10.0, noise_dbfs=-40.0, seed=12345, chunk=1 << 20):
    """Same signal family generated on the GPU with torch (bench input; float64 phase)."""
    import torch

    out = torch.empty(2 * nframes, dtype=torch.float32, device=device)
    ncar = max(len(carrier_ifs), 1)
    a = 0.5 / ncar
    f = torch.tensor(list(carrier_ifs), dtype=torch.float64, device=device)[:, None]
    fm = (fm_base + fm_step * torch.arange(len(carrier_ifs), dtype=torch.float64, device=device))[:, None]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    scale = 10.0 ** (noise_dbfs / 20.0) / (2 ** 0.5)
    for s in range(0, nframes, chunk):
        e = min(s + chunk, nframes)
        t = (torch.arange(s, e, dtype=torch.float64, device=device) + start_frame) / float(input_rate)
        ph = 2 * torch.pi * f * t[None, :] + beta * torch.sin(2 * torch.pi * fm * t[None, :])
        # reduce the phase before the float32 trig
        ph = torch.remainder(ph, 2 * torch.pi).to(torch.float32)
        re = a * torch.cos(ph).sum(0) + scale * torch.randn(e - s, device=device, generator=g)
        im = a * torch.sin(ph).sum(0) + scale * torch.randn(e - s, device=device, generator=g)
        out[2 * s:2 * e:2] = re
        out[2 * s + 1:2 * e:2] = im
    return out
