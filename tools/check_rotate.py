"""ROTATE vs EXACT NCO on the same stream: random block lengths, retunes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fs, d1, d2 = 2_400_000, 10, 5
dev = Device(0, torch.cuda.current_stream().cuda_stream)
nch = 70
ifs = rng.integers(-fs // 2, fs // 2, nch)
worst, where = 0.0, None
for uniform in (True, False):
    tuners = [Tuner(dev, fs, 128, 4000, m) for m in (capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE)]
    chans = []
    for t in tuners:
        t.lib.wr_tuner_keep_stages(t.h, 1)
        chans.append([t.add_receiver(int(f), 100_000 if uniform else int(60_000 + 1000 * i), fs // d1,
                                     capi.WR_USB, 20_000, fs // d1 // d2) for i, f in enumerate(ifs)])
    for b in range(12):
        n = int(rng.choice([10, 20, 50, 60, 70, 200, 1000, 4000]))
        x = rng.standard_normal(2 * n).astype(np.float32) * 0.3
        if b % 3 == 2:
            c = int(rng.integers(nch)); f = int(rng.integers(-fs // 2, fs // 2))
            for t, ch in zip(tuners, chans):
                t.set_if(ch[c], f)
        if b == 5:
            for t, ch in zip(tuners, chans):
                ch.append(t.add_receiver(12345, 100_000, fs // d1, capi.WR_USB, 20_000, fs // d1 // d2))
        for t in tuners:
            t.submit_host(x)
        for c in range(len(chans[0])):
            a = tuners[0].fetch(chans[0][c], capi.WR_STAGE_CHAN_IQ, 2 * (n // d1))
            r = tuners[1].fetch(chans[1][c], capi.WR_STAGE_CHAN_IQ, 2 * (n // d1))
            if a.size:
                e = float(np.abs(a - r).max())
                if e > worst:
                    worst, where = e, (b, c, n, float(np.abs(a).max()))
    for t in tuners:
        t.destroy()
    print("uniform taps" if uniform else "per-lane taps", "worst absolute difference", worst, "at (block, chan, frames, peak)", where)
