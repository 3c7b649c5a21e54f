import sys, numpy as np
sys.path.insert(0, '.')
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner
FS, N = 2_000_000, 40_000
def ifs(n): return [(-(n // 2) + c) * 6250 + 99 for c in range(n)]
dev = Device(0)
nch = 70
iq = synth.fm_stream(6 * N, FS, ifs(nch)[::5], amp=0.1, fm_base=30.0, beta=2.0)
x = torch.from_numpy(iq).cuda(); torch.cuda.synchronize()
def run(stream, sizes):
    t = Tuner(dev, FS, nch, N, capi.WR_NCO_ROTATE)
    ch = [t.add_receiver(f, 128_000, 5_000, [capi.WR_USB, capi.WR_FM][c % 2], 160, 1_000) for c, f in enumerate(ifs(nch))]
    t.streaming(stream)
    pos = 0; out = []
    for grp in sizes:
        for fr in grp:
            t.submit_device(x[2 * pos: 2 * (pos + fr)], fr); pos += fr
        a = t.fetch_audio_all().copy()
        q = np.stack([t.fetch(c, capi.WR_STAGE_CHAN_IQ, 2 * N) for c in ch[:4]])
        out.append((a, q))
    info = t.stream_info(); t.destroy()
    return out, info
for sizes in ([[N // 2]], [[N], [N // 2]], [[N // 2, N // 2]], [[N, N], [N // 2], [N // 2, N // 2]]):
    a, _ = run(False, sizes); b, info = run(True, sizes)
    print(sizes, info)
    for i, ((ua, uq), (va, vq)) in enumerate(zip(a, b)):
        da = np.abs(ua - va).max(); dq = np.abs(uq - vq).max()
        bad = np.argwhere(uq.view(np.uint32) != vq.view(np.uint32))
        print("  group", i, "audio maxdiff", da, "iq maxdiff", dq, "first bad iq idx", bad[:6].tolist(), "n bad", len(bad), "of", uq.size)
