import sys, time, numpy as np
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
for rep in range(2):
    t = Tuner(dev, FS, nch, N, capi.WR_NCO_ROTATE)
    ch = [t.add_receiver(f, 128_000, 5_000, [capi.WR_USB, capi.WR_FM][c % 2], 160, 1_000) for c, f in enumerate(ifs(nch))]
    t.audio_ring(2)
    t.streaming(True)
    for b in range(4):
        t0 = time.time()
        t.submit_device(x[2 * N * b: 2 * N * (b + 1)], N)
        t1 = time.time()
        a, seq = t.ring_acquire()
        t2 = time.time()
        t.ring_release()
        print(rep, b, "submit %.1f ms acquire %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), t.stream_info(), flush=True)
    t.destroy()
