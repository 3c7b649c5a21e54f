import sys, time, os
sys.path.insert(0, '.')
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner
cfg = synth.C2; n = cfg["block_frames"]; ifs = synth.c2_ifs(256); nb = 12
x = synth.fm_stream_torch(n * nb, cfg["input_rate"], ifs[::4], "cuda", seed=1)
blocks = [x[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
dev = Device(0, torch.cuda.current_stream().cuda_stream)
t = Tuner(dev, cfg["input_rate"], 256, n, capi.WR_NCO_ROTATE)
for f in ifs:
    t.add_receiver(f, cfg["chan_passband"], cfg["chan_rate"], capi.WR_FM, cfg["audio_passband"], cfg["audio_rate"])
t.streaming(True)
for K in (200, 1, 2, 5, 10, 20, 50, 100):
    for rep in range(3):
        t.flush(); torch.cuda.synchronize()
        t.profile(1)
        t0 = time.perf_counter()
        for i in range(K):
            t.submit_device(blocks[i % nb], n)
        t.flush()
        if os.environ.get("DEVSYNC"): dev.sync()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        got, ms = t.profile_read(); t.profile(False)
    print("K=%3d wall %.1f us total, kernel %.1f us (%d launch) -> fixed part vs 200-block rate" % (K, (t2 - t0) * 1e6, ms * 1e3, got), flush=True)
t.destroy()
