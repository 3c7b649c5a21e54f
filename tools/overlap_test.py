import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner
c2 = synth.C2
fs, n = c2["input_rate"], c2["block_frames"]
ifs = synth.c2_ifs(256)
dev = Device(0, torch.cuda.current_stream().cuda_stream)
x = synth.fm_stream_torch(n, fs, ifs[::4], "cuda")
t = Tuner(dev, fs, 256, n)
for f in ifs:
    t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"], c2["audio_rate"])
for prof in (False, True):
    t.profile(prof)
    for _ in range(3):
        t.submit_device(x, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        t.submit_device(x, n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("overlap=%s profiling=%s: %.2f us/step, %.1f Msps" % (os.environ.get("WR_OVERLAP", "0") == "1", prof, dt * 1e6, n / dt / 1e6))
    if prof:
        print("   ddc event ms", t.profile_read())
