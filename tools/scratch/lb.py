import sys, time
sys.path.insert(0, '.')
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner
cfg = synth.C2; n = cfg["block_frames"]; ifs = synth.c2_ifs(256); nb = 4
x = synth.fm_stream_torch(n * nb, cfg["input_rate"], ifs[::4], "cuda", seed=1)
blocks = [x[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
dev = Device(0, torch.cuda.current_stream().cuda_stream)
t = Tuner(dev, cfg["input_rate"], 256, n, capi.WR_NCO_ROTATE)
for f in ifs:
    t.add_receiver(f, cfg["chan_passband"], cfg["chan_rate"], capi.WR_FM, cfg["audio_passband"], cfg["audio_rate"])
t.streaming(True)
for K, pace in ((20, 0), (20, 0.002), (9, 0)):
    for i in range(K):
        t.submit_device(blocks[i % nb], n)
        if pace: time.sleep(pace)
    t.flush(); dev.sync()
    a = t.fetch_audio_all()
    print(K, pace, t.stream_info(), t.stream_long_blocks())
t.destroy()
