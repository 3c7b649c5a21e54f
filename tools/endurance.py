"""Sustained throughput of the C2 hot path: the bench's steady-state loop (256 receivers, 12 resident 4 M-frame blocks cycled,
four blocks per launch) for minutes on end, tuner-input Gsps per 10-second window -- does the rate hold once the part is hot?
usage: python tools/endurance.py [seconds=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
c2 = synth.C2
fs, n, B, nb = c2["input_rate"], c2["block_frames"], 4, 12
ifs = synth.c2_ifs(256)
dev = Device(0, torch.cuda.current_stream().cuda_stream)
xs = synth.fm_stream_torch(n * nb, fs, ifs[::4], "cuda")
blocks = [xs[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
t = Tuner(dev, fs, 256, n * B, capi.WR_NCO_ROTATE)
for f in ifs:
    t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"], c2["audio_rate"])
t.blocks_per_launch(B)
torch.cuda.synchronize()
t_start = time.perf_counter()
step, rows = 0, []
while time.perf_counter() - t_start < secs:
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 10.0:
        for _ in range(1200):                      # ~36 ms of launches per host check
            t.submit_device(blocks[step % nb], n)
            step += 1
        k += 1200
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows.append(k * n / dt / 1e9)
    print("t = %5.0f s: %.1f Gsps (%.2f us per 4 M-frame block)" % (time.perf_counter() - t_start, rows[-1], dt / k * 1e6), flush=True)
t.flush()
a = t.fetch(0, capi.WR_STAGE_AUDIO, n * B)
assert a.size and bool((a == a).all())
print("%d windows: min %.1f  mean %.1f  max %.1f Gsps; first %.1f, last %.1f" % (len(rows), min(rows), sum(rows) / len(rows), max(rows), rows[0], rows[-1]))
t.destroy()
