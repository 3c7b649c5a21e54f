"""Secondary measurements quoted in DESIGN.md (bench.py stays the single contract line):
  1. C2 with the block handed over in HOST memory (PCIe-inclusive rate)
  2. C3: SpectrumSink waterfall, 65536-point FFT, 50 % overlap, off a resident stream
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner, Spectrum

c2 = synth.C2
fs, n = c2["input_rate"], c2["block_frames"]
ifs = synth.c2_ifs()
dev = Device(0, torch.cuda.current_stream().cuda_stream)
x = synth.fm_stream_torch(n, fs, ifs[::4], "cuda")
torch.cuda.synchronize()
out = {}

# 1. host-resident input: pageable and pinned
t = Tuner(dev, fs, 256, n)
for f in ifs:
    t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"], c2["audio_rate"])
xh = x.cpu()
xp = xh.pin_memory()
for name, buf in (("pageable", xh), ("pinned", xp)):
    for _ in range(2):
        capi.check(t.lib.wr_tuner_submit(t.h, capi.ptr(buf), n, capi.WR_HOST))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        capi.check(t.lib.wr_tuner_submit(t.h, capi.ptr(buf), n, capi.WR_HOST))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["c2_host_%s_msps" % name] = round(n / dt / 1e6, 1)
    out["c2_host_%s_ms" % name] = round(dt * 1e3, 3)
# 1b. the same stream in the RTL-SDR byte format (2 B per frame): PCIe-inclusive and resident
u8 = ((x.clamp(-1, 1) * 127.0) + 127.5).round().clamp(0, 255).to(torch.uint8)
u8h = u8.cpu()
for name, buf, where in (("host_u8", u8h, capi.WR_HOST), ("resident_u8", u8, capi.WR_DEVICE)):
    for _ in range(2):
        capi.check(t.lib.wr_tuner_submit_u8(t.h, capi.ptr(buf), n, where))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        capi.check(t.lib.wr_tuner_submit_u8(t.h, capi.ptr(buf), n, where))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["c2_%s_msps" % name] = round(n / dt / 1e6, 1)
    out["c2_%s_ms" % name] = round(dt * 1e3, 3)
t.destroy()

# 2. C3 waterfall
N, hop = 65536, 32768
rows = (n - N) // hop + 1
s = Spectrum(dev, N, hop)
db = torch.empty(rows * N, dtype=torch.float32, device="cuda")
for _ in range(2):
    s.batch_db(x, rows, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    s.batch_db(x, rows, db)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
out["c3_rows"] = rows
out["c3_ms_per_block"] = round(ms, 4)
out["c3_frames_per_s"] = round(rows / ms * 1e3, 1)
out["c3_msps_new_samples"] = round(rows * hop / ms / 1e3, 1)
out["c3_algorithmic_GBps"] = round(rows * 786432 / ms / 1e6, 1)
out["c3_frac_hbm"] = round(rows * 786432 / ms / 1e6 / 8000, 4)
s.destroy()
print(json.dumps(out))
