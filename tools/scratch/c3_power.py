import sys, time, subprocess, threading, re
sys.path.insert(0, '.')
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Spectrum
cfg = synth.C2; n = cfg["block_frames"]; nb = 4
x = synth.fm_stream_torch(n * nb, cfg["input_rate"], synth.c2_ifs(256)[::4], "cuda", seed=1)
blocks = [x[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
dev = Device(0, torch.cuda.current_stream().cuda_stream)
N, HOP = 65536, 32768
rows = (n - N) // HOP + 1
outs = [torch.empty(rows * N, dtype=torch.float32, device="cuda") for _ in range(2)]
spec = Spectrum(dev, N, HOP)
samples = []; stop = [False]
def sampler():
    while not stop[0]:
        o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
        w = re.findall(r'Package Power \(W\)": "([0-9.]+)"', o); c = re.findall(r'"sclk clock speed:": "\(([0-9]+)Mhz\)"', o)
        samples.append((time.perf_counter(), float(w[0]) if w else 0, int(c[0]) if c else 0)); time.sleep(0.1)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 6:
    for i in range(100):
        spec.batch_db(blocks[i % nb], rows, outs[i % 2])
    dev.sync(); k += 100
t1 = time.perf_counter()
stop[0] = True; th.join()
w = [s for s in samples if t0 + 1 < s[0] < t1]
print("C3 waterfall batch: %.2f us per block of %d rows; package W mean %.0f max %.0f; sclk mean %.0f" % ((t1 - t0) / k * 1e6, rows, sum(s[1] for s in w) / len(w), max(s[1] for s in w), sum(s[2] for s in w) / len(w)))
