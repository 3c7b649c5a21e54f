"""What the package draws (rocm-smi, sampled from a thread every 0.2 s) while the GPU runs: nothing; the streaming launch back to
back (C2, 400-block launches); the same with the post stage's body switched off (WR_STREAM_DBG=1: results wrong); the isolated
tap mix (tools/ubench_tap, built by hand: see its first line); a plain HBM copy.  -> profiles/r06_power.txt
    python tools/power_probe.py"""
import sys, time, os, subprocess, threading, re
sys.path.insert(0, '.')
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner

samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.perf_counter(), o))
        except Exception as e:
            samples.append((time.perf_counter(), repr(e)))
        time.sleep(0.2)

cfg = synth.C2; n = cfg["block_frames"]; ifs = synth.c2_ifs(256); nb = 12
x = synth.fm_stream_torch(n * nb, cfg["input_rate"], ifs[::4], "cuda", seed=1)
blocks = [x[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
dev = Device(0, torch.cuda.current_stream().cuda_stream)
t = Tuner(dev, cfg["input_rate"], 256, n, capi.WR_NCO_ROTATE)
for f in ifs:
    t.add_receiver(f, cfg["chan_passband"], cfg["chan_rate"], capi.WR_FM, cfg["audio_passband"], cfg["audio_rate"])
t.streaming(True)
th = threading.Thread(target=sampler); th.start()
marks = []
time.sleep(2.0); marks.append(("idle", 0, time.perf_counter()))
K = 400
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 12.0:
    for i in range(K):
        t.submit_device(blocks[i % nb], n)
    t.flush(); torch.cuda.synchronize(); reps += 1
t1 = time.perf_counter()
marks.append(("stream", (t1 - t0) / (reps * K) * 1e6, t1))
os.environ["WR_STREAM_DBG"] = "1"
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 8.0:
    for i in range(K):
        t.submit_device(blocks[i % nb], n)
    t.flush(); torch.cuda.synchronize(); reps += 1
t1 = time.perf_counter()
marks.append(("stream, no post body", (t1 - t0) / (reps * K) * 1e6, t1))
del os.environ["WR_STREAM_DBG"]
t.destroy()
if os.path.exists("tools/ubench_tap"):
    for _ in range(5):
        o = subprocess.run(["tools/ubench_tap"], capture_output=True, text=True, timeout=120).stdout
    marks.append(("ubench_tap x 5", 0, time.perf_counter()))
    print(o[-1500:])
# a plain memory stream for comparison: torch copy of 1 GB back and forth
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
t0 = time.perf_counter(); ncopy = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(50):
        b.copy_(a)
    torch.cuda.synchronize(); ncopy += 50
t1 = time.perf_counter()
marks.append(("hbm copy %.2f TB/s (read + written)" % (ncopy * 2.0 * a.numel() * 4 / (t1 - t0) / 1e12), 0, t1))
time.sleep(1.0)
stop = True; th.join()
prev = samples[0][0]
for name, us, tend in marks:
    sel = [s for (ts, s) in samples if prev < ts <= tend]
    pw = []; ck = []
    for s in sel:
        m = re.findall(r'"(?:Average Graphics Package Power|Current Socket Graphics Package Power) \(W\)": "([0-9.]+)"', s)
        pw += [float(v) for v in m]
        m = re.findall(r'"sclk clock speed:": "\(([0-9]+)Mhz\)"', s)
        ck += [int(v) for v in m]
    print("%-42s %s samples %d  power W: %s  sclk MHz: %s" % (name, ("%.2f us/blk" % us) if us else "", len(sel), (min(pw), sum(pw) / len(pw), max(pw)) if pw else None, (min(ck), sum(ck) / len(ck), max(ck)) if ck else None))
    prev = tend
print("a sample during the stream:", samples[len(samples) // 5][1][:600])
