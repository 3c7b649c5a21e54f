import sys, time, os, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
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
lib = capi.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(8):                      # clocks up
    for i in range(100): t.submit_device(blocks[i % nb], n)
    t.flush(); torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    for i in range(K): t.submit_device(blocks[i % nb], n)
    t.flush(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.wr_debug_stream_tl(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    A = buf.reshape(8192, 8).astype(np.float64)
    t00 = A[6999, 0]
    d = [(A[7000 + j, 0] - t00) / 100 for j in range(K)]; p = [(A[7000 + j, 1] - t00) / 100 for j in range(K)]
    print("K=%d wall %.1f us = %.2f us/blk; long %d" % (K, wall, wall / K, t.stream_long_blocks()))
    print("  ddc_ready:", " ".join("%.0f" % v for v in d))
    print("  post_ready:", " ".join("%.0f" % v for v in p))
    print("  last post - last ddc = %.1f; wall - last post = %.1f (launch, wake-up, state roll, exit, sync)" % (p[-1] - d[-1], wall - p[-1]))
t.destroy()
