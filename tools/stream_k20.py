"""-DSTREAM_TL build, WR_STREAM_DBG=16: when does each block of a K-block stream finish, from the launch's first instruction?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes, numpy as np
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = capi.load()
for rep in range(4):
    # keep the clock up: a long stream first, then the short one at once
    for i in range(3000):
        t.submit_device(blocks[i % nb], n)
    t.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        t.submit_device(blocks[i % nb], n)
    t.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.wr_debug_stream_tl(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    A = buf.reshape(8192, 8).astype(np.float64)
    s0 = A[6999, 0]
    d = (A[7000:7000 + K, 0] - s0) / 100.0
    p = (A[7000:7000 + K, 1] - s0) / 100.0
    print("K=%d wall %.1f us = %.2f us/blk; ddc_ready[us after the launch's first instruction]: %s" % (K, (t2 - t0) * 1e6, (t2 - t0) * 1e6 / K, " ".join("%.0f" % v for v in d)))
    print("    post_ready: %s" % " ".join("%.0f" % v for v in p))
    print("    ddc block times: %s ; last post_ready - last ddc_ready = %.1f" % (" ".join("%.1f" % v for v in np.diff(np.concatenate([[0], d]))), p[-1] - d[-1]))
t.destroy()
