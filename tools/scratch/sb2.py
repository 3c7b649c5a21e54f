"""development: one streaming launch of K C2 blocks, timed, and (WR_STREAM_DBG & 16) where the waves' time went"""
import sys, time, os
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for rep in range(reps):
    t.flush(); dev.sync()
    t0 = time.perf_counter()
    for i in range(K):
        t.submit_device(blocks[i % nb], n)
    t.flush(); dev.sync()
    t2 = time.perf_counter()
    print("dbg=%s K=%d total %.2f us/blk" % (os.environ.get("WR_STREAM_DBG", "0"), K, (t2 - t0) / K * 1e6), t.stream_info(), flush=True)
if int(os.environ.get("WR_STREAM_DBG", "0")) & 16:
    import ctypes
    lib = capi.load()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.wr_debug_stream_tl(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    a = buf.reshape(8192, 8)[:8000]
    a = a[(a[:, 0] + a[:, 1] + a[:, 4]) > 0]
    a = a[a[:, 0] < 10**9]
    f = a.astype(np.float64)
    tot = f[:, :5].sum(axis=1) + f[:, 7]
    print("waves", len(a), "total us median %.0f" % (np.median(tot) / 100))
    for nm, c in (("unit", 0), ("task", 1), ("gate", 2), ("poll", 3), ("gpoll", 7), ("other", 4)):
        print("  %-5s us per wave: p10 %.0f median %.0f p90 %.0f max %.0f" % (nm, *(np.percentile(f[:, c], q) / 100 for q in (10, 50, 90, 100))))
    u = np.maximum(f[:, 5], 1); tk = np.maximum(f[:, 6], 1)
    print("  units/wave median %.0f  us/unit p10 %.2f median %.2f p90 %.2f" % (np.median(f[:, 5]), *(np.percentile(f[:, 0] / u / 100, q) for q in (10, 50, 90))))
    print("  tasks/wave median %.0f  us/task p10 %.2f median %.2f p90 %.2f max %.2f" % (np.median(f[:, 6]), *(np.percentile(f[:, 1] / tk / 100, q) for q in (10, 50, 90, 100))))
    w8 = np.arange(len(a)) % 8
    for k in range(8):
        m = w8 == k
        print("  wave %d of its workgroup: unit %.0f task %.0f gate %.0f other %.0f us; us/task %.2f" % (k, *(np.median(f[m, c]) / 100 for c in (0, 1, 2, 4)), np.median(f[m, 1] / tk[m]) / 100))
t.destroy()
