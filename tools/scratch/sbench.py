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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for rep in range(6):
    t.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        t.submit_device(blocks[i % nb], n)
    t1 = time.perf_counter()
    t.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("dbg=%s K=%d submit loop %.1f us/blk, total %.2f us/blk" % (os.environ.get("WR_STREAM_DBG", "0"), K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), t.stream_info(), flush=True)
t.destroy()
if int(os.environ.get("WR_STREAM_DBG", "0")) & 16:
    import ctypes, numpy as np
    lib = capi.load()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.wr_debug_stream_tl(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    a = buf.reshape(8192, 8)
    nzr = np.nonzero(a.any(axis=1))[0]; print("nonzero rows:", len(nzr), nzr[:3], nzr[-5:], a[6001])
    pw = a[6000:6300].astype(np.float64); pw = pw[pw[:, 2] > 0]
    if len(pw):
        print("post WGs", len(pw), "blocks", np.median(pw[:, 2]), "cycles per block: wait for the block median %.0f p90 %.0f; work+sync median %.0f p90 %.0f max %.0f" % (np.median(pw[:, 0] / pw[:, 2]), np.percentile(pw[:, 0] / pw[:, 2], 90), np.median(pw[:, 1] / pw[:, 2]), np.percentile(pw[:, 1] / pw[:, 2], 90), (pw[:, 1] / pw[:, 2]).max()))
    a = a[:6000]
    a = a[a[:, 3] > 0].astype(np.float64)
    u = a[:, 3]
    print("waves", len(a), "units/wave median", np.median(u))
    for nm, col in (("taps", 0), ("wait", 1), ("rest", 2)):
        per = a[:, col] / u
        print("  %s cycles/unit: p10 %.0f median %.0f p90 %.0f max %.0f" % (nm, np.percentile(per, 10), np.median(per), np.percentile(per, 90), per.max()))
    ringn = (a[:, 7].astype(np.uint64) >> np.uint64(32)).astype(np.float64); ep = (a[:, 7].astype(np.uint64) & np.uint64(0xffffffff)).astype(np.float64)
    print("  ring waits per wave median %.0f (of ~%d blocks), cycles per wait median %.0f, ring cycles/unit median %.0f; epochs median %.0f" % (np.median(ringn), 200, np.median(a[:, 6] / np.maximum(ringn, 1)), np.median(a[:, 6] / u), np.median(ep)))
    tot = (a[:, 0] + a[:, 1] + a[:, 2]) / u
    print("  total cycles/unit median %.0f -> x %.1f units/blk/wave" % (np.median(tot), 20000 / 4088 * 1.0))
    hw = a[:, 4].astype(np.int64); xcc = a[:, 5].astype(np.int64) & 15
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    import collections
    percu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    print("CUs with DDC waves:", len(percu), "waves per CU histogram:", sorted(collections.Counter(percu.values()).items()))
    persimd = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    print("waves per SIMD histogram:", sorted(collections.Counter(persimd.values()).items()))
    ring = a[:, 6]
    for xq in range(8):
        m = xcc == xq
        print("  XCD %d: waves %d ring-wait cycles per wave median %.0f p10 %.0f p90 %.0f ; taps/unit median %.0f" % (xq, m.sum(), np.median(ring[m]), np.percentile(ring[m], 10), np.percentile(ring[m], 90), np.median(a[m, 0] / u[m])))
    # per CU: min ring wait (the slowest waves wait least)
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    cus = {}
    for kk, r in zip(key.tolist(), ring.tolist()):
        cus.setdefault(kk, []).append(r)
    med = np.array([np.median(v) for v in cus.values()])
    print("  per-CU median ring wait: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f  (total cycles per wave ~ %.0f)" % (med.min(), np.percentile(med, 10), np.median(med), np.percentile(med, 90), med.max(), np.median((a[:, 0] + a[:, 1] + a[:, 2]))))
    A = buf.reshape(8192, 8).astype(np.float64)
    t00 = A[7000 + 100, 0]
    print("block: ddc_ready post_ready | postWG0 start end | postWG1 start end | DDC wave0 enter (after wait) | wave1000 | wave3000   [us rel.]")
    for jj in range(100, 108):
        f = lambda v: "%8.1f" % ((v - t00) / 100.0) if v else "    -   "
        print(jj, f(A[7000 + jj, 0]), f(A[7000 + jj, 1]), "|", f(A[7300 + jj, 0]), f(A[7300 + jj, 1]), "|", f(A[7300 + jj, 2]), f(A[7300 + jj, 3]), "|", f(A[7600 + jj, 0]), f(A[7600 + jj, 4]), "|", f(A[7600 + jj, 1]), f(A[7600 + jj, 5]), "|", f(A[7600 + jj, 3]), f(A[7600 + jj, 7]))
    widx = np.nonzero(buf.reshape(8192, 8)[:6000, 3] > 0)[0]
    for w8 in range(8):
        m = (widx % 8) == w8
        print("  wave %d of its workgroup: ring-wait cycles median %.0f p10 %.0f ; ring waits median %.0f ; taps/unit median %.0f rest/unit %.0f" % (w8, np.median(ring[m]), np.percentile(ring[m], 10), np.median(ringn[m]), np.median(a[m, 0] / u[m]), np.median(a[m, 2] / u[m])))
