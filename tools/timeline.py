"""Per-wave timeline of k_tuner_ddc at C2 (development aid): needs a library built with
-DDDC_TIMELINE (tools/mkvariant.sh tl -DDDC_TIMELINE) in place of the product library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner

c2 = synth.C2
fs, n = c2["input_rate"], c2["block_frames"] * int(os.environ.get("TL_BLOCKS", "1"))      # blocks per launch
ifs = synth.c2_ifs(256)
dev = Device(0, torch.cuda.current_stream().cuda_stream)
x = synth.fm_stream_torch(n, fs, ifs[::4], "cuda")
t = Tuner(dev, fs, 256, n, capi.WR_NCO_ROTATE)
for f in ifs:
    t.add_receiver(f, c2["chan_passband"], c2["chan_rate"], capi.WR_FM, c2["audio_passband"], c2["audio_rate"])
for _ in range(int(os.environ.get('TL_WARM', '6'))):
    t.submit_device(x, n)
torch.cuda.synchronize()
lib = C.CDLL(capi.LIB_PATH)
try:
    # only the LAST launch's stamps: clear what the warm-up launches left, then two more launches (the first of
    # them without a riding post stage if the warm-up ended on a flush)
    t.submit_device(x, n)
    torch.cuda.synchronize()
    lib.wr_debug_timeline_reset()
    t.submit_device(x, n)
    torch.cuda.synchronize()
except AttributeError:
    pass
SL, NW = 12, 16384
buf = np.zeros(NW * SL, dtype=np.uint64)
rc = lib.wr_debug_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
tl = buf.reshape(NW, SL).astype(np.int64)
used = tl[:, 0] > 0
tl = tl[used]
# only the waves of the LAST launch (stamps of earlier launches with more waves linger)
nlast = int(os.environ.get("TL_WAVES", "0"))
if nlast:
    tl = tl[:nlast]
nA = int(os.environ.get("TL_A_WAVES", "0"))
t0 = tl[:, 0].min()
print("rc", rc, "waves stamped", tl.shape[0])
def stat(name, v):
    v = np.sort(v)
    print("%-34s min %8d  p10 %8d  median %8d  p90 %8d  max %8d" % (name, v[0], v[len(v) // 10], v[len(v) // 2], v[len(v) * 9 // 10], v[-1]))
stat("wave start - first start", tl[:, 0] - t0)
stat("prologue (tables, state roll)", tl[:, 1] - tl[:, 0])
stat("state load issue", tl[:, 2] - tl[:, 1])
for u in range(8):
    a = tl[:, 3 + u]; b = tl[:, 2 + u]
    ok = (a > 0) & (b > 0)
    if ok.sum():
        stat("unit %d (%d waves)" % (u, ok.sum()), (a - b)[ok])
stat("wave end - first start", tl[:, 11] - t0)
stat("wave life", tl[:, 11] - tl[:, 0])
if nA:
    stat("wave life, post-tile workgroups", (tl[:, 11] - tl[:, 0])[:nA])
    stat("wave life, the others", (tl[:, 11] - tl[:, 0])[nA:])
    stat("prologue, post-tile workgroups", (tl[:, 1] - tl[:, 0])[:nA])
print("kernel span (ticks): %d" % (tl[:, 11].max() - t0))
try:
    rb = np.zeros(NW * 2, dtype=np.uint64)
    lib.wr_debug_timeline_rt(rb.ctypes.data_as(C.c_void_p), C.c_size_t(rb.size))
    rt = rb.reshape(NW, 2).astype(np.int64)[used][: tl.shape[0]]
    life_rt = (rt[:, 1] - rt[:, 0]).astype(np.float64)
    ok = life_rt > 500                                  # waves that lived > 5 us
    clk = (tl[:, 11] - tl[:, 0])[ok] / life_rt[ok] * 100e6 / 1e9
    stat("shader clock per wave (MHz)", (clk * 1000).astype(np.int64))
    stat("wave life (ns, constant clock)", (life_rt[ok] * 10).astype(np.int64))
    print("first start to last end, constant clock: %.2f us" % ((rt[:, 1].max() - rt[:, 0][rt[:, 0] > 0].min()) / 100.0))
    # by XCD (workgroup b runs on XCD b % 8; 8-wave workgroups): when do an XCD's DDC waves end, on the constant clock?
    wpw = int(os.environ.get("TL_WAVES_PER_WG", "8"))
    wids = np.nonzero(used)[0][: tl.shape[0]]
    xcd = (wids // wpw) % 8
    t00 = rt[:, 0][rt[:, 0] > 0].min()
    for x in range(8):
        m = (xcd == x) & ok
        if m.sum():
            e = (rt[m, 1] - t00) / 100.0
            print("XCD %d: %4d waves  end of wave (us after the first start): median %.1f  p90 %.1f  max %.1f   clock median %.0f MHz  life median %.1f us" % (
                x, m.sum(), np.median(e), np.percentile(e, 90), e.max(), np.median(clk[m[ok]]) * 1000, np.median(life_rt[m]) / 100.0))
    try:
        hb = np.zeros(NW * 2, dtype=np.uint32)
        lib.wr_debug_timeline_hw(hb.ctypes.data_as(C.c_void_p), C.c_size_t(hb.size))
        hw = hb.reshape(NW, 2)[used][: tl.shape[0]]
        # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]...; XCC_ID[3:0]
        cu = ((hw[:, 1] & 15).astype(np.int64) << 12) | (((hw[:, 0] >> 13) & 7).astype(np.int64) << 8) | (((hw[:, 0] >> 12) & 1).astype(np.int64) << 4) | ((hw[:, 0] >> 8) & 15)
        simd = (hw[:, 0] >> 4) & 3
        per_cu = {}
        for c_, w_ in zip(cu[ok], wg[ok] if False else (wids // wpw)[ok]):
            per_cu.setdefault(int(c_), set()).add(int(w_))
        counts = np.array([len(v) for v in per_cu.values()])
        print("DDC workgroups per CU: %d CUs seen; %s" % (counts.size, dict(zip(*np.unique(counts, return_counts=True)))))
        lifeus = life_rt / 100.0
        for n_ in sorted(set(counts)):
            cus = {c_ for c_, v in per_cu.items() if len(v) == n_}
            m = ok & np.array([int(c_) in cus for c_ in cu])
            print("  CUs with %d DDC workgroup(s): wave life median %.1f us  p90 %.1f  (%d waves)" % (n_, np.median(lifeus[m]), np.percentile(lifeus[m], 90), m.sum()))
        for sd in range(4):
            m = ok & (simd == sd)
            print("  SIMD %d: %d waves, life median %.1f us" % (sd, m.sum(), np.median(lifeus[m])))
    except AttributeError:
        pass
    # by CU-sized neighbourhoods: the spread of last-wave end times over workgroups
    wg = wids // wpw
    ends = {}
    for w, e in zip(wg[ok], (rt[ok, 1] - t00) / 100.0):
        ends[w] = max(ends.get(w, 0.0), e)
    ev = np.array(sorted(ends.values()))
    print("last wave of each DDC workgroup ends (us): p10 %.1f  median %.1f  p90 %.1f  max %.1f  (%d workgroups)" % (
        np.percentile(ev, 10), np.median(ev), np.percentile(ev, 90), ev.max(), ev.size))
except AttributeError:
    pass
t.destroy()

# the post-stage tenants (previous block's demod + audio filter in extra workgroups of the launch)
pb = np.zeros(8192 * 4, dtype=np.uint64)
try:
    rc = lib.wr_debug_timeline_post(pb.ctypes.data_as(C.c_void_p), C.c_size_t(pb.size))
    pt = pb.reshape(8192, 4).astype(np.int64)
    pt = pt[pt[:, 0] > 0]
    print("post waves stamped", pt.shape[0])
    stat("post: wave start - first start", pt[:, 0] - t0)
    stat("post: stage phase (loads + demod)", pt[:, 1] - pt[:, 0])
    stat("post: filter phase", pt[:, 2] - pt[:, 1])
    stat("post: audio write", pt[:, 3] - pt[:, 2])
    stat("post: wave end - first start", pt[:, 3] - t0)
except AttributeError:
    pass
