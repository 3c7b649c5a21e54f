import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner
g = np.load("tests/golden/reference_c1.npz"); c1 = synth.C1
n = int(g["block_frames"]); raw = g["u8"]; nblk = raw.size // (2 * n)
dev = Device(0)
def mk():
    t = Tuner(dev, c1["input_rate"], 1, n, capi.WR_NCO_ROTATE)
    t.add_receiver(c1["if_hz"], c1["chan_passband"], c1["chan_rate"], capi.WR_FM, c1["audio_passband"], c1["audio_rate"])
    t.audio_ring(nblk)
    return t
x = torch.from_numpy(raw).cuda(); torch.cuda.synchronize()
t = mk()
for b in range(nblk):
    t.submit_u8_device(x[2*n*b:2*n*(b+1)], n)
t.flush()
want = []
for b in range(nblk):
    a, s = t.ring_acquire(); want.append(a.copy()); t.ring_release()
t.destroy()
for nbuf in (4, 2):
    bufs = [torch.empty(2*n, dtype=torch.uint8, pin_memory=True) for _ in range(nbuf)]
    t = mk(); t.streaming(2)
    got = []
    for b in range(nblk):
        dev.lib.wr_dev_wait_uploads(dev.h)
        bufs[b % nbuf].numpy()[:] = raw[2*n*b:2*n*(b+1)]
        t.submit_u8_host(bufs[b % nbuf].numpy())
        print(nbuf, "block", b, "staging", t.last_staging(), t.stream_info())
        a, s = t.ring_acquire(); got.append(a.copy()); t.ring_release()
    t.flush(); t.destroy()
    for b in range(nblk):
        print("  block", b, "max diff", float(np.abs(got[b][0] - want[b][0]).max()))
print("---- a launch per block (flush behind every submit), host bytes / device bytes")
for mode in ("host", "dev"):
    bufs = [torch.empty(2*n, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
    t = mk(); t.streaming(2)
    got = []
    for b in range(nblk):
        if mode == "host":
            dev.lib.wr_dev_wait_uploads(dev.h)
            bufs[b % 4].numpy()[:] = raw[2*n*b:2*n*(b+1)]
            t.submit_u8_host(bufs[b % 4].numpy())
        else:
            t.submit_u8_device(x[2*n*b:2*n*(b+1)], n)
        a, s = t.ring_acquire(); got.append(a.copy()); t.ring_release()
        t.flush()
    print(mode, t.stream_info(), [float(np.abs(got[b][0] - want[b][0]).max()) for b in range(nblk)])
    t.destroy()
print("---- closes in odd places: wr_dev_sync behind block 1's audio (what a SpectrumSink's first push does)")
from webradio_amd.device import Spectrum
xf = torch.from_numpy(((raw.astype(np.float32) - 128.0) / 128.0)).cuda()
for variant in ("sync@1", "push-every-block", "sync@0,1"):
    bufs = [torch.empty(2*n, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    t = mk(); t.streaming(2); sp = Spectrum(dev, 512)
    got = []
    for b in range(nblk):
        dev.lib.wr_dev_wait_uploads(dev.h)
        bufs[b % 2].numpy()[:] = raw[2*n*b:2*n*(b+1)]
        t.submit_u8_host(bufs[b % 2].numpy())
        a, s = t.ring_acquire(); got.append(a.copy())
        if variant == "sync@1" and b == 1: dev.sync()
        if variant == "sync@0,1" and b <= 1: dev.sync()
        if variant == "push-every-block":
            sp.push_device(xf[2*n*b:2*n*(b+1)], n)
            if b == 1: sp.get_db()
        t.ring_release()
    print(variant, t.stream_info(), [float(np.abs(got[b][0] - want[b][0]).max()) for b in range(nblk)])
    t.flush(); sp.destroy(); t.destroy()
