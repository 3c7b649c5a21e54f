"""Quick C2-size timing of the fused tuner path (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner

c2 = synth.C2
fs, n = c2["input_rate"], c2["block_frames"] * int(os.environ.get("QT_COALESCE", "1"))   # n blocks per launch
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = {"split": capi.WR_NCO_SPLIT, "exact": capi.WR_NCO_EXACT, "rotate": capi.WR_NCO_ROTATE}
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["split", "exact"]
ifs = synth.c2_ifs(nch)
stream = torch.cuda.current_stream().cuda_stream
dev = Device(0, stream)
nb = int(os.environ.get("QT_BLOCKS", "1"))        # 12: cycle through more input than the Infinity Cache holds, as bench.py does
xs = synth.fm_stream_torch(n * nb, fs, ifs[::4], "cuda")
blocks = [xs[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
x = blocks[0]
torch.cuda.synchronize()
for name in which:
    t = Tuner(dev, fs, nch, n, modes[name])
    if os.environ.get("QT_KEEP") == "1":
        t.keep_stages(capi.WR_STAGE_DEMOD)
    # QT_L1B: a second channel stage of that many taps, 250 k -> 50 k (the audio filter then 50 k -> 10 k)
    st2 = (int(os.environ["QT_L1B"]), c2["chan_rate"] // 16, c2["chan_rate"] // 5) if os.environ.get("QT_L1B") else None
    mixed = os.environ.get("QT_MIXED") == "1"     # per-lane taps: a different passband per channel
    for i, f in enumerate(ifs):
        odd = os.environ.get("QT_ONE_ODD") == "1" and i == 70       # one receiver with its own passband
        t.add_receiver(f, c2["chan_passband"] + (3_000_000 * (i % 7) if mixed else 0) + (3_000_000 if odd else 0),
                       c2["chan_rate"], capi.WR_FM,
                       c2["audio_passband"] // (5 if st2 else 1), c2["audio_rate"] // (5 if st2 else 1),
                       fir_lengths=(int(os.environ.get("QT_L1", "64")), int(os.environ.get("QT_L2", "64")))
                       if os.environ.get("QT_L1") or os.environ.get("QT_L2") else None,   # 128 / 256: k_tuner_ddc_long; r05: QT_L2, the audio filter
                       stage2=st2)
    for i in range(4):
        t.submit_device(blocks[i % nb], n)
    torch.cuda.synchronize()
    reps = 3 if name == "exact" else int(os.environ.get("QT_REPS", "20"))
    t.profile(int(os.environ.get("QT_PROFILE", "1")))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        t.submit_device(blocks[i % nb], n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nl, dms = t.profile_read()
    print("%s: ddc kernel %.4f ms (%d launches)" % (name, dms, nl))
    print("%s: %d ch, %.3f ms/block, %.1f Msps, %.1f GB/s algorithmic (%.2f%% of 8 TB/s)" % (
        name, nch, ms, n / ms / 1e3, n * 8.512 / ms / 1e6, n * 8.512 / ms / 1e6 / 8000 * 100))
    t.destroy()
