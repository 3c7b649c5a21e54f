"""development: the frontend 'streaming + newest frame' loop of bench.py, many times: does a launch run into its deadline?"""
import sys, time, os
sys.path.insert(0, '.')
import torch
from webradio_amd import capi, synth
from webradio_amd.device import Device, Tuner, Spectrum
cfg = synth.C2; n = cfg["block_frames"]; ifs = synth.c2_ifs(256); nb = 12
x = synth.fm_stream_torch(n * nb, cfg["input_rate"], ifs[::4], "cuda", seed=1)
blocks = [x[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
dev = Device(0, torch.cuda.current_stream().cuda_stream)
t = Tuner(dev, cfg["input_rate"], 256, n * 4, capi.WR_NCO_ROTATE)
for f in ifs:
    t.add_receiver(f, cfg["chan_passband"], cfg["chan_rate"], capi.WR_FM, cfg["audio_passband"], cfg["audio_rate"])
spec = Spectrum(dev, 65536, 32768)
t.streaming(True)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 60
poll = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
bad = 0
for rep in range(reps):
    t0 = time.perf_counter()
    try:
        for i in range(k):
            t.submit_device(blocks[i % nb], n)
            spec.push_device(blocks[i % nb], n)
            if poll and (i + 1) % poll == 0:
                spec.get_db()
        t.flush(); torch.cuda.synchronize()
        t.fetch(0, capi.WR_STAGE_AUDIO, n)
    except Exception as e:
        bad += 1
        print("rep", rep, "FAILED after %.2f s:" % (time.perf_counter() - t0), str(e)[:160], flush=True)
        continue
    print("rep", rep, "%.1f us/blk" % ((time.perf_counter() - t0) / k * 1e6), t.stream_info(), spec.lazy_info(), flush=True)
print("failures:", bad, "of", reps)
