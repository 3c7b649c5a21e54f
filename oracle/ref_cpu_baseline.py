#!/usr/bin/env python3
"""bench.py's cpu_baseline leg with the REAL reference (kind "reference"): oracle/_ref/libwr_ref_chain.so's
ref_bench_receivers -- the reference's own DownConverter / LowPass / Demodulator classes wired as radio.cxx:68-83 wires
them, T pipeline threads with disjoint subsets of the 256 receivers -- on the C2 workload's synthetic stream.  Runs in a
process of its own (the library's FFTW calls go to hipFFTW, i.e. the system's HIP runtime; bench.py's process holds
torch's), prints ONE JSON object.  TEST / MEASUREMENT INFRASTRUCTURE, like everything under oracle/.

    python oracle/ref_cpu_baseline.py [blocks=0 (about 12 s)] [channels=256]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import wr_oracle as o  # noqa: E402
from webradio_amd import synth  # noqa: E402


def mem_available_gb():
    try:
        for l in open("/proc/meminfo"):
            if l.startswith("MemAvailable"):
                return int(l.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    channels = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    if o.ref_chain() is None:
        print(json.dumps({"error": "oracle/_ref/libwr_ref_chain.so not built"}))
        return 1
    cfg = synth.C2
    n = cfg["block_frames"]
    ifs = synth.c2_ifs(channels)
    # the reference keeps, per receiver, the mixer's full-rate output and the channel filter's history + block copy
    # (2 x 32 MB at C2) and per thread two copies of the tuner block: stay well inside the box's memory
    need_gb = channels * 0.07 + min(channels, len(os.sched_getaffinity(0))) * 0.07
    have_gb = mem_available_gb()
    if have_gb and need_gb > 0.6 * have_gb:
        print(json.dumps({"error": "the reference's buffers for %d receivers need about %.0f GB, %.0f GB available" % (channels, need_gb, have_gb)}))
        return 1
    iq = synth.fm_stream(n, cfg["input_rate"], ifs[::4], seed=12345)
    args = (cfg["input_rate"], ifs, cfg["chan_passband"], cfg["chan_rate"], o.FM, cfg["audio_passband"], cfg["audio_rate"], iq)
    cores = max(1, min(len(os.sched_getaffinity(0)), channels))
    one, _, _ = o.ref_bench_receivers(*args, 1, 1) if channels <= 256 else (0, 0, 0)
    if blocks <= 0:
        probe, _, _ = o.ref_bench_receivers(*args, 2, cores)
        blocks = int(max(2, min(256, round(12.0 / max(probe / 2.0, 1e-3)))))
    secs, frames, s = o.ref_bench_receivers(*args, blocks, cores)
    print(json.dumps({
        "value": round(n * blocks / secs / 1e6, 4),
        "unit": "complex Msamples/s (tuner input, all %d channels)" % channels,
        "cores": cores, "kind": "reference",
        "sample": "the reference's own dsp/{dspblock,downconverter,lowpass,demodulator}.cxx (oracle/_ref/libwr_ref_chain.so): "
                  "%d receivers x %d block(s) of %d frames on %d pipeline threads (disjoint receiver subsets), %.1f s" % (
                      channels, blocks, n, cores, secs),
        "audio_frames": frames, "audio_abs_sum": round(s, 3),
        "one_core": {"value": round(n / one / 1e6, 4), "cores": 1,
                     "sample": "%d receivers x 1 block of %d frames on one thread, %.1f s" % (channels, n, one)},
    }))
    return 0


if __name__ == "__main__":
    sys.exit(main())
