#!/usr/bin/env python3
"""Writes profiles/traffic.json -- the HBM bytes per launch bench.py quotes in `roofline.traffic` -- from the PMC dumps of
tools/profile_round.sh (gpurun_out/<round>_pmc_fetch_size.txt / _pmc_write_size.txt: FETCH_SIZE and WRITE_SIZE per kernel and
launch shape, separate rocprofv3 --pmc passes).  No hand step: `python tools/make_traffic.py r04` (profile_round.sh calls it).

Reading of the counters (MI355X_MICROARCH.md, HBM / rocprofv3): both are KB per launch; WRITE_SIZE as counted; FETCH_SIZE as
counted for the DDC (8-byte loads per lane) and doubled for the FFT passes (wide coalesced streaming reads, which gfx950
tallies at 64 of their 128 bytes)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^(?:void )?(\S+?(?:<[^>]*>)?) \[(\d+) workgroups\] (\w+) mean=([0-9.]+) KB per launch \(n=(\d+)\)")


def read(path):
    rows = {}
    for l in open(path):
        m = LINE.match(l.strip())
        if m:
            rows[(m.group(1), int(m.group(2)))] = (float(m.group(4)), int(m.group(5)))
    return rows


def main(rnd):
    src = os.path.join(ROOT, "gpurun_out")
    f = read(os.path.join(src, "%s_pmc_fetch_size.txt" % rnd))
    w = read(os.path.join(src, "%s_pmc_write_size.txt" % rnd))
    # the riding DDC kernel (audio decimation 5, two lane groups per wave) in its two launch shapes: the larger grid is the
    # four-block launch of bench.py's default, the smaller the one-block launch of secondary.c2_one_block_per_launch
    ddc = sorted([k for k in f if k[0].startswith("k_tuner_ddc<2, true, 5u") and k in w], key=lambda k: -k[1])
    assert len(ddc) >= 2, "the PMC dump holds %d launch shapes of the riding DDC kernel" % len(ddc)
    four, one = ddc[0], ddc[-1]
    p1 = [k for k in f if k[0].startswith("k_fft64k_pass1") and k in w][0]
    p2 = [k for k in f if k[0].startswith("k_fft64k_pass2") and k in w][0]
    kb = 1024.0
    out = {
        "round": rnd,
        "made_by": "tools/make_traffic.py from gpurun_out/%s_pmc_fetch_size.txt and _pmc_write_size.txt (tools/profile_round.sh)" % rnd,
        "nco": "rotate",
        "kernel": "%s (DDC of block b + demod/audio filter of block b-1 in one launch), %d workgroups" % four,
        "workload": "C2, 256 channels, 4 000 000-frame blocks, 12 resident blocks cycled, 4 blocks per launch (bench.py default)",
        "fetch_size_kb": f[four][0], "write_size_kb": w[four][0], "launches_counted": [f[four][1], w[four][1]],
        "hbm_bytes_per_launch": int(round((f[four][0] + w[four][0]) * kb)),
        "frames_per_launch": 16000000,
        "one_block_per_launch": {"frames_per_launch": 4000000, "workgroups": one[1], "fetch_size_kb": f[one][0],
                                 "write_size_kb": w[one][0],
                                 "hbm_bytes_per_launch": int(round((f[one][0] + w[one][0]) * kb))},
        "c3": {
            "kernels": "%s + %s, 121 frames of 65536 points per launch pair (one 4 M-frame block at 50 %% overlap)" % (p1[0], p2[0]),
            "frames_per_launch": 121,
            "pass1": {"fetch_size_kb_counted": f[p1][0], "fetch_size_kb_corrected": 2 * f[p1][0], "write_size_kb": w[p1][0]},
            "pass2": {"fetch_size_kb_counted": f[p2][0], "fetch_size_kb_corrected": 2 * f[p2][0], "write_size_kb": w[p2][0]},
            "hbm_bytes_per_launch": int(round((2 * f[p1][0] + w[p1][0] + 2 * f[p2][0] + w[p2][0]) * kb)),
            "correction": "FETCH_SIZE doubled: on gfx950 rocprofv3 tallies the 128-byte requests of wide coalesced streaming reads "
                          "at 64 bytes (MI355X_MICROARCH.md, HBM); WRITE_SIZE as counted",
        },
    }
    path = os.path.join(src, "%s_traffic.json" % rnd)
    json.dump(out, open(path, "w"), indent=2)
    print("wrote", path, ": C2 %.1f MB per four-block launch, C3 %.1f MB per 121 frames" % (
        out["hbm_bytes_per_launch"] / 1e6, out["c3"]["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
