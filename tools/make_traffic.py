#!/usr/bin/env python3
"""Writes profiles/traffic.json -- the HBM bytes per launch bench.py quotes in `roofline.traffic` -- from the PMC dumps of
tools/profile_round.sh (gpurun_out/<round>_pmc_fetch_size.txt / _pmc_write_size.txt: FETCH_SIZE and WRITE_SIZE per kernel and
launch shape, separate rocprofv3 --pmc passes).  No hand step: `python tools/make_traffic.py r05` (profile_round.sh calls it).

Reading of the counters, settled in r05 on known byte counts (tools/ubench_fetch.hip, profiles/r05_fetch_calibration.txt):
both are KB per launch; on gfx950 every read request the L2 sends to memory is 128 bytes and FETCH_SIZE tallies it at 64 --
for 4-, 8- and 16-byte-per-lane loads alike, streaming or in the DDC's 512-byte windows 3200 bytes apart (0.500 x the bytes
asked for in all of them) -- so FETCH_SIZE is DOUBLED for every kernel (r04 doubled it for the FFT passes only and took the
DDC's as counted: its 1.12 x algorithmic was 1.5 x); WRITE_SIZE counts the bytes written (1.000 x for 8-byte plain and
write-through stores, 4-byte stores)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^(?:void )?(\S+?(?:<[^>]*>)?) \[(\d+) workgroups\] (\w+) mean=([0-9.]+) KB per launch \(n=(\d+)\)")
ALGO_PER_BLOCK = 8.512 * 4.0e6


def read(path):
    rows = {}
    if not os.path.exists(path):
        return rows
    for l in open(path):
        m = LINE.match(l.strip())
        if m:
            rows[(m.group(1), int(m.group(2)))] = (float(m.group(4)), int(m.group(5)))
    return rows


def main(rnd):
    src = os.path.join(ROOT, "gpurun_out")
    kb = 1024.0
    rule = ("2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies every 128-byte read request at 64 bytes, whatever the load's width "
            "(profiles/r05_fetch_calibration.txt: 0.500 x the bytes asked for with 4, 8 and 16 bytes per lane, streaming and in "
            "the DDC's windows); WRITE_SIZE as counted")
    out = {"round": rnd, "nco": "rotate", "fetch_rule": rule,
           "made_by": "tools/make_traffic.py from gpurun_out/%s_pmc*_fetch_size.txt and _write_size.txt (tools/profile_round.sh)" % rnd}
    # the streaming launch: ONE launch of `blocks` blocks in its PMC pass (profile_round.sh: no warm-up, no settling steps)
    fs, ws = read(os.path.join(src, "%s_pmc_stream_fetch_size.txt" % rnd)), read(os.path.join(src, "%s_pmc_stream_write_size.txt" % rnd))
    st = [k for k in fs if k[0].startswith("k_tuner_stream") and k in ws]
    if st:
        k = st[0]
        blocks = int(open(os.path.join(src, "%s_pmc_stream_blocks.txt" % rnd)).read().split()[0])
        per = (2 * fs[k][0] + ws[k][0]) * kb / blocks
        out["streaming"] = {"kernel": "%s, %d workgroups" % k, "blocks_in_the_counted_launch": blocks,
                            "fetch_size_kb_counted": fs[k][0], "write_size_kb": ws[k][0],
                            "hbm_bytes_per_block": int(round(per)), "x_algorithmic": round(per / ALGO_PER_BLOCK, 3)}
    f, w = read(os.path.join(src, "%s_pmc_fetch_size.txt" % rnd)), read(os.path.join(src, "%s_pmc_write_size.txt" % rnd))
    # the riding DDC kernel (audio decimation 5, two lane groups per wave) in its two launch shapes: the larger grid is the
    # four-block launch, the smaller the one-block launch (bench.py --no-stream; secondary.c2_* in the default run)
    ddc = sorted([k for k in f if k[0].startswith("k_tuner_ddc<2, true, 5u") and k in w], key=lambda k: -k[1])
    if len(ddc) >= 2:
        four, one = ddc[0], ddc[-1]
        out.update({
            "kernel": "%s (DDC of block b + demod/audio filter of block b-1 in one launch), %d workgroups" % four,
            "workload": "C2, 256 channels, 4 000 000-frame blocks, 12 resident blocks cycled, 4 blocks per launch (bench.py --no-stream --blocks-per-launch 4)",
            "fetch_size_kb_counted": f[four][0], "write_size_kb": w[four][0], "launches_counted": [f[four][1], w[four][1]],
            "hbm_bytes_per_launch": int(round((2 * f[four][0] + w[four][0]) * kb)),
            "frames_per_launch": 16000000,
            "x_algorithmic": round((2 * f[four][0] + w[four][0]) * kb / (4 * ALGO_PER_BLOCK), 3),
            "one_block_per_launch": {"frames_per_launch": 4000000, "workgroups": one[1], "fetch_size_kb_counted": f[one][0],
                                     "write_size_kb": w[one][0],
                                     "hbm_bytes_per_launch": int(round((2 * f[one][0] + w[one][0]) * kb)),
                                     "x_algorithmic": round((2 * f[one][0] + w[one][0]) * kb / ALGO_PER_BLOCK, 3)},
        })
    # (the waterfall batch's launches: the LARGEST grids -- the frontend secondary's pushes and polls transform single frames
    # with the same kernels, r06)
    p1 = sorted([k for k in f if k[0].startswith("k_fft64k_pass1") and k in w], key=lambda k: -k[1])
    p2 = sorted([k for k in f if k[0].startswith("k_fft64k_pass2") and k in w], key=lambda k: -k[1])
    if p1 and p2:
        p1, p2 = p1[0], p2[0]
        out["c3"] = {
            "kernels": "%s + %s, 121 frames of 65536 points per launch pair (one 4 M-frame block at 50 %% overlap)" % (p1[0], p2[0]),
            "frames_per_launch": 121,
            "pass1": {"fetch_size_kb_counted": f[p1][0], "fetch_size_kb_corrected": 2 * f[p1][0], "write_size_kb": w[p1][0]},
            "pass2": {"fetch_size_kb_counted": f[p2][0], "fetch_size_kb_corrected": 2 * f[p2][0], "write_size_kb": w[p2][0]},
            "hbm_bytes_per_launch": int(round((2 * f[p1][0] + w[p1][0] + 2 * f[p2][0] + w[p2][0]) * kb)),
        }
    path = os.path.join(src, "%s_traffic.json" % rnd)
    json.dump(out, open(path, "w"), indent=2)
    print("wrote", path, json.dumps({k: v for k, v in out.items() if k in ("streaming", "hbm_bytes_per_launch", "x_algorithmic")}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
