#!/usr/bin/env python3
"""profiles/<round>_bench_boxes.txt from the per-box files tools/bench_box.sh left in gpurun_out/ (one fresh lease each):
the driver's invocation `python bench.py --steps 20 --warmup 5` on the boxes of the pool.
    python tools/bench_boxes_table.py r06"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_box_*.json" % rnd))):
    try:
        j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        continue
    sec = j.get("secondary", {})
    hf = {}
    for r in (sec.get("host_fed") or {}).get("runs", []):
        key = (r.get("source", "")[:3], r.get("audio", "")[:7], "nostream" if "WEBRADIO_STREAM=0" in r.get("staging", "") else r.get("staging", "")[:6])
        hf[key] = r.get("ms_per_block")
    fe = sec.get("frontend") or {}
    c1 = sec.get("c1") or {}
    rows.append({
        "box": os.path.basename(path)[len(rnd) + 5:-5], "gsps": j["value"] / 1e3, "us": j["ms_per_step"] * 1e3,
        "launch_us": j["roofline"]["kernel_ms"] * 1e3, "frac": j["roofline"]["frac"],
        "one": j.get("value_one_block_per_launch", 0.0) / 1e3, "four": (sec.get("c2_four_blocks_per_launch") or {}).get("value", 0.0) / 1e3,
        "c3_us": sec["c3"]["ms_per_block"] * 1e3, "c3_frac": sec["c3"]["roofline"]["frac"],
        "c1_us": c1.get("ms_per_block", 0) * 1e3, "c1s_us": (c1.get("streaming") or {}).get("ms_per_block", 0) * 1e3,
        "fe_launch": (fe.get("per_block_launches") or {}).get("us_per_block"), "fe_closed": (fe.get("streaming_closed_per_block") or {}).get("us_per_block"),
        "fe_newest": (fe.get("streaming_newest_frame") or {}).get("us_per_block_between_polls"), "fe_poll": (fe.get("streaming_newest_frame") or {}).get("us_per_poll"),
        "dev_late": hf.get(("dev", "one blo", "none: ")), "dev_late_off": hf.get(("dev", "one blo", "nostream")), "dev_on": hf.get(("dev", "on time", "none: ")),
        "u8_on": hf.get(("u8 ", "on time", "sparse")), "u8_late": hf.get(("u8 ", "one blo", "sparse")),
        "f32_on": hf.get(("f32", "on time", "sparse")), "f32_late": hf.get(("f32", "one blo", "sparse")),
    })
out = ["python bench.py --steps 20 --warmup 5 --no-cpu-baseline, one fresh lease per row (tools/bench_box.sh; this table: tools/bench_boxes_table.py %s)" % rnd,
       "`value` = the streaming launch: 20 blocks through ONE persistent launch, opened and closed inside the timed region",
       "box (UTC)          Gsps   us/step  the launch us    frac   launch/block Gsps  4 blocks/launch Gsps   C3 us (frac)    C1 us: launch / stream   "
       "FrontEnd us: launches+waterfall / stream closed / stream newest frame (+ per poll)   host classes ms per block: device late / (no stream) / on time | u8 on time / late | f32 on time / late"]
for r in rows:
    out.append("%-16s %7.1f %8.2f %12.1f %9.4f %14.1f %18.1f %12.1f (%.3f) %10.1f / %.1f   %18s / %s / %s (+ %s)   %14s / %s / %s | %s / %s | %s / %s" % (
        r["box"], r["gsps"], r["us"], r["launch_us"], r["frac"], r["one"], r["four"], r["c3_us"], r["c3_frac"], r["c1_us"], r["c1s_us"],
        r["fe_launch"], r["fe_closed"], r["fe_newest"], r["fe_poll"], r["dev_late"], r["dev_late_off"], r["dev_on"], r["u8_on"], r["u8_late"], r["f32_on"], r["f32_late"]))
if rows:
    v = [r["gsps"] for r in rows]
    u = [r["us"] for r in rows]
    out.append("%d boxes: %.1f-%.1f Gsps, mean %.1f (%.2f-%.2f us per step); launch per block mean %.1f, four blocks per launch mean %.1f"
               % (len(v), min(v), max(v), sum(v) / len(v), min(u), max(u), sum(r["one"] for r in rows) / len(rows), sum(r["four"] for r in rows) / len(rows)))
path = os.path.join(ROOT, "profiles", "%s_bench_boxes.txt" % rnd)
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
