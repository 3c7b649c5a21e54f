#!/usr/bin/env python3
"""profiles/r05_bench_boxes.txt from the per-box files tools/bench_box.sh left in gpurun_out/ (one fresh lease each):
the driver's invocation `python bench.py --steps 20 --warmup 5` on the boxes of the pool."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r05_box_*.json"))):
    try:
        j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        continue
    hf = {(r.get("source", "")[:3], r.get("audio", "")[:7], r.get("staging", "")[:6]): r.get("ms_per_block")
          for r in (j.get("secondary", {}).get("host_fed") or {}).get("runs", [])}
    sec = j.get("secondary", {})
    rows.append((os.path.basename(path)[8:-5], j["value"] / 1e3, j["ms_per_step"] * 1e3, j["roofline"]["kernel_ms"] * 1e3,
                 j["roofline"]["frac"], j["value_one_block_per_launch"] / 1e3,
                 (sec.get("c2_four_blocks_per_launch") or {}).get("value", 0.0) / 1e3, sec["c3"]["ms_per_block"] * 1e3,
                 sec["c3"]["roofline"]["frac"], hf.get(("u8 ", "on time", "sparse")), hf.get(("f32", "on time", "sparse"))))
out = ["python bench.py --steps 20 --warmup 5 --no-cpu-baseline, one fresh lease per row (tools/bench_box.sh; this table: tools/bench_boxes_table.py)",
       "r05: `value` = the streaming launch (20 blocks through ONE persistent launch, opened and closed inside the timed region)",
       "box (UTC)          Gsps   us/step   the launch us   frac    launch per block Gsps   4 blocks/launch Gsps   streaming / 4 blocks   C3 us/block  C3 frac   host on time ms: u8 / f32"]
FINAL = "20260930"                                # leases of the round's final code (the drain shared) carry this date
for r in rows:
    out.append("%-16s %7.1f %8.2f %12.1f %10.4f %14.1f %22.1f %15.3f %17.1f %9.4f      %s / %s" % (r[:7] + (r[1] / r[6] if r[6] else 0.0,) + r[7:]))
if rows:
    v = [r[1] for r in rows]
    out.append("%d boxes: %.1f-%.1f Gsps, mean %.1f" % (len(v), min(v), max(v), sum(v) / len(v)))
    fin = [r for r in rows if r[0].startswith(FINAL)]
    if fin:
        v, q = [r[1] for r in fin], sorted(r[1] / r[6] for r in fin if r[6])
        out.append("the final code's %d leases: %.1f-%.1f Gsps, mean %.1f; launch per block mean %.1f, four blocks per launch mean %.1f; "
                   "streaming / four blocks %.3f-%.3f, median %.3f, >= 0.97 on %d of %d"
                   % (len(v), min(v), max(v), sum(v) / len(v), sum(r[5] for r in fin) / len(fin), sum(r[6] for r in fin) / len(fin),
                      q[0], q[-1], q[len(q) // 2], sum(1 for x in q if x >= 0.97), len(q)))
out.append("-- the rows before " + FINAL + ": before the drain was shared (profiles/r05_stream_timeline.txt).  Twelve runs in a row on ONE box of the final code\n"
           "   (tools/scratch/bench_k.sh): 32.5-33.3 us per step (119.9-123.2 Gsps), the launch 615-632 us -- the spread between rows is between boxes\n"
           "   (and one 20-block sample each), not between runs.")
text = "\n".join(out) + "\n"
sys.stdout.write(text)
open(os.path.join(ROOT, "profiles", "r05_bench_boxes.txt"), "w").write(text)
