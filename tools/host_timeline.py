"""usage: python tools/host_timeline.py <dir with rocprofv3 csv traces> -- where the time of a host_bench block goes:
GPU activity (kernels + copies) per block period, the idle gaps, and the HIP calls that took the host longest."""
import csv, glob, sys, collections
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(d + "/**/*" + pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
k = rows("kernel_trace.csv"); m = rows("memory_copy_trace.csv"); a = rows("hip_api_trace.csv")
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28]) for r in k]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:20]) for r in m]
ev.sort()
conv = [e for e in ev if "u8_to_f32" in e[2]]
if len(conv) < 12:
    conv = [e for e in ev if "k_tuner_ddc" in e[2]]
t0, t1 = conv[len(conv) // 2][0], conv[len(conv) // 2 + 8][0]
print("period over 8 blocks in the middle: %.1f us per block" % ((t1 - t0) / 8e3))
win = [e for e in ev if t0 <= e[0] < t1]
busy = collections.defaultdict(float)
for s, e, n in win:
    busy[n] += (e - s) / 8e3
tot = 0
for n, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("  %-30s %7.1f us per block" % (n, v)); tot += v
print("  GPU busy %.1f us per block" % tot)
gaps = []
last_end = win[0][1]
for s, e, n in win[1:]:
    if s > last_end:
        gaps.append(((s - last_end) / 1e3, n))
    last_end = max(last_end, e)
print("  idle gaps: %.1f us per block; largest:" % (sum(g for g, _ in gaps) / 8), sorted(gaps, reverse=True)[:8])
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in a]
api = [x for x in api if t0 <= x[0] < t1]
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in api:
    tot[n][0] += 1; tot[n][1] += (e - s) / 8e3
print("HIP calls per block in that window:")
for n, (c, v) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-28s %5.1f calls %8.1f us" % (n, c / 8, v))
if len(sys.argv) > 2:      # one block's host calls in order: start (us from the block's first call), duration, gap before it
    b0 = conv[len(conv) // 2 + 2][0]
    first = [x for x in sorted(api) if x[2] == "hipHostGetDevicePointer" and x[0] <= b0]
    s0 = first[-1][0] if first else t0
    seq = [x for x in sorted(api) if s0 <= x[0] < s0 + 700000 and x[2] not in ("hipSetDevice", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipGetLastError")]
    prev = s0
    for s, e, n in seq[:40]:
        print("  +%7.1f us  %-24s %7.1f us   (host gap before: %6.1f us)" % ((s - s0) / 1e3, n, (e - s) / 1e3, (s - prev) / 1e3))
        prev = e
if len(sys.argv) > 2:      # the GPU's side of the same blocks
    print("GPU activity from the same point on (start us, duration us, what):")
    for s, e, n in [x for x in ev if s0 <= x[0] < s0 + 700000][:40]:
        print("  +%7.1f us  %7.1f us  %s" % ((s - s0) / 1e3, (e - s) / 1e3, n))
