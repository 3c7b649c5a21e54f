#!/bin/bash
# The same launch, 1500 times back to back from an idle GPU: kernel duration by time since the first
# launch (rocprofv3 kernel trace).  Shows the clock ramp bench.py's --settle-ms waits out.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ramp
QT_REPS=1500 QT_BLOCKS=12 QT_PROFILE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/ramp -o ramp -- python $R/tools/quick_time.py 256 rotate > /tmp/ramp.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/ramp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tuner_ddc" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
t0 = rows[0][0]
print("k_tuner_ddc<ROTATE, folded taps, post stage riding>, C2, 12 resident blocks cycled, launched back to back from an idle MI355X")
for i in range(0, len(rows), 75):
    seg = rows[i:i + 75]
    d = sorted(x[1] for x in seg)
    gap = (seg[-1][0] - seg[0][0]) / 1e3 / max(1, len(seg) - 1)
    print("launch %4d at %6.2f ms: median %.1f us  min %.1f  max %.1f   start-to-start %.1f us" % (
        i, (seg[0][0] - t0) / 1e6, d[len(d) // 2], d[0], d[-1], gap))
PY
