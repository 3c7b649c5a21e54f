#!/bin/bash
# usage: [C=4] tools/try_steady.sh -> per-variant steady-state kernel time: ~1500 blocks back to back (C blocks per
# launch, default 1), median of the last launches, per 4 M-frame block
R=$GRAFT_REPO_ROOT
C=${C:-1}
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st
  QT_COALESCE=$C QT_REPS=$((1600 / C)) QT_BLOCKS=$((12 / C > 2 ? 12 / C : 3)) QT_PROFILE=0 timeout 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o st -- python $R/tools/quick_time.py 256 rotate > /tmp/st.log 2>&1
  python3 - "$(basename $v .so)" $C <<'PY'
import csv, glob, sys
c = int(sys.argv[2]); rows = []
for f in glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tuner_ddc" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
d = sorted(x[1] for x in rows[-max(20, 400 // c):])
print("%-16s %d block(s) per launch: steady median %.1f us per launch (min %.1f) = %.2f us per 4 M-frame block" % (sys.argv[1], c, d[len(d)//2], d[0], d[len(d)//2] / c))
PY
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
