#!/bin/bash
# usage: tools/try_steady.sh -> per-variant steady-state kernel time: 1500 launches back to back, median of the last 400
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st
  QT_REPS=1500 QT_BLOCKS=12 QT_PROFILE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o st -- python $R/tools/quick_time.py 256 rotate > /tmp/st.log 2>&1
  python3 - "$(basename $v .so)" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tuner_ddc" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
d = sorted(x[1] for x in rows[-400:]); e = sorted(x[1] for x in rows[100:300])
print("%-16s steady median %.1f us (min %.1f)   early (launch 100-300) median %.1f" % (sys.argv[1], d[len(d)//2], d[0], e[len(e)//2]))
PY
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
