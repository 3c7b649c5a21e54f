R=$GRAFT_REPO_ROOT
for c in 1 2 4 8; do
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st
  QT_COALESCE=$c QT_REPS=$((1600 / c)) QT_BLOCKS=$((12 / c > 2 ? 12 / c : 3)) QT_PROFILE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o st -- python $R/tools/quick_time.py 256 rotate > /tmp/st.log 2>&1
  python3 - $c <<'PY'
import csv, glob, sys
c = int(sys.argv[1]); rows = []
for f in glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tuner_ddc" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
d = sorted(x[1] for x in rows[-max(20, 400 // c):])
print("%d blocks per launch: steady median %.1f us per launch = %.2f us per 4 M-frame block" % (c, d[len(d)//2], d[len(d)//2] / c))
PY
done
