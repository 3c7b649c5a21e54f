#!/bin/bash
# kernel durations and the kernel sequence of the on-time drop-in path (tests/cxx/host_bench), f32 and u8 sources
export LD_LIBRARY_PATH=$PWD/webradio_amd/lib:$PWD/webradio_amd/host:$PWD/tests/cxx:$LD_LIBRARY_PATH
R=$PWD; cd /tmp && export TMPDIR=/tmp
for src in f32 u8; do
  rm -rf /tmp/ht
  WEBRADIO_QUIET=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ht -o ht -- $R/tests/cxx/host_bench 256 100 4000000 $src > /tmp/ht.log 2>&1
  echo "== $src"
  python3 - <<PY
import csv, glob
f=glob.glob("/tmp/ht/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
rows=[]
for f in glob.glob("/tmp/ht/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
mid=len(rows)//2
t0=rows[mid][0]
for s,e,n in rows[mid:mid+14]:
    print("  +%8.1f us  dur %7.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
done
