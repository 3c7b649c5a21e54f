#!/bin/bash
# usage: tools/kstats.sh <cmd...> -> per-kernel mean/min duration from rocprofv3 --kernel-trace (µs)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o ks -- "$@" > /tmp/ks.log 2>&1
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in sorted(glob.glob('/tmp/ks/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith('k_'):
        v2 = sorted(v)
        last = sorted(v[-200:])                  # in launch order: the steady state of a long run (clock ramp, see bench.py --settle-ms)
        print('%-40s n=%4d  median %.1f  min %.1f  max %.1f us   last %d launches: median %.1f' % (k, len(v), v2[len(v2) // 2], v2[0], v2[-1], len(last), last[len(last) // 2]))
PY
